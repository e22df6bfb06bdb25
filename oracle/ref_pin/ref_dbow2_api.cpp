// ORACLE — TEST INFRASTRUCTURE ONLY.  `make -C oracle/ref_pin dbow2`: the reference's own vendored DBoW2 (Thirdparty/DBoW2/DBoW2/*.cpp, TemplatedVocabulary.h,
// DUtils/Random.cpp), compiled unmodified, where it lies, against the stub cv:: layer -- ORBVocabulary::loadFromTextFile (what System.cc:64-73 calls on
// ORBvoc.txt) and transform(features, BowVector&, FeatureVector&, levelsup) (what Frame::ComputeBoW calls, src/Frame.cc:474-481).  compare_slices.py
// runs it against the oracle's restatement of the loader and the transform on vocabularies written in the text format.
#include <cstdint>
#include <cstring>
#include <vector>
#include "Thirdparty/DBoW2/DBoW2/FORB.h"
#include "Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h"

typedef DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB> ORBVocabulary;      // include/ORBVocabulary.h:30-31

extern "C" {
// returns 0, or -1 when the file does not load.  info[5] = {k, L, scoring, weighting, words}; BowVector as (word, value) in key order; FeatureVector as CSR
int ref_vocab_compute_bow(const char* path, const uint8_t* feat, int n, int levelsup, int32_t* info, int32_t* bw, double* bv, int32_t* nb,
                          int32_t* fn, int32_t* fp, int32_t* ff, int32_t* nf) {
    ORBVocabulary voc;
    if (!voc.loadFromTextFile(path)) return -1;
    info[0] = voc.getBranchingFactor(); info[1] = voc.getDepthLevels(); info[2] = (int)voc.getScoringType(); info[3] = (int)voc.getWeightingType(); info[4] = (int)voc.size();
    std::vector<cv::Mat> vCurrentDesc;                      // Converter::toDescriptorVector(mDescriptors): one 1 x 32 row per feature
    for (int i = 0; i < n; ++i) { cv::Mat d(1, 32, CV_8U); std::memcpy(d.data, feat + (size_t)i * 32, 32); vCurrentDesc.push_back(d); }
    DBoW2::BowVector bow; DBoW2::FeatureVector fv;
    voc.transform(vCurrentDesc, bow, fv, levelsup);
    int k = 0;
    for (DBoW2::BowVector::const_iterator it = bow.begin(); it != bow.end(); ++it, ++k) { bw[k] = (int32_t)it->first; bv[k] = it->second; }
    *nb = k;
    int j = 0, pos = 0; fp[0] = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++j) {
        fn[j] = (int32_t)it->first;
        for (size_t q = 0; q < it->second.size(); ++q) ff[pos++] = (int32_t)it->second[q];
        fp[j + 1] = pos;
    }
    *nf = j;
    return 0;
}
}
