// ORACLE — TEST INFRASTRUCTURE ONLY (oracle/ref_pin).  Stand-in declarations for the SLAM types that line-range slices of the
// reference's src/ORBmatcher.cc, src/LSDmatcher.cpp and src/Frame.cc are compiled against (`make -C oracle/ref_pin pin-stub`).
// The real headers (include/Frame.h, include/ORBmatcher.h, include/LSDmatcher.h) pull in Eigen, DBoW2, g2o and the whole map; the slices
// only touch the members declared here, with the reference's names, types and static-ness (include/Frame.h:155-219,
// include/ORBmatcher.h:40-91, include/LSDmatcher.h:37-66).  The function BODIES are the reference's own lines, cut at build time.
#pragma once
#include <climits>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>
#include "../stub_cv/stub_cv.hpp"

namespace cv {
namespace line_descriptor {
struct KeyLine {            // opencv_contrib line_descriptor/descriptor.hpp: 68 bytes
    float angle; int class_id; int octave; Point2f pt; float response; float size;
    float startPointX, startPointY, endPointX, endPointY, sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY, lineLength;
    int numOfPixels;
};
static_assert(sizeof(KeyLine) == 68, "KeyLine is 68 bytes");
}  // namespace line_descriptor

// cv::BFMatcher(NORM_HAMMING, crossCheck=false).knnMatch(query, train, matches, k): UPSTREAM-RECALL (modules/features2d/src/matchers.cpp,
// modules/core/src/batch_distance.cpp): per query row the k nearest train rows in ascending distance, equal distances in ascending train
// index; fewer than k rows when the train set is smaller.  This leaf is NOT pinned by the slices -- the code around it is.
class BFMatcher {
public:
    BFMatcher(int normType = NORM_L2, bool crossCheck = false) { assert(normType == NORM_HAMMING && !crossCheck); }
    void knnMatch(const Mat& q, const Mat& t, std::vector<std::vector<DMatch> >& matches, int k) const {
        matches.clear();
        for (int i = 0; i < q.rows; ++i) {
            std::vector<DMatch> all;
            for (int j = 0; j < t.rows; ++j) {
                int d = 0;
                for (int b = 0; b < q.cols; ++b) d += __builtin_popcount((unsigned)(q.ptr(i)[b] ^ t.ptr(j)[b]));
                all.push_back(DMatch(i, j, (float)d));
            }
            std::stable_sort(all.begin(), all.end());
            if ((int)all.size() > k) all.resize(k);
            matches.push_back(all);
        }
    }
};
}  // namespace cv

using namespace std;
using namespace cv;
using namespace cv::line_descriptor;

// include/auxiliar.h:47-74 (the comparators of lineDescriptorMAD / SerachForInitialize) is spliced in by the build right after this header.
#define SSLAM_PIN_STUB_SLAM 1

namespace StructureSLAM {
#ifndef FRAME_GRID_ROWS
#define FRAME_GRID_ROWS 48      // include/Frame.h:45-46 (the build greps the two defines and fails if they ever change)
#define FRAME_GRID_COLS 64
#endif

class Frame {
public:
    Frame() : N(0), NL(0) {}
    // src/Frame.cc:133-148, 462-472, 368-421, 423-460, 190-215 -- bodies from the reference
    void AssignFeaturesToGrid();
    bool PosInGrid(const cv::KeyPoint& kp, int& posX, int& posY);
    vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel = -1, const int maxLevel = -1) const;
    vector<size_t> GetLinesInArea(const float& x1, const float& y1, const float& x2, const float& y2, const float& r, const int minLevel = -1, const int maxLevel = -1) const;
    void lineDescriptorMAD(vector<vector<DMatch> > matches, double& nn_mad, double& nn12_mad) const;

    int N, NL;
    std::vector<cv::KeyPoint> mvKeysUn;
    cv::Mat mDescriptors, mLdesc;
    std::vector<KeyLine> mvKeylinesUn;
    static float mfGridElementWidthInv, mfGridElementHeightInv;
    std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
};

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10);
    static const int TH_LOW, TH_HIGH, HISTO_LENGTH;
    void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);
    float mfNNratio; bool mbCheckOrientation;
};

class LSDmatcher {
public:
    static int DescriptorDistance(const Mat& a, const Mat& b);
    int SerachForInitialize(Frame& InitialFrame, Frame& CurrentFrame, vector<pair<int, int> >& LineMatches);
};
}  // namespace StructureSLAM
