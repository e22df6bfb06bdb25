// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points around the reference functions that `make -C oracle/ref_pin pin-stub` cuts out of
// src/ORBmatcher.cc, src/LSDmatcher.cpp and src/Frame.cc (this file is appended to the generated oracle/_ref/ref_slices.cc; it holds no
// reference code).  compare_stub.py calls them through ctypes next to the oracle's orc_* functions on the same arrays.
namespace StructureSLAM {
float Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv, Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY;
}
using StructureSLAM::Frame;

static void fill_frame(Frame& F, const cv::KeyPoint* kp, const uint8_t* desc, int n, const float* bounds) {
    // what Frame's constructor does around the slices (src/Frame.cc:104-131): bounds, inverse cell sizes, N, then AssignFeaturesToGrid()
    Frame::mnMinX = bounds[0]; Frame::mnMaxX = bounds[1]; Frame::mnMinY = bounds[2]; Frame::mnMaxY = bounds[3];
    Frame::mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / static_cast<float>(Frame::mnMaxX - Frame::mnMinX);
    Frame::mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / static_cast<float>(Frame::mnMaxY - Frame::mnMinY);
    F.N = n; F.mvKeysUn.assign(kp, kp + n);
    if (desc) { F.mDescriptors.create(n, 32, CV_8UC1); if (n) std::memcpy(F.mDescriptors.data, desc, (size_t)n * 32); }
    F.AssignFeaturesToGrid();
}

extern "C" {
int ref_descriptor_distance(const uint8_t* a, const uint8_t* b) {
    cv::Mat A(1, 32, CV_8UC1, (void*)a), B(1, 32, CV_8UC1, (void*)b);
    int d = StructureSLAM::ORBmatcher::DescriptorDistance(A, B);
    return d == StructureSLAM::LSDmatcher::DescriptorDistance(A, B) ? d : -1;      // the two bodies (src/ORBmatcher.cc:1650, src/LSDmatcher.cpp:364) must agree
}
int ref_features_in_area(const cv::KeyPoint* kp, int n, const float* bounds, float x, float y, float r, int minLevel, int maxLevel, int32_t* out) {
    Frame* F = new Frame(); fill_frame(*F, kp, nullptr, n, bounds);
    std::vector<size_t> v = F->GetFeaturesInArea(x, y, r, minLevel, maxLevel);
    for (size_t i = 0; i < v.size(); ++i) out[i] = (int32_t)v[i];
    delete F; return (int)v.size();
}
int ref_lines_in_area(const KeyLine* kl, int n, float x1, float y1, float x2, float y2, float r, int minLevel, int maxLevel, int32_t* out) {
    Frame* F = new Frame(); F->NL = n; F->mvKeylinesUn.assign(kl, kl + n);
    std::vector<size_t> v = F->GetLinesInArea(x1, y1, x2, y2, r, minLevel, maxLevel);
    for (size_t i = 0; i < v.size(); ++i) out[i] = (int32_t)v[i];
    delete F; return (int)v.size();
}
int ref_search_for_initialization(const cv::KeyPoint* kp1, const uint8_t* d1, int n1, const cv::KeyPoint* kp2, const uint8_t* d2, int n2,
                                  float* prevMatched, int32_t* m12, int window, float nnratio, int checkOri, const float* bounds) {
    Frame* F1 = new Frame(); Frame* F2 = new Frame();
    fill_frame(*F1, kp1, d1, n1, bounds); fill_frame(*F2, kp2, d2, n2, bounds);
    std::vector<cv::Point2f> prev(n1); for (int i = 0; i < n1; ++i) prev[i] = cv::Point2f(prevMatched[2 * i], prevMatched[2 * i + 1]);
    std::vector<int> v12;
    StructureSLAM::ORBmatcher m(nnratio, checkOri != 0);
    int n = m.SearchForInitialization(*F1, *F2, prev, v12, window);
    for (int i = 0; i < n1; ++i) { m12[i] = v12[i]; prevMatched[2 * i] = prev[i].x; prevMatched[2 * i + 1] = prev[i].y; }
    delete F1; delete F2; return n;
}
// LSDmatcher::SerachForInitialize (src/LSDmatcher.cpp:257-284) over Frame::lineDescriptorMAD (src/Frame.cc:190-215)
int ref_line_search_for_initialize(const uint8_t* l1, int n1, const uint8_t* l2, int n2, int32_t* pairs, int cap, double* nn_mad, double* nn12_mad) {
    Frame* A = new Frame(); Frame* B = new Frame();
    A->mLdesc.create(n1, 32, CV_8UC1); std::memcpy(A->mLdesc.data, l1, (size_t)n1 * 32);
    B->mLdesc.create(n2, 32, CV_8UC1); std::memcpy(B->mLdesc.data, l2, (size_t)n2 * 32);
    std::vector<std::pair<int, int> > lm;
    StructureSLAM::LSDmatcher m;
    int n = m.SerachForInitialize(*A, *B, lm);
    for (int i = 0; i < (int)lm.size() && i < cap; ++i) { pairs[2 * i] = lm[i].first; pairs[2 * i + 1] = lm[i].second; }
    {   // the MAD values the call used, recomputed through the same reference function for the report
        cv::BFMatcher bfm(cv::NORM_HAMMING, false); std::vector<std::vector<cv::DMatch> > k; bfm.knnMatch(A->mLdesc, B->mLdesc, k, 2);
        B->lineDescriptorMAD(k, *nn_mad, *nn12_mad);
    }
    delete A; delete B; return n;
}
}
