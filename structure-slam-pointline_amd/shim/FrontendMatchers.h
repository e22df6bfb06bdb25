// Device-backed bodies for the reference's hot matcher entry points.  The reference methods take
// Frame/KeyFrame objects (tracker state that stays on the host); each function below takes exactly
// the members that method reads, so the body of the reference method becomes one call
// (INTEGRATION.md shows the patch).
#pragma once
#include <map>
#include <string>
#include <utility>
#include <vector>
#include "cv_min.h"

namespace sslam_shim
{
// The pose algebra of the two per-frame tracking calls (ORBmatcher / LSDmatcher ::SearchByProjection(Frame&, const Frame&, th, bMono),
// src/ORBmatcher.cc:1340-1349,1363-1364, src/LSDmatcher.cpp:25-34,50-59): 3x3 * 3x1 products on CV_32F matrices.  With OpenCV present these
// are the reference's own cv::Mat expressions (same gemm, same roundings); with the stand-in Mat (cv_min.h) plain float arithmetic in
// row order -- OpenCV's small-matrix path accumulates a 3-term row in float as well, but that is upstream behaviour this build cannot pin.
struct Rt { float R[9], t[3]; };
inline Rt PoseRt(const cv::Mat &Tcw) {          // Rcw = Tcw.rowRange(0,3).colRange(0,3), tcw = Tcw.rowRange(0,3).col(3)
    Rt p;
    for (int r = 0; r < 3; ++r) { const float *row = Tcw.ptr<float>(r); p.R[3 * r] = row[0]; p.R[3 * r + 1] = row[1]; p.R[3 * r + 2] = row[2]; p.t[r] = row[3]; }
    return p;
}
inline void RxPlusT(const Rt &p, const float x[3], float out[3]) {      // R * x + t
#ifdef SSLAM_HAVE_OPENCV
    const cv::Mat R(3, 3, CV_32F, (void *)p.R), t(3, 1, CV_32F, (void *)p.t), xv(3, 1, CV_32F, (void *)x);
    const cv::Mat o = R * xv + t;
    for (int r = 0; r < 3; ++r) out[r] = o.at<float>(r);
#else
    for (int r = 0; r < 3; ++r) out[r] = ((p.R[3 * r] * x[0] + p.R[3 * r + 1] * x[1]) + p.R[3 * r + 2] * x[2]) + p.t[r];
#endif
}
inline void MinusRtT(const Rt &p, float out[3]) {                      // -R.t() * t
#ifdef SSLAM_HAVE_OPENCV
    const cv::Mat R(3, 3, CV_32F, (void *)p.R), t(3, 1, CV_32F, (void *)p.t);
    const cv::Mat o = -R.t() * t;
    for (int r = 0; r < 3; ++r) out[r] = o.at<float>(r);
#else
    for (int r = 0; r < 3; ++r) out[r] = ((-p.R[r]) * p.t[0] + (-p.R[3 + r]) * p.t[1]) + (-p.R[6 + r]) * p.t[2];
#endif
}

// ORBmatcher::DescriptorDistance / LSDmatcher::DescriptorDistance (src/ORBmatcher.cc:1650-1666): host popcount, unchanged semantics.
int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);

// ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:408-523).  bounds = Frame::mnMinX,mnMaxX,mnMinY,mnMaxY.
int SearchForInitialization(const std::vector<cv::KeyPoint> &keysUn1, const cv::Mat &desc1,
                            const std::vector<cv::KeyPoint> &keysUn2, const cv::Mat &desc2, const float bounds[4],
                            std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12,
                            int windowSize, float nnratio, bool checkOri);

// The projection-window family: ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th) (src/ORBmatcher.cc:45-129, mode 0),
// ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) (:1331-1473, mode 1) and, with keylines,
// LSDmatcher::SearchByProjection(Frame&, vector<MapLine*>&, th) / (Frame&, const Frame&, ...) (src/LSDmatcher.cpp:185-255, 22-141).
// The caller keeps the projection / visibility loop and fills one sslam_proj_query per map point (include/sslam_frontend.h);
// assigned[i] >= 0 means F.mvpMapPoints[i] = map point of query assigned[i].
struct ProjQuery { float u, v, u2, v2, radius; int min_level, max_level; float angle, ur; int valid, obs_positive; };
int SearchByProjection(int mode, const std::vector<cv::KeyPoint> &keysUn, const cv::Mat &desc, const float bounds[4],
                       const std::vector<float> *uRight, const std::vector<unsigned char> &occupied,
                       const std::vector<ProjQuery> &queries, const cv::Mat &queryDesc, float nnratio, int thDist, bool checkOri,
                       std::vector<int> &assigned, long frameId = -1);
// frameId >= 0 (Frame::mnId, include/Frame.h:128): the frame's features stay on the device between calls (a small cache in the shim), so the
// retry of TrackWithMotionModel (src/Tracking.cc:1243) and SearchLocalPoints (:1736) on the same frame upload nothing but their queries.
// The cache keys on (mnId, keypoint count, a fingerprint of a few keypoints and descriptor rows): ids repeat after Tracking::Reset()
// (src/Tracking.cc:2150).  InvalidateResidentFrames() drops every resident frame -- one line for Tracking::Reset.
void InvalidateResidentFrames();
template <class T> auto FrameId(const T &f, int) -> decltype((long)f.mnId) { return (long)f.mnId; }
template <class T> long FrameId(const T &, ...) { return -1; }
int SearchLinesByProjection(const std::vector<cv::line_descriptor::KeyLine> &keylinesUn, const cv::Mat &ldesc,
                            const std::vector<unsigned char> &occupied, const std::vector<ProjQuery> &queries, const cv::Mat &queryDesc,
                            float nnratio, int thDist, std::vector<int> &assigned, int mode = 0);
// mode 1 (both functions): best candidate only and a feature that already holds a map point / line is skipped -- with the argument
// mapping of include/sslam_frontend.h this is also ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)
// (src/ORBmatcher.cc:1475-1602) and the loop-closing SearchByProjection(KeyFrame*, Scw, ...) overloads (:293-406, src/LSDmatcher.cpp:558-683).

// cv::BFMatcher(NORM_HAMMING,false).knnMatch(q,t,m,2) as used by every LSDmatcher entry point.
void KnnMatch2(const cv::Mat &query, const cv::Mat &train, std::vector<int> &idx /*nq*2*/, std::vector<int> &dist /*nq*2*/);

// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:276-306) / MapLine::ComputeDistinctiveDescriptors
// (src/MapLine.cpp:280-311): index of the observed descriptor with the least median Hamming distance to the others
// (replaces the N x N loop + per-row sort; `BestIdx = sslam_shim::DistinctiveIndex(vDescriptors);`).
int DistinctiveIndex(const std::vector<cv::Mat> &vDescriptors);

// ORBVocabulary (include/ORBVocabulary.h:31-32 = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>): the two members the
// reference calls on the hot path -- loadFromTextFile (src/System.cc:64-73) and transform(features, BowVector&, FeatureVector&,
// levelsup) (Frame::ComputeBoW src/Frame.cc:474-481, KeyFrame::ComputeBoW src/KeyFrame.cc:71-80).  BowVector / FeatureVector keep
// DBoW2's container types (std::map<WordId, WordValue>, std::map<NodeId, std::vector<unsigned> >), so
// `mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4);` compiles unchanged against this class.
typedef std::map<unsigned int, double> BowVector;
typedef std::map<unsigned int, std::vector<unsigned int> > FeatureVector;
class ORBVocabulary
{
public:
    ORBVocabulary();
    ~ORBVocabulary();
    bool loadFromTextFile(const std::string &filename);
    bool empty() const;
    unsigned int size() const;                  // number of words
    void transform(const std::vector<cv::Mat> &features, BowVector &v, FeatureVector &fv, int levelsup) const;
private:
    ORBVocabulary(const ORBVocabulary &); ORBVocabulary &operator=(const ORBVocabulary &);
    void *mHandle;
};

// LSDmatcher::SerachForInitialize / SearchForTriangulation (gate_scale 0.5 / 0.1) and, with ratioMode,
// SearchByProjection(KF,F) / SearchByDescriptor (src/LSDmatcher.cpp:143-183,257-362,382-415).
int LineMatch(const cv::Mat &ldesc1, const cv::Mat &ldesc2, double gateScale, bool ratioMode, std::vector<std::pair<int,int> > &matches);
}
