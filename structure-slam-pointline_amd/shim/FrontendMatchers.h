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
                       std::vector<int> &assigned);
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
