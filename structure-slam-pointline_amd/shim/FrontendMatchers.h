// Device-backed bodies for the reference's hot matcher entry points.  The reference methods take
// Frame/KeyFrame objects (tracker state that stays on the host); each function below takes exactly
// the members that method reads, so the body of the reference method becomes one call
// (INTEGRATION.md shows the patch).
#pragma once
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

// cv::BFMatcher(NORM_HAMMING,false).knnMatch(q,t,m,2) as used by every LSDmatcher entry point.
void KnnMatch2(const cv::Mat &query, const cv::Mat &train, std::vector<int> &idx /*nq*2*/, std::vector<int> &dist /*nq*2*/);

// LSDmatcher::SerachForInitialize / SearchForTriangulation (gate_scale 0.5 / 0.1) and, with ratioMode,
// SearchByProjection(KF,F) / SearchByDescriptor (src/LSDmatcher.cpp:143-183,257-362,382-415).
int LineMatch(const cv::Mat &ldesc1, const cv::Mat &ldesc2, double gateScale, bool ratioMode, std::vector<std::pair<int,int> > &matches);
}
