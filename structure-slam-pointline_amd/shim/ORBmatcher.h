// Drop-in for the reference's include/ORBmatcher.h:40-91: the class keeps its name, constructor, constants and call syntax; the methods of
// the tracking thread's hot path are member templates over the frame / map-point types they are called with (template argument deduction
// keeps every call site of src/Tracking.cc unchanged), and their candidate search + selection runs on the GPU through the C ABI
// (sslam_orb_search_for_initialization, sslam_search_by_projection).  The templates read exactly the members the reference bodies read
// (include/Frame.h:150-200, include/MapPoint.h): mvKeysUn, mDescriptors, mnMinX.., mvScaleFactors, mvuRight, mvpMapPoints; MapPoint::
// mbTrackInView, isBad(), mnTrackScaleLevel, mTrackViewCos, mTrackProjX/Y/XR, GetDescriptor(), Observations().
//
// Provided here (src/ORBmatcher.cc bodies to delete): DescriptorDistance :1650-1666, SearchForInitialization :408-523,
// SearchByProjection(Frame&, vector<MapPoint*>&, th) :45-129 with RadiusByViewingCos :131-137.
// Left in the reference source (they need cv::Mat pose algebra / DBoW2 FeatureVector walks that stay on the host; INTEGRATION.md §3 shows
// the one-call bodies over sslam_shim::SearchByProjection / sslam_orb_search_by_bow / sslam_fuse_search): SearchByProjection(Frame&, const
// Frame&, ...) :1331-1473, SearchByProjection(Frame&, KeyFrame*, ...) :1475-1602, SearchByProjection(KeyFrame*, Scw, ...) :293-406,
// SearchByBoW x2, SearchForTriangulation, SearchBySim3, Fuse x2.  Those stay declared exactly as in the reference header when the reference's
// own types are in scope (SSLAM_REFERENCE_TYPES, defined by including this file after Frame.h / KeyFrame.h / MapPoint.h).
#pragma once
#include <set>
#include <utility>
#include <vector>
#include "FrontendMatchers.h"

namespace StructureSLAM
{
class ORBmatcher
{
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    // Computes the Hamming distance between two ORB descriptors (src/ORBmatcher.cc:1650-1666)
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return sslam_shim::DescriptorDistance(a, b); }

    // Search matches between Frame keypoints and projected MapPoints (Tracking::SearchLocalPoints, src/Tracking.cc:1729-1736).
    // The visibility flags and projections were set by Frame::isInFrustum (host); one query per visible map point goes to the device, which
    // replays GetFeaturesInArea, best / second-best, the same-level ratio test and the "keypoint already holds an observed point" rule.
    template <class FrameT, class MapPointT>
    int SearchByProjection(FrameT &F, const std::vector<MapPointT *> &vpMapPoints, const float th = 3)
    {
        const bool bFactor = th != 1.0;
        std::vector<sslam_shim::ProjQuery> q; std::vector<MapPointT *> owner;
        std::vector<unsigned char> qd;
        for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {
            MapPointT *pMP = vpMapPoints[iMP];
            if (!pMP->mbTrackInView) continue;
            if (pMP->isBad()) continue;
            const int nPredictedLevel = pMP->mnTrackScaleLevel;
            float r = RadiusByViewingCos(pMP->mTrackViewCos);
            if (bFactor) r *= th;
            sslam_shim::ProjQuery e{};
            e.u = pMP->mTrackProjX; e.v = pMP->mTrackProjY; e.radius = r * F.mvScaleFactors[nPredictedLevel];
            e.min_level = nPredictedLevel - 1; e.max_level = nPredictedLevel; e.ur = pMP->mTrackProjXR; e.valid = 1; e.obs_positive = 1;
            const cv::Mat d = pMP->GetDescriptor();
            qd.insert(qd.end(), d.ptr(0), d.ptr(0) + 32);
            q.push_back(e); owner.push_back(pMP);
        }
        if (q.empty()) return 0;
        std::vector<unsigned char> occupied(F.mvKeysUn.size(), 0);
        for (size_t i = 0; i < occupied.size(); ++i) occupied[i] = F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0;
        const float bounds[4] = {(float)F.mnMinX, (float)F.mnMaxX, (float)F.mnMinY, (float)F.mnMaxY};
        cv::Mat qdesc((int)q.size(), 32, CV_8U, qd.data());
        std::vector<int> assigned;
        const int n = sslam_shim::SearchByProjection(0, F.mvKeysUn, F.mDescriptors, bounds, &F.mvuRight, occupied, q, qdesc, mfNNratio, TH_HIGH, false, assigned);
        for (size_t i = 0; i < assigned.size(); ++i) if (assigned[i] >= 0) F.mvpMapPoints[i] = owner[assigned[i]];
        return n;
    }

    // Matching for the Map Initialization (src/Tracking.cc:365-366)
    template <class FrameT>
    int SearchForInitialization(FrameT &F1, FrameT &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize = 10)
    {
        const float bounds[4] = {(float)F2.mnMinX, (float)F2.mnMaxX, (float)F2.mnMinY, (float)F2.mnMaxY};
        return sslam_shim::SearchForInitialization(F1.mvKeysUn, F1.mDescriptors, F2.mvKeysUn, F2.mDescriptors, bounds, vbPrevMatched, vnMatches12, windowSize,
                                                   mfNNratio, mbCheckOrientation);
    }

#ifdef SSLAM_REFERENCE_TYPES      // declarations of the methods whose bodies stay in the reference's src/ORBmatcher.cc
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono);
    int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th, const int ORBdist);
    int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched, int th);
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches);
    int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12);
    int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo);
    int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12, const float th);
    int Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th = 3.0);
    int Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint);
#endif

public:
    static const int TH_LOW = 50;        // src/ORBmatcher.cc:35-37
    static const int TH_HIGH = 100;
    static const int HISTO_LENGTH = 30;

protected:
    float RadiusByViewingCos(const float &viewCos) { return viewCos > 0.998 ? 2.5f : 4.0f; }      // src/ORBmatcher.cc:131-137
#ifdef SSLAM_REFERENCE_TYPES
    bool CheckDistEpipolarLine(const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const cv::Mat &F12, const KeyFrame *pKF);
    void ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3);
#endif
    float mfNNratio;
    bool mbCheckOrientation;
};
}  // namespace StructureSLAM
