// Drop-in for the reference's include/ORBmatcher.h:40-91: the class keeps its name, constructor, constants and call syntax; the methods of
// the tracking thread's hot path are member templates over the frame / map-point types they are called with (template argument deduction
// keeps every call site of src/Tracking.cc unchanged), and their candidate search + selection runs on the GPU through the C ABI
// (sslam_orb_search_for_initialization, sslam_search_by_projection).  The templates read exactly the members the reference bodies read
// (include/Frame.h:150-200, include/MapPoint.h): mvKeysUn, mDescriptors, mnMinX.., mvScaleFactors, mvuRight, mvpMapPoints; MapPoint::
// mbTrackInView, isBad(), mnTrackScaleLevel, mTrackViewCos, mTrackProjX/Y/XR, GetDescriptor(), Observations().
//
// Provided here (src/ORBmatcher.cc bodies to delete): DescriptorDistance :1650-1666, SearchForInitialization :408-523,
// SearchByProjection(Frame&, vector<MapPoint*>&, th) :45-129 with RadiusByViewingCos :131-137, SearchByProjection(Frame&, const Frame&, th,
// bMono) :1331-1473 (the per-frame tracking call; round 3).
// Left in the reference source (relocalisation / mapping / loop-closing calls: DBoW2 FeatureVector walks and Sim3 algebra that stay on the
// host; INTEGRATION.md §3 shows the one-call bodies over sslam_shim::SearchByProjection / sslam_orb_search_by_bow / sslam_fuse_search):
// SearchByProjection(Frame&, KeyFrame*, ...) :1475-1602, SearchByProjection(KeyFrame*, Scw, ...) :293-406,
// SearchByBoW x2, SearchForTriangulation, SearchBySim3, Fuse x2.  Those stay declared exactly as in the reference header when the reference's
// own types are in scope (SSLAM_REFERENCE_TYPES, defined by including this file after Frame.h / KeyFrame.h / MapPoint.h).
#pragma once
#include <set>
#include <type_traits>
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
            e.min_level = nPredictedLevel - 1; e.max_level = nPredictedLevel; e.ur = pMP->mTrackProjXR; e.valid = 1;
            e.obs_positive = pMP->Observations() > 0;      // re-read by later queries for a keypoint this one takes (src/ORBmatcher.cc:86-88)
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
        const int n = sslam_shim::SearchByProjection(0, F.mvKeysUn, F.mDescriptors, bounds, &F.mvuRight, occupied, q, qdesc, mfNNratio, TH_HIGH, false, assigned,
                                                     sslam_shim::FrameId(F, 0));
        for (size_t i = 0; i < assigned.size(); ++i) if (assigned[i] >= 0) F.mvpMapPoints[i] = owner[assigned[i]];
        return n;
    }

    // Project MapPoints tracked in the last frame into the current frame and search matches (Tracking::TrackWithMotionModel,
    // src/Tracking.cc:1227, retried with 2*th at :1243): src/ORBmatcher.cc:1331-1473.  The projection loop stays here (poses, map points:
    // tracker state); GetFeaturesInArea, the best-candidate selection, the "keypoint already holds an observed point" rule, the stereo
    // gate and the rotation histogram run on the device (mode 1).  The second parameter is `const FrameT&`, which is what distinguishes this
    // overload from the map-point one above (a vector) exactly as in the reference header (include/ORBmatcher.h:54,58).
    template <class FrameT>
    int SearchByProjection(FrameT &CurrentFrame, const FrameT &LastFrame, const float th, const bool bMono)
    {
        typedef typename std::remove_pointer<typename std::decay<decltype(CurrentFrame.mvpMapPoints[0])>::type>::type MapPointT;
        const sslam_shim::Rt cw = sslam_shim::PoseRt(CurrentFrame.mTcw), lw = sslam_shim::PoseRt(LastFrame.mTcw);
        float twc[3], tlc[3];
        sslam_shim::MinusRtT(cw, twc);                 // twc = -Rcw.t()*tcw
        sslam_shim::RxPlusT(lw, twc, tlc);             // tlc = Rlw*twc+tlw
        const bool bForward = tlc[2] > CurrentFrame.mb && !bMono;
        const bool bBackward = -tlc[2] > CurrentFrame.mb && !bMono;
        std::vector<sslam_shim::ProjQuery> q; std::vector<MapPointT *> owner; std::vector<unsigned char> qd;
        for (int i = 0; i < LastFrame.N; i++) {
            MapPointT *pMP = LastFrame.mvpMapPoints[i];
            if (!pMP || LastFrame.mvbOutlier[i]) continue;
            const cv::Mat x3Dw = pMP->GetWorldPos();
            const float xw[3] = {x3Dw.template at<float>(0), x3Dw.template at<float>(1), x3Dw.template at<float>(2)};
            float x3Dc[3];
            sslam_shim::RxPlusT(cw, xw, x3Dc);         // x3Dc = Rcw*x3Dw+tcw
            const float xc = x3Dc[0], yc = x3Dc[1];
            const float invzc = 1.0 / x3Dc[2];
            if (invzc < 0) continue;
            const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
            const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
            if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
            if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
            const int nLastOctave = LastFrame.mvKeys[i].octave;
            sslam_shim::ProjQuery e{};
            e.u = u; e.v = v; e.radius = th * CurrentFrame.mvScaleFactors[nLastOctave];
            if (bForward) { e.min_level = nLastOctave; e.max_level = -1; }                 // GetFeaturesInArea(u, v, radius, nLastOctave)
            else if (bBackward) { e.min_level = 0; e.max_level = nLastOctave; }
            else { e.min_level = nLastOctave - 1; e.max_level = nLastOctave + 1; }
            e.angle = LastFrame.mvKeysUn[i].angle; e.ur = u - CurrentFrame.mbf * invzc; e.valid = 1; e.obs_positive = pMP->Observations() > 0;
            const cv::Mat d = pMP->GetDescriptor();
            qd.insert(qd.end(), d.ptr(0), d.ptr(0) + 32);
            q.push_back(e); owner.push_back(pMP);
        }
        if (q.empty()) return 0;
        std::vector<unsigned char> occupied(CurrentFrame.mvKeysUn.size(), 0);
        for (size_t i = 0; i < occupied.size(); ++i) occupied[i] = CurrentFrame.mvpMapPoints[i] && CurrentFrame.mvpMapPoints[i]->Observations() > 0;
        const float bounds[4] = {(float)CurrentFrame.mnMinX, (float)CurrentFrame.mnMaxX, (float)CurrentFrame.mnMinY, (float)CurrentFrame.mnMaxY};
        cv::Mat qdesc((int)q.size(), 32, CV_8U, qd.data());
        std::vector<int> assigned;
        const int n = sslam_shim::SearchByProjection(1, CurrentFrame.mvKeysUn, CurrentFrame.mDescriptors, bounds, &CurrentFrame.mvuRight, occupied, q, qdesc, mfNNratio, TH_HIGH,
                                                     mbCheckOrientation, assigned, sslam_shim::FrameId(CurrentFrame, 0));
        for (size_t i = 0; i < assigned.size(); ++i) {
            if (assigned[i] >= 0) CurrentFrame.mvpMapPoints[i] = owner[assigned[i]];
            else if (assigned[i] == -2) CurrentFrame.mvpMapPoints[i] = static_cast<MapPointT *>(NULL);      // matched, then removed by the rotation check (:1465)
        }
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
