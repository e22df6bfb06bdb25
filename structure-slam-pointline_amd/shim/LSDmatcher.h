// Drop-in for the reference's include/LSDmatcher.h:37-66: same class name, constructor and call syntax; the methods below are member
// templates over the frame / keyframe / map-line types (deduced at the call sites of src/Tracking.cc and src/LocalMapping.cc, which stay
// unchanged) and run cv::BFMatcher::knnMatch(..., 2) + Frame::lineDescriptorMAD (src/Frame.cc:190-215) + the gate on the GPU through
// sslam_line_match; the windowed search is sslam_search_by_projection (kind 1).
//
// Provided here (src/LSDmatcher.cpp bodies to delete): DescriptorDistance :364-380, SearchByProjection(KeyFrame*, Frame&, ...) :143-183,
// SearchByProjection(Frame&, vector<MapLine*>&, th) :185-255 with RadiusByViewingCos :550-556, SerachForInitialize :257-284,
// SearchByDescriptor x2 :286-362, SearchForTriangulation :382-415, SearchByProjection(Frame&, const Frame&, th, bMono) :22-141 (round 3).
// Left in the reference source (Sim3 algebra on cv::Mat stays on the host; INTEGRATION.md §3 has their one-call bodies):
// SearchByProjection(KeyFrame*, Scw, ...) :558-683, SearchBySim3 :685-929, Fuse x2.
#pragma once
#include <type_traits>
#include <utility>
#include <vector>
#include "FrontendMatchers.h"

namespace StructureSLAM
{
class LSDmatcher
{
public:
    static const int TH_HIGH = 100, TH_LOW = 50;      // src/LSDmatcher.cpp:16-17

    LSDmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return sslam_shim::DescriptorDistance(a, b); }

    // knn-2 of the keyframe's LBD descriptors against the frame's, `d1/d2 < 1/1.5` (src/LSDmatcher.cpp:143-183; Tracking.cc:1024,1234)
    template <class KeyFrameT, class FrameT, class MapLineT>
    int SearchByProjection(KeyFrameT *pKF, FrameT &currentF, std::vector<MapLineT *> &vpMapLineMatches) { return KeyFrameToFrame(pKF, currentF, vpMapLineMatches); }
    template <class KeyFrameT, class FrameT, class MapLineT>
    int SearchByDescriptor(KeyFrameT *pKF, FrameT &currentF, std::vector<MapLineT *> &vpMapLineMatches) { return KeyFrameToFrame(pKF, currentF, vpMapLineMatches); }

    // keyframe against keyframe, `d2 - d1 > 0.5 MAD12` (src/LSDmatcher.cpp:327-362).  Disambiguated from the overload above by the second
    // argument being a pointer.
    template <class KeyFrameT, class MapLineT>
    int SearchByDescriptor(KeyFrameT *pKF, KeyFrameT *pKF2, std::vector<MapLineT *> &vpMapLineMatches)
    {
        const std::vector<MapLineT *> vpMapLinesKF = pKF->GetMapLineMatches();
        const std::vector<MapLineT *> vpMapLinesKF2 = pKF2->GetMapLineMatches();
        vpMapLineMatches = std::vector<MapLineT *>(vpMapLinesKF.size(), static_cast<MapLineT *>(NULL));
        std::vector<std::pair<int, int> > pairs;
        sslam_shim::LineMatch(pKF->mLineDescriptors, pKF2->mLineDescriptors, 0.5, false, pairs);
        int nmatches = 0;
        for (size_t i = 0; i < pairs.size(); ++i) {
            MapLineT *mapLine = vpMapLinesKF2[pairs[i].second];
            if (mapLine) { vpMapLineMatches[pairs[i].first] = mapLine; nmatches++; }
        }
        return nmatches;
    }

    // local map lines against the frame (Tracking::SearchLocalLines, src/Tracking.cc:1777-1783): src/LSDmatcher.cpp:185-255
    template <class FrameT, class MapLineT>
    int SearchByProjection(FrameT &F, const std::vector<MapLineT *> &vpMapLines, const float th = 3)
    {
        const bool bFactor = th != 1.0;
        std::vector<sslam_shim::ProjQuery> q; std::vector<MapLineT *> owner; std::vector<unsigned char> qd;
        for (size_t k = 0; k < vpMapLines.size(); ++k) {
            MapLineT *pML = vpMapLines[k];
            if (!pML || pML->isBad() || !pML->mbTrackInView) continue;
            const int nPredictLevel = pML->mnTrackScaleLevel;
            float r = RadiusByViewingCos(pML->mTrackViewCos);
            if (bFactor) r *= th;
            sslam_shim::ProjQuery e{};
            e.u = pML->mTrackProjX1; e.v = pML->mTrackProjY1; e.u2 = pML->mTrackProjX2; e.v2 = pML->mTrackProjY2;
            e.radius = r * F.mvScaleFactors[nPredictLevel]; e.min_level = nPredictLevel - 1; e.max_level = nPredictLevel; e.valid = 1;
            e.obs_positive = pML->Observations() > 0;      // re-read by later queries for a line this one takes (src/LSDmatcher.cpp:221-223)
            const cv::Mat d = pML->GetDescriptor();
            qd.insert(qd.end(), d.ptr(0), d.ptr(0) + 32);
            q.push_back(e); owner.push_back(pML);
        }
        if (q.empty()) return 0;
        std::vector<unsigned char> occupied(F.mvKeylinesUn.size(), 0);
        for (size_t i = 0; i < occupied.size(); ++i) occupied[i] = F.mvpMapLines[i] && F.mvpMapLines[i]->Observations() > 0;
        cv::Mat qdesc((int)q.size(), 32, CV_8U, qd.data());
        std::vector<int> assigned;
        const int n = sslam_shim::SearchLinesByProjection(F.mvKeylinesUn, F.mLdesc, occupied, q, qdesc, mfNNratio, TH_HIGH, assigned, 0);
        for (size_t i = 0; i < assigned.size(); ++i) if (assigned[i] >= 0) F.mvpMapLines[i] = owner[assigned[i]];
        return n;
    }

    // last frame's map lines projected into the current frame (the line twin of ORBmatcher::SearchByProjection(Frame&, const Frame&, ...)):
    // src/LSDmatcher.cpp:22-141.  Projection loop here, GetLinesInArea + best / second-best + the same-level ratio test + "already holds an
    // observed line" on the device (kind 1, mode 0).  `LastFrame.mvKeys[i].octave` (the POINT list indexed by a line index, :80) is the
    // reference's own expression and is kept.
    template <class FrameT>
    int SearchByProjection(FrameT &CurrentFrame, const FrameT &LastFrame, const float th, const bool bMono)
    {
        typedef typename std::remove_pointer<typename std::decay<decltype(CurrentFrame.mvpMapLines[0])>::type>::type MapLineT;
        const sslam_shim::Rt cw = sslam_shim::PoseRt(CurrentFrame.mTcw), lw = sslam_shim::PoseRt(LastFrame.mTcw);
        float twc[3], tlc[3];
        sslam_shim::MinusRtT(cw, twc);
        sslam_shim::RxPlusT(lw, twc, tlc);
        const bool bForward = tlc[2] > CurrentFrame.mb && !bMono;
        const bool bBackward = -tlc[2] > CurrentFrame.mb && !bMono;
        std::vector<sslam_shim::ProjQuery> q; std::vector<MapLineT *> owner; std::vector<unsigned char> qd;
        for (int i = 0; i < LastFrame.NL; i++) {
            MapLineT *pML = LastFrame.mvpMapLines[i];
            if (!pML || pML->isBad() || LastFrame.mvbLineOutlier[i]) continue;
            const auto P = pML->GetWorldPos();             // Vector6d
            const float SP[3] = {(float)P(0), (float)P(1), (float)P(2)}, EP[3] = {(float)P(3), (float)P(4), (float)P(5)};
            float SPc[3], EPc[3];
            sslam_shim::RxPlusT(cw, SP, SPc);
            sslam_shim::RxPlusT(cw, EP, EPc);
            if (SPc[2] < 0.0f || EPc[2] < 0.0f) continue;
            const float invz1 = 1.0f / SPc[2];
            const float u1 = CurrentFrame.fx * SPc[0] * invz1 + CurrentFrame.cx;
            const float v1 = CurrentFrame.fy * SPc[1] * invz1 + CurrentFrame.cy;
            if (u1 < CurrentFrame.mnMinX || u1 > CurrentFrame.mnMaxX) continue;
            if (v1 < CurrentFrame.mnMinY || v1 > CurrentFrame.mnMaxY) continue;
            const float invz2 = 1.0f / EPc[2];
            const float u2 = CurrentFrame.fx * EPc[0] * invz2 + CurrentFrame.cx;
            const float v2 = CurrentFrame.fy * EPc[1] * invz2 + CurrentFrame.cy;
            if (u2 < CurrentFrame.mnMinX || u2 > CurrentFrame.mnMaxX) continue;
            if (v2 < CurrentFrame.mnMinY || v2 > CurrentFrame.mnMaxY) continue;
            const int nLastOctave = LastFrame.mvKeys[i].octave;
            sslam_shim::ProjQuery e{};
            e.u = u1; e.v = v1; e.u2 = u2; e.v2 = v2; e.radius = th * CurrentFrame.mvScaleFactors[nLastOctave];
            if (bForward) { e.min_level = nLastOctave; e.max_level = -1; }
            else if (bBackward) { e.min_level = 0; e.max_level = nLastOctave; }
            else { e.min_level = nLastOctave - 1; e.max_level = nLastOctave + 1; }
            e.valid = 1; e.obs_positive = pML->Observations() > 0;
            const cv::Mat d = pML->GetDescriptor();
            qd.insert(qd.end(), d.ptr(0), d.ptr(0) + 32);
            q.push_back(e); owner.push_back(pML);
        }
        if (q.empty()) return 0;
        std::vector<unsigned char> occupied(CurrentFrame.mvKeylinesUn.size(), 0);
        for (size_t i = 0; i < occupied.size(); ++i) occupied[i] = CurrentFrame.mvpMapLines[i] && CurrentFrame.mvpMapLines[i]->Observations() > 0;
        cv::Mat qdesc((int)q.size(), 32, CV_8U, qd.data());
        std::vector<int> assigned;
        const int n = sslam_shim::SearchLinesByProjection(CurrentFrame.mvKeylinesUn, CurrentFrame.mLdesc, occupied, q, qdesc, mfNNratio, TH_HIGH, assigned, 0);
        for (size_t i = 0; i < assigned.size(); ++i) if (assigned[i] >= 0) CurrentFrame.mvpMapLines[i] = owner[assigned[i]];
        return n;
    }

    // src/LSDmatcher.cpp:257-284 (Tracking.cc:367-368)
    template <class FrameT>
    int SerachForInitialize(FrameT &InitialFrame, FrameT &CurrentFrame, std::vector<std::pair<int, int> > &LineMatches)
    {
        return sslam_shim::LineMatch(InitialFrame.mLdesc, CurrentFrame.mLdesc, 0.5, false, LineMatches);
    }

    // src/LSDmatcher.cpp:382-415 (LocalMapping.cc:920,988): gate 0.1 MAD12; pairs whose line already has a map line are skipped
    template <class KeyFrameT>
    int SearchForTriangulation(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<std::pair<size_t, size_t> > &vMatchedPairs)
    {
        vMatchedPairs.clear();
        std::vector<std::pair<int, int> > pairs;
        sslam_shim::LineMatch(pKF1->mLineDescriptors, pKF2->mLineDescriptors, 0.1, false, pairs);
        for (size_t i = 0; i < pairs.size(); ++i) {
            if (pKF1->GetMapLine(pairs[i].first) || pKF2->GetMapLine(pairs[i].second)) continue;
            vMatchedPairs.push_back(std::make_pair((size_t)pairs[i].first, (size_t)pairs[i].second));
        }
        return (int)vMatchedPairs.size();
    }

#ifdef SSLAM_REFERENCE_TYPES      // bodies stay in the reference's src/LSDmatcher.cpp
    int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapLine *> &vpLines, std::vector<MapLine *> &vpMatched, int th);
    int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapLine *> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12, const float th);
    int Fuse(KeyFrame *pKF, const std::vector<MapLine *> &vpMapLines, const float th = 3.0);
    int Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapLine *> &vpLines, float th, std::vector<MapLine *> &vpReplaceLine);
#endif

protected:
    template <class KeyFrameT, class FrameT, class MapLineT>
    int KeyFrameToFrame(KeyFrameT *pKF, FrameT &currentF, std::vector<MapLineT *> &vpMapLineMatches)
    {
        const std::vector<MapLineT *> vpMapLinesKF = pKF->GetMapLineMatches();
        vpMapLineMatches = std::vector<MapLineT *>(currentF.NL, static_cast<MapLineT *>(NULL));
        std::vector<std::pair<int, int> > pairs;
        sslam_shim::LineMatch(pKF->mLineDescriptors, currentF.mLdesc, 0.5, true, pairs);      // ratio gate d1/d2 < 1/1.5
        int nmatches = 0;
        for (size_t i = 0; i < pairs.size(); ++i) {
            MapLineT *mapLine = vpMapLinesKF[pairs[i].first];
            if (mapLine) { vpMapLineMatches[pairs[i].second] = mapLine; nmatches++; }
        }
        return nmatches;
    }
    float RadiusByViewingCos(const float &viewCos) { return viewCos > 0.998 ? 5.0f : 8.0f; }      // src/LSDmatcher.cpp:550-556
    float mfNNratio;
    bool mbCheckOrientation;
};
}  // namespace StructureSLAM
