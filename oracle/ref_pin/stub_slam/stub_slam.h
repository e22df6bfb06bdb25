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
#include <set>
#include <map>
#include <mutex>
#include <iostream>
#include "../stub_cv/stub_cv.hpp"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"      // the reference's own vendored DBoW2 header, where it lies (-I$(REF)); FeatureVector.cpp is compiled beside the slices

namespace cv {
namespace line_descriptor {
struct KeyLine {            // opencv_contrib line_descriptor/descriptor.hpp: 68 bytes
    float angle; int class_id; int octave; Point2f pt; float response; float size;
    float startPointX, startPointY, endPointX, endPointY, sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY, lineLength;
    int numOfPixels;
};
static_assert(sizeof(KeyLine) == 68, "KeyLine is 68 bytes");
}  // namespace line_descriptor

// cv::Ptr, LSDDetector::createLSDDetector()->detect(img, keylines, scale, numOctaves), BinaryDescriptor::createBinaryDescriptor()->compute(img,
// keylines, descriptors): the two opencv_contrib classes LineSegment::ExtractLineSegment drives (src/ExtractLineSegment.cpp:38-40, 53).  Leaves:
// they forward to the oracle's restatements (oracle/lsd_oracle.cpp, lbd_oracle.cpp, UPSTREAM-RECALL) through liboracle.so.
}  // namespace cv
#include "../../lines_types.h"
namespace cv {
template <class T> struct Ptr { std::shared_ptr<T> p; T* operator->() const { return p.get(); } };
namespace line_descriptor {
static_assert(sizeof(KeyLine) == sizeof(orc::KeyLine), "KeyLine layouts");
struct LSDDetector {
    static Ptr<LSDDetector> createLSDDetector() { return Ptr<LSDDetector>{std::make_shared<LSDDetector>()}; }
    void detect(const Mat& image, std::vector<KeyLine>& keylines, int scale, int numOctaves, const Mat& = Mat()) {
        assert(scale == 1 && numOctaves == 1);      // what `int scale = 1.2` truncates to (include/ExtractLineSegment.h:62)
        std::vector<orc::KeyLine> k; orc::lsd_detect_keylines(stubdetail::toImg(image), k, nullptr);
        keylines.resize(k.size()); if (!k.empty()) std::memcpy(keylines.data(), k.data(), sizeof(KeyLine) * k.size());
    }
};
struct BinaryDescriptor {
    static Ptr<BinaryDescriptor> createBinaryDescriptor() { return Ptr<BinaryDescriptor>{std::make_shared<BinaryDescriptor>()}; }
    void compute(const Mat& image, std::vector<KeyLine>& keylines, Mat& descriptors, bool = false) const {
        std::vector<orc::KeyLine> k(keylines.size()); if (!k.empty()) std::memcpy(k.data(), keylines.data(), sizeof(KeyLine) * k.size());
        std::vector<uint8_t> d; orc::lbd_compute(stubdetail::toImg(image), k, d, nullptr);
        descriptors.create((int)k.size(), 32, CV_8UC1);
        for (size_t i = 0; i < k.size(); ++i) std::memcpy(descriptors.ptr((int)i), &d[i * 32], 32);
    }
};
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

// auxiliar.h:42 `typedef Matrix<double,6,1> Vector6d;` (Eigen, un-vendored): the slices only read P(0) .. P(5)
struct Vector6d { double v[6]; double operator()(int i) const { return v[i]; } };
// Eigen::Vector3d as src/ExtractLineSegment.cpp:56-68 uses it (Eigen is not vendored either): `v << a, b, c`, cross, v / s, v(i); all in double
namespace Eigen {
struct Vector3d {
    double v[3];
    struct Comma { Vector3d* t; int i; Comma& operator,(double x) { t->v[i++] = x; return *this; } };
    Comma operator<<(double x) { v[0] = x; return Comma{this, 1}; }
    Vector3d& operator<<(const Vector3d& o) { *this = o; return *this; }
    Vector3d cross(const Vector3d& o) const { Vector3d r; r.v[0] = v[1] * o.v[2] - v[2] * o.v[1]; r.v[1] = v[2] * o.v[0] - v[0] * o.v[2]; r.v[2] = v[0] * o.v[1] - v[1] * o.v[0]; return r; }
    Vector3d operator/(double s) const { Vector3d r; r.v[0] = v[0] / s; r.v[1] = v[1] / s; r.v[2] = v[2] / s; return r; }
    double operator()(int i) const { return v[i]; }
};
}
using namespace Eigen;

namespace StructureSLAM {
#ifndef FRAME_GRID_ROWS
#define FRAME_GRID_ROWS 48      // include/Frame.h:45-46 (the build greps the two defines and fails if they ever change)
#define FRAME_GRID_COLS 64
#endif

// include/MapPoint.h / include/MapLine.h: what the tracking matchers read of a map point / map line (names and types as in the reference;
// the tracking fields are filled by Frame::isInFrustum in the real system, by the test driver here)
class KeyFrame; class Frame;
extern std::mutex gStubMutex;       // one mutex behind every stand-in's mMutexFeatures (the stand-ins stay movable)
class MapPoint {
public:
    // MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:247-312) reads these
    void ComputeDistinctiveDescriptors();
    std::mutex& mMutexFeatures = gStubMutex; bool mbBad = false; std::map<KeyFrame*, size_t> mObservations; cv::Mat mDescriptor;
    bool mbTrackInView = false; int mnTrackScaleLevel = 0; float mTrackViewCos = 0, mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0;
    bool isBad();                                     // ref_slices_api.cpp (returns `bad`; the Fuse harness notes which map point the loop is at)
    cv::Mat GetDescriptor() { return desc.clone(); }
    cv::Mat GetWorldPos() { return worldPos.clone(); }
    int Observations() { return nObs; }
    bool bad = false; int nObs = 1; cv::Mat desc, worldPos;      // (stand-in state)
    // what ORBmatcher::Fuse reads and does (include/MapPoint.h); PredictScale / Get*DistanceInvariance bodies from src/MapPoint.cc:378-405
    std::mutex& mMutexPos = gStubMutex; float mfMinDistance = 0, mfMaxDistance = 0; cv::Mat normal;
    bool IsInKeyFrame(KeyFrame*) { return inKF; } bool inKF = false;
    cv::Mat GetNormal() { return normal.clone(); }
    float GetMinDistanceInvariance(); float GetMaxDistanceInvariance();
    int PredictScale(const float& currentDist, KeyFrame* pKF);
    int PredictScale(const float& currentDist, Frame* pF);      // src/MapPoint.cc:407-422
    int GetIndexInKeyFrame(KeyFrame*) { return idxInKF2; } int idxInKF2 = -1;      // (SearchBySim3 asks for the index in pKF2 only)
    void Replace(MapPoint* pMP);                      // ref_slices_api.cpp: recorded, the keyframe's slot re-pointed as src/MapPoint.cc:180-230 does
    void AddObservation(KeyFrame* pKF, size_t idx);   // recorded
};
class MapLine {
public:
    void ComputeDistinctiveDescriptors();      // src/MapLine.cpp:246-317
    std::mutex& mMutexFeatures = gStubMutex; bool mbBad = false; std::map<KeyFrame*, size_t> mObservations; cv::Mat mLDescriptor;
    bool mbTrackInView = false; int mnTrackScaleLevel = 0; float mTrackViewCos = 0, mTrackProjX1 = 0, mTrackProjY1 = 0, mTrackProjX2 = 0, mTrackProjY2 = 0;
    bool isBad();                                     // ref_slices_api.cpp (as MapPoint::isBad)
    cv::Mat GetDescriptor() { return desc.clone(); }
    Vector6d GetWorldPos() { return worldPos; }
    int Observations() { return nObs; }
    bool bad = false; int nObs = 1; cv::Mat desc; Vector6d worldPos;
    // what LSDmatcher::Fuse reads and does (include/MapLine.h); Get*DistanceInvariance / PredictScale bodies from src/MapLine.cpp:374-394
    std::mutex& mMutexPos = gStubMutex; float mfMinDistance = 0, mfMaxDistance = 0; Eigen::Vector3d normal;
    Eigen::Vector3d GetNormal() { return normal; }
    float GetMinDistanceInvariance(); float GetMaxDistanceInvariance();
    int PredictScale(const float& currentDist, const float& logScaleFactor);
    void Replace(MapLine* pML);                       // ref_slices_api.cpp (recorded)
    int GetIndexInKeyFrame(KeyFrame*) { return idxInKF2; } int idxInKF2 = -1;      // (SearchBySim3 asks for the index in pKF2 only)
    void AddObservation(KeyFrame* pKF, size_t idx);
};

class Frame {
public:
    Frame() : N(0), NL(0) {}
    DBoW2::FeatureVector mFeatVec;
    // what the tracking matchers read besides the features (include/Frame.h:97-189)
    cv::Mat mTcw; float mb = 0, mbf = 0, fx = 0, fy = 0, cx = 0, cy = 0; int mnScaleLevels = 8; float mfLogScaleFactor = 0;      // (the last two: MapPoint::PredictScale(dist, Frame*))
    std::vector<cv::KeyPoint> mvKeys; std::vector<float> mvuRight, mvScaleFactors;
    std::vector<MapPoint*> mvpMapPoints; std::vector<bool> mvbOutlier;
    std::vector<MapLine*> mvpMapLines; std::vector<bool> mvbLineOutlier;
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

// include/KeyFrame.h: what SearchByBoW reads of a keyframe
class KeyFrame {
public:
    bool isBad() { return bad; } bool bad = false;
    std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
    std::vector<MapLine*> GetMapLineMatches() { return mvpMapLines; }
    MapLine* GetMapLine(const size_t& idx);           // ref_slices_api.cpp (as GetMapPoint)
    void AddMapLine(MapLine* pML, const size_t& idx) { mvpMapLines[idx] = pML; }
    std::set<MapLine*> GetMapLines();                 // ref_slices_api.cpp: the good map lines of the keyframe (src/KeyFrame.cc:737-750)
    std::vector<KeyLine> mvKeyLines;                  // (include/KeyFrame.h:201; mLineDescriptors is declared below)
    std::vector<size_t> GetLinesInArea(const float& x1, const float& y1, const float& x2, const float& y2, const float& r, const int minLevel = -1, const int maxLevel = -1) const;      // src/KeyFrame.cc:651-684
    void lineDescriptorMAD(std::vector<std::vector<cv::DMatch> > line_matches, double& nn_mad, double& nn12_mad) const;      // src/KeyFrame.cc:820-845
    cv::Mat mLineDescriptors; std::vector<MapLine*> mvpMapLines;
    // what ORBmatcher::SearchForTriangulation / CheckDistEpipolarLine read (include/KeyFrame.h)
    MapPoint* GetMapPoint(const size_t& idx);         // ref_slices_api.cpp (returns mvpMapPoints[idx]; the Fuse harness notes the index)
    int lastQueried = -1;
    cv::Mat GetCameraCenter() { return Ow.clone(); }
    cv::Mat GetRotation() { return Tcw.rowRange(0, 3).colRange(0, 3).clone(); }
    cv::Mat GetTranslation() { return Tcw.rowRange(0, 3).col(3).clone(); }
    int N = 0; float fx = 0, fy = 0, cx = 0, cy = 0;
    std::vector<float> mvuRight, mvScaleFactors, mvLevelSigma2;
    cv::Mat Tcw, Ow;      // (stand-in state)
    DBoW2::FeatureVector mFeatVec; cv::Mat mDescriptors; std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<MapPoint*> mvpMapPoints;      // (stand-in state)
    // what ORBmatcher::Fuse reads (include/KeyFrame.h:150-250); GetFeaturesInArea / IsInImage bodies from src/KeyFrame.cc:610-649, 686-689
    float mbf = 0; int mnGridCols = FRAME_GRID_COLS, mnGridRows = FRAME_GRID_ROWS; float mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
    int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;      // (int in the reference's KeyFrame, include/KeyFrame.h:225-228)
    int mnScaleLevels = 8; float mfLogScaleFactor = 0; std::vector<float> mvInvLevelSigma2;
    std::vector<std::vector<std::vector<size_t> > > mGrid;
    std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r) const;
    bool IsInImage(const float& x, const float& y) const;
    void AddMapPoint(MapPoint* pMP, const size_t& idx) { mvpMapPoints[idx] = pMP; }
    std::set<MapPoint*> GetMapPoints();               // ref_slices_api.cpp: the good map points of the keyframe (src/KeyFrame.cc:300-313)
};

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10);
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3);
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs, const bool bOnlyStereo);
    bool CheckDistEpipolarLine(const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const cv::Mat& F12, const KeyFrame* pKF);
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0);
    int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint);      // loop closing, :980-1103
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12, const cv::Mat& t12, const float th);      // :1105-1329
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist);      // relocalisation, :1475-1602
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th);      // loop closing, :293-406
    float RadiusByViewingCos(const float& viewCos);
    static const int TH_LOW, TH_HIGH, HISTO_LENGTH;
    void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);
    float mfNNratio; bool mbCheckOrientation;
};

class LineSegment {       // include/ExtractLineSegment.h:51-78
public:
    LineSegment();
    void ExtractLineSegment(const Mat& img, vector<KeyLine>& keylines, Mat& ldesc, vector<Vector3d>& keylineFunctions, int scale = 1.2, int numOctaves = 1);
};

class LSDmatcher {
public:
    static const int TH_HIGH, TH_LOW;
    LSDmatcher(float nnratio = 0.6, bool checkOri = true);
    static int DescriptorDistance(const Mat& a, const Mat& b);
    int SerachForInitialize(Frame& InitialFrame, Frame& CurrentFrame, vector<pair<int, int> >& LineMatches);
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);
    int SearchByProjection(Frame& F, const std::vector<MapLine*>& vpMapLines, const float th = 3);
    int SearchByProjection(KeyFrame* pKF, Frame& currentF, vector<MapLine*>& vpMapLineMatches);
    int SearchByDescriptor(KeyFrame* pKF, Frame& currentF, std::vector<MapLine*>& vpMapLineMatches);
    int SearchByDescriptor(KeyFrame* pKF, KeyFrame* pKF2, std::vector<MapLine*>& vpMapLineMatches);
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<std::pair<size_t, size_t> >& vMatchedPairs);
    int Fuse(KeyFrame* pKF, const std::vector<MapLine*>& vpMapLines, const float th = 3.0);      // src/LSDmatcher.cpp:417-548
    int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapLine*>& vpLines, float th, std::vector<MapLine*>& vpReplaceLine);      // :931-1063
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapLine*>& vpLines, std::vector<MapLine*>& vpMatched, int th);      // :558-683
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapLine*>& vpMatches12, const float& s12, const cv::Mat& R12, const cv::Mat& t12, const float th);      // :685-929
    float RadiusByViewingCos(const float& viewCos);
    float mfNNratio; bool mbCheckOrientation;
};
}  // namespace StructureSLAM
