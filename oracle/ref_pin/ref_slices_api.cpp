// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points around the reference functions that `make -C oracle/ref_pin pin-stub` cuts out of
// src/ORBmatcher.cc, src/LSDmatcher.cpp and src/Frame.cc (this file is appended to the generated oracle/_ref/ref_slices.cc; it holds no
// reference code).  compare_stub.py calls them through ctypes next to the oracle's orc_* functions on the same arrays.
std::mutex StructureSLAM::gStubMutex;
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

// ---- the tracking thread's projection matchers (round 4): src/ORBmatcher.cc:45-137, 1331-1473; src/LSDmatcher.cpp:22-141, 185-255 ----
// state[i] of a frame feature: 0 = holds nothing, 1 = holds a map point / line with observations, 2 = holds one without
// (src/ORBmatcher.cc:89-91: only the first kind is skipped).  assigned[i]: >= 0 the map point / line matched to feature i, -1 NULL after the
// call, -2 / -3 still the occupant it had before (with / without observations).
struct MpIn { int inView, bad, level, nObs; float viewCos, projX, projY, projXR; };
struct MlIn { int inView, bad, level, nObs; float viewCos, x1, y1, x2, y2; };
static void set_frame_common(Frame& F, const cv::KeyPoint* kp, const uint8_t* desc, int n, const float* bounds, const float* scale8, const float* uright) {
    fill_frame(F, kp, desc, n, bounds);
    F.mvKeys.assign(kp, kp + n); F.mvScaleFactors.assign(scale8, scale8 + 8); F.mvuRight.assign(n, -1.f);
    if (uright) F.mvuRight.assign(uright, uright + n);
}
template <class T> static void encode(const std::vector<T*>& now, const std::vector<T>& pool, const std::vector<T>& occ, int32_t* assigned) {
    for (size_t i = 0; i < now.size(); ++i) {
        const T* p = now[i];
        assigned[i] = !p ? -1 : p == &occ[0] ? -2 : p == &occ[1] ? -3 : (int32_t)(p - pool.data());
    }
}
// ORBmatcher::Fuse(KeyFrame*, vector<MapPoint*>, th) (src/ORBmatcher.cc:828-977): what it does to the map is recorded per map point
struct FuseMp { float wp[3], nrm[3]; float minDist, maxDist; int nObs, bad, inKF; };
struct FuseLog { StructureSLAM::KeyFrame* kf = nullptr; StructureSLAM::MapPoint* pool = nullptr; int npool = 0; StructureSLAM::MapPoint* occ = nullptr; int nocc = 0;
                 std::vector<int> idx, act; int cur = -1; };
static FuseLog gFuse;
// index of p in the harness's pool of map points, -1 for any other object (integer arithmetic: p may point into another array)
static long fuse_pool_index(const StructureSLAM::MapPoint* p) {
    const uintptr_t a = (uintptr_t)p, b = (uintptr_t)gFuse.pool, e = b + (uintptr_t)gFuse.npool * sizeof(StructureSLAM::MapPoint);
    return gFuse.pool && a >= b && a < e ? (long)((a - b) / sizeof(StructureSLAM::MapPoint)) : -1;
}
namespace StructureSLAM {
bool MapPoint::isBad() {                                      // Fuse asks every map point of its list first (:850): that is the point the loop is at
    const long i = fuse_pool_index(this);
    if (i >= 0) gFuse.cur = (int)i;
    return bad;
}
MapPoint* KeyFrame::GetMapPoint(const size_t& idx) {          // Fuse: only behind "bestDist <= TH_LOW" (:950) -- the feature the current map point is fused to
    lastQueried = (int)idx;
    if (gFuse.kf == this && gFuse.cur >= 0) { gFuse.idx[gFuse.cur] = (int)idx; if (!gFuse.act[gFuse.cur]) gFuse.act[gFuse.cur] = 4; }      // 4: the slot holds a bad point (counted, nothing done)
    return mvpMapPoints[idx];
}
std::set<MapPoint*> KeyFrame::GetMapPoints() {              // src/KeyFrame.cc:300-313: every non-NULL, non-bad entry of mvpMapPoints
    std::set<MapPoint*> s;
    for (size_t i = 0; i < mvpMapPoints.size(); ++i) if (mvpMapPoints[i] && !mvpMapPoints[i]->bad) s.insert(mvpMapPoints[i]);
    return s;
}
void MapPoint::AddObservation(KeyFrame*, size_t idx) {      // "pMP->AddObservation(pKF, bestIdx)": a new measurement of pMP
    const long i = fuse_pool_index(this);
    if (i >= 0) { gFuse.idx[i] = (int)idx; gFuse.act[i] = 1; }
}
void MapPoint::Replace(MapPoint* pMP) {
    const int at = gFuse.kf->lastQueried;                    // pKF->GetMapPoint(bestIdx) came right before
    const long i = fuse_pool_index(this), j = fuse_pool_index(pMP);
    if (i >= 0 && gFuse.cur == i) { gFuse.idx[i] = at; gFuse.act[i] = 2; }              // pMP->Replace(pMPinKF): the keyframe's point survives
    else if (j >= 0) { gFuse.idx[j] = at; gFuse.act[j] = 3; gFuse.kf->mvpMapPoints[at] = pMP; }      // pMPinKF->Replace(pMP): the slot now holds pMP
    bad = true;                                              // src/MapPoint.cc:196 mbBad = true
}
}
static void fill_keyframe(StructureSLAM::KeyFrame& K, StructureSLAM::Frame& F, const cv::KeyPoint* kp, const uint8_t* desc, int n, const float* bounds, const float* scale8,
                          const float* invSigma2_8, float logScaleFactor, const float* uright, const float* cam, const float* Tcw, const float* Ow) {
    fill_frame(F, kp, desc, n, bounds);                      // Frame::AssignFeaturesToGrid (reference body); KeyFrame::KeyFrame copies the grid (src/KeyFrame.cc:45-50)
    K.N = n; K.mvKeysUn.assign(kp, kp + n); K.mDescriptors = F.mDescriptors.clone();
    K.mvuRight.assign(n, -1.f); if (uright) K.mvuRight.assign(uright, uright + n);
    K.mvScaleFactors.assign(scale8, scale8 + 8); K.mvInvLevelSigma2.assign(invSigma2_8, invSigma2_8 + 8); K.mfLogScaleFactor = logScaleFactor; K.mnScaleLevels = 8;
    K.fx = cam[0]; K.fy = cam[1]; K.cx = cam[2]; K.cy = cam[3]; K.mbf = cam[4];
    K.mnMinX = (int)bounds[0]; K.mnMaxX = (int)bounds[1]; K.mnMinY = (int)bounds[2]; K.mnMaxY = (int)bounds[3];      // const int members initialised from Frame's floats (include/KeyFrame.h:221-228)
    K.mfGridElementWidthInv = StructureSLAM::Frame::mfGridElementWidthInv; K.mfGridElementHeightInv = StructureSLAM::Frame::mfGridElementHeightInv;
    K.mGrid.assign(FRAME_GRID_COLS, std::vector<std::vector<size_t> >(FRAME_GRID_ROWS));
    for (int i = 0; i < FRAME_GRID_COLS; ++i) for (int j = 0; j < FRAME_GRID_ROWS; ++j) K.mGrid[i][j] = F.mGrid[i][j];
    K.Tcw = cv::Mat(4, 4, CV_32F); std::memcpy(K.Tcw.data, Tcw, 64); K.Ow = cv::Mat(3, 1, CV_32F); std::memcpy(K.Ow.data, Ow, 12);
}
static void fill_fuse_point(StructureSLAM::MapPoint& q, const FuseMp& m, const uint8_t* d) {
    q.worldPos = cv::Mat(3, 1, CV_32F); std::memcpy(q.worldPos.data, m.wp, 12); q.normal = cv::Mat(3, 1, CV_32F); std::memcpy(q.normal.data, m.nrm, 12);
    q.mfMinDistance = m.minDist; q.mfMaxDistance = m.maxDist; q.nObs = m.nObs; q.bad = m.bad != 0; q.inKF = m.inKF != 0;
    q.desc = cv::Mat(1, 32, CV_8UC1, (void*)d);
}
// LSDmatcher::Fuse(KeyFrame*, vector<MapLine*>, th) (src/LSDmatcher.cpp:417-548): the same bookkeeping for map lines
struct FuseMl { double wp[6], nrm[3]; float minDist, maxDist; int nObs, bad; };
struct FuseLogL { StructureSLAM::KeyFrame* kf = nullptr; StructureSLAM::MapLine* pool = nullptr; int npool = 0; std::vector<int> idx, act; int cur = -1; };
static FuseLogL gFuseL;
static long fuse_line_index(const StructureSLAM::MapLine* p) {
    const uintptr_t a = (uintptr_t)p, b = (uintptr_t)gFuseL.pool, e = b + (uintptr_t)gFuseL.npool * sizeof(StructureSLAM::MapLine);
    return gFuseL.pool && a >= b && a < e ? (long)((a - b) / sizeof(StructureSLAM::MapLine)) : -1;
}
namespace StructureSLAM {
bool MapLine::isBad() { const long i = fuse_line_index(this); if (i >= 0) gFuseL.cur = (int)i; return bad; }
MapLine* KeyFrame::GetMapLine(const size_t& idx) {
    lastQueried = (int)idx;
    if (gFuseL.kf == this && gFuseL.cur >= 0) { gFuseL.idx[gFuseL.cur] = (int)idx; if (!gFuseL.act[gFuseL.cur]) gFuseL.act[gFuseL.cur] = 4; }
    return mvpMapLines[idx];
}
std::set<MapLine*> KeyFrame::GetMapLines() {
    std::set<MapLine*> s;
    for (size_t i = 0; i < mvpMapLines.size(); ++i) if (mvpMapLines[i] && !mvpMapLines[i]->bad) s.insert(mvpMapLines[i]);
    return s;
}
void MapLine::AddObservation(KeyFrame*, size_t idx) { const long i = fuse_line_index(this); if (i >= 0) { gFuseL.idx[i] = (int)idx; gFuseL.act[i] = 1; } }
void MapLine::Replace(MapLine* pML) {
    const int at = gFuseL.kf->lastQueried;
    const long i = fuse_line_index(this), j = fuse_line_index(pML);
    if (i >= 0 && gFuseL.cur == i) { gFuseL.idx[i] = at; gFuseL.act[i] = 2; }
    else if (j >= 0) { gFuseL.idx[j] = at; gFuseL.act[j] = 3; gFuseL.kf->mvpMapLines[at] = pML; }
    bad = true;
}
}
static void fill_fuse_line(StructureSLAM::MapLine& q, const FuseMl& m, const uint8_t* d) {
    for (int k = 0; k < 6; ++k) q.worldPos.v[k] = m.wp[k];
    for (int k = 0; k < 3; ++k) q.normal.v[k] = m.nrm[k];
    q.mfMinDistance = m.minDist; q.mfMaxDistance = m.maxDist; q.nObs = m.nObs; q.bad = m.bad != 0; q.desc = cv::Mat(1, 32, CV_8UC1, (void*)d);
}
static void fill_line_keyframe(StructureSLAM::KeyFrame& K, const KeyLine* kl, const uint8_t* ldesc, int n, const float* bounds, const float* scale8, float logScaleFactor, const float* cam,
                               const float* Tcw, const float* Ow) {
    K.mvKeyLines.assign(kl, kl + n); K.mLineDescriptors.create(n, 32, CV_8UC1); if (n) std::memcpy(K.mLineDescriptors.data, ldesc, (size_t)n * 32);
    K.mvScaleFactors.assign(scale8, scale8 + 8); K.mfLogScaleFactor = logScaleFactor; K.mnScaleLevels = 8;
    K.fx = cam[0]; K.fy = cam[1]; K.cx = cam[2]; K.cy = cam[3]; K.mbf = cam[4];
    K.mnMinX = (int)bounds[0]; K.mnMaxX = (int)bounds[1]; K.mnMinY = (int)bounds[2]; K.mnMaxY = (int)bounds[3];
    K.Tcw = cv::Mat(4, 4, CV_32F); std::memcpy(K.Tcw.data, Tcw, 64); K.Ow = cv::Mat(3, 1, CV_32F); std::memcpy(K.Ow.data, Ow, 12);
}
extern "C" {
int ref_line_fuse(const KeyLine* kl, const uint8_t* ldesc, int n, const float* bounds, const float* scale8, float logScaleFactor, const uint8_t* state, const int32_t* stateObs,
                  const float* cam, const float* Tcw, const float* Ow, const FuseMl* ml, const uint8_t* mlDesc, int nml, float th, int32_t* fusedIdx, int32_t* action) {
    StructureSLAM::KeyFrame K; fill_line_keyframe(K, kl, ldesc, n, bounds, scale8, logScaleFactor, cam, Tcw, Ow);
    std::vector<StructureSLAM::MapLine> pool(nml), occ(n);
    std::vector<StructureSLAM::MapLine*> vp(nml);
    for (int i = 0; i < nml; ++i) { fill_fuse_line(pool[i], ml[i], mlDesc + (size_t)i * 32); vp[i] = &pool[i]; }
    K.mvpMapLines.assign(n, nullptr);
    for (int i = 0; i < n; ++i) if (state[i]) { occ[i].nObs = stateObs[i]; occ[i].bad = state[i] == 2; K.mvpMapLines[i] = &occ[i]; }
    gFuseL.kf = &K; gFuseL.pool = pool.data(); gFuseL.npool = nml; gFuseL.idx.assign(nml, -1); gFuseL.act.assign(nml, 0);
    StructureSLAM::LSDmatcher m(0.6f, true);
    const int r = m.Fuse(&K, vp, th);
    for (int i = 0; i < nml; ++i) { fusedIdx[i] = gFuseL.idx[i]; action[i] = gFuseL.act[i]; }
    gFuseL = FuseLogL(); return r;
}
// the windows of its projection block (:437-497): q[k] = {u1, v1, u2, v2, radius, predicted level, valid}
struct FuseQL { float u1, v1, u2, v2, radius; int level, valid; };
int ref_line_fuse_queries(const float* bounds, const float* scale8, float logScaleFactor, const float* cam, const float* Tcw, const float* Owv, const FuseMl* ml, int nml, float th, FuseQL* q) {
    StructureSLAM::KeyFrame K; fill_line_keyframe(K, nullptr, nullptr, 0, bounds, scale8, logScaleFactor, cam, Tcw, Owv);
    cv::Mat Rcw = K.GetRotation(), tcw = K.GetTranslation(), Ow = K.GetCameraCenter();
    const float fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
    uint8_t zero[32] = {0};
    for (int i = 0; i < nml; ++i) {
        FuseQL& Q = q[i]; Q = FuseQL();
        StructureSLAM::MapLine L; fill_fuse_line(L, ml[i], zero);
        if (L.isBad()) continue;
        Vector6d P = L.GetWorldPos();
        cv::Mat SP = (Mat_<float>(3, 1) << P(0), P(1), P(2)); cv::Mat EP = (Mat_<float>(3, 1) << P(3), P(4), P(5));
        const cv::Mat SPc = Rcw * SP + tcw; const cv::Mat EPc = Rcw * EP + tcw;
        const float SPcX = SPc.at<float>(0), SPcY = SPc.at<float>(1), SPcZ = SPc.at<float>(2), EPcX = EPc.at<float>(0), EPcY = EPc.at<float>(1), EPcZ = EPc.at<float>(2);
        if (SPcZ < 0.0f || EPcZ < 0.0f) continue;
        const float invz1 = 1.0f / SPcZ; const float u1 = fx * SPcX * invz1 + cx; const float v1 = fy * SPcY * invz1 + cy;
        if (u1 < K.mnMinX || u1 > K.mnMaxX) continue;
        if (v1 < K.mnMinY || v1 > K.mnMaxY) continue;
        const float invz2 = 1.0f / EPcZ; const float u2 = fx * EPcX * invz2 + cx; const float v2 = fy * EPcY * invz2 + cy;
        if (u2 < K.mnMinX || u2 > K.mnMaxX) continue;
        if (v2 < K.mnMinY || v2 > K.mnMaxY) continue;
        const float maxDistance = L.GetMaxDistanceInvariance(); const float minDistance = L.GetMinDistanceInvariance();
        const cv::Mat OM = 0.5 * (SP + EP) - Ow; const float dist = cv::norm(OM);
        if (dist < minDistance || dist > maxDistance) continue;
        Vector3d Pn = L.GetNormal(); cv::Mat pn = (Mat_<float>(3, 1) << Pn(0), Pn(1), Pn(2));
        if (OM.dot(pn) < 0.5 * dist) continue;
        const int lvl = L.PredictScale(dist, K.mfLogScaleFactor);
        Q.u1 = u1; Q.v1 = v1; Q.u2 = u2; Q.v2 = v2; Q.level = lvl; Q.valid = lvl >= 0 && lvl < 8 ? 1 : 2;      // 2: the reference would index mvScaleFactors out of range (no clamp in MapLine::PredictScale)
        if (Q.valid == 1) Q.radius = th * K.mvScaleFactors[lvl];
    }
    return 0;
}
// LSDmatcher::SearchByProjection(KeyFrame*, Scw, vpLines, vpMatched, th) (src/LSDmatcher.cpp:558-683); matched[idx] as in ref_search_by_projection_sim3
int ref_line_search_by_projection_sim3(const KeyLine* kl, const uint8_t* ldesc, int n, const float* bounds, const float* scale8, float logScaleFactor, const float* cam, const float* Scw,
                                       const FuseMl* ml, const uint8_t* mlDesc, int nml, int th, int32_t* matched) {
    const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, O3[3] = {0, 0, 0};
    StructureSLAM::KeyFrame K; fill_line_keyframe(K, kl, ldesc, n, bounds, scale8, logScaleFactor, cam, I4, O3);
    std::vector<StructureSLAM::MapLine> pool(nml), other(1);
    std::vector<StructureSLAM::MapLine*> vp(nml), vm(n, nullptr);
    for (int i = 0; i < nml; ++i) { fill_fuse_line(pool[i], ml[i], mlDesc + (size_t)i * 32); vp[i] = &pool[i]; }
    for (int i = 0; i < n; ++i) vm[i] = matched[i] == -1 ? nullptr : matched[i] == -2 ? &other[0] : &pool[matched[i]];
    cv::Mat S(4, 4, CV_32F); std::memcpy(S.data, Scw, 64);
    StructureSLAM::LSDmatcher m(0.75f, true);
    const int r = m.SearchByProjection(&K, S, vp, vm, th);
    for (int i = 0; i < n; ++i) matched[i] = !vm[i] ? -1 : vm[i] == &other[0] ? -2 : (int32_t)(vm[i] - pool.data());
    return r;
}
// LSDmatcher::Fuse(KeyFrame*, Scw, vpLines, th, vpReplaceLine) (src/LSDmatcher.cpp:931-1063); state / kfSlot / outputs as in ref_fuse_sim3
int ref_line_fuse_sim3(const KeyLine* kl, const uint8_t* ldesc, int n, const float* bounds, const float* scale8, float logScaleFactor, const uint8_t* state, const int32_t* kfSlot,
                       const float* cam, const float* Scw, const FuseMl* ml, const uint8_t* mlDesc, int nml, float th, int32_t* fusedIdx, int32_t* action) {
    const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, O3[3] = {0, 0, 0};
    StructureSLAM::KeyFrame K; fill_line_keyframe(K, kl, ldesc, n, bounds, scale8, logScaleFactor, cam, I4, O3);
    std::vector<StructureSLAM::MapLine> pool(nml), occ(n);
    std::vector<StructureSLAM::MapLine*> vp(nml), rep(nml, nullptr);
    for (int i = 0; i < nml; ++i) { fill_fuse_line(pool[i], ml[i], mlDesc + (size_t)i * 32); vp[i] = &pool[i]; }
    K.mvpMapLines.assign(n, nullptr);
    for (int i = 0; i < n; ++i) if (state[i]) { occ[i].bad = state[i] == 2; K.mvpMapLines[i] = &occ[i]; }
    for (int k = 0; k < nml; ++k) if (kfSlot[k] >= 0) K.mvpMapLines[kfSlot[k]] = &pool[k];
    gFuseL.kf = &K; gFuseL.pool = pool.data(); gFuseL.npool = nml; gFuseL.idx.assign(nml, -1); gFuseL.act.assign(nml, 0);
    cv::Mat S(4, 4, CV_32F); std::memcpy(S.data, Scw, 64);
    StructureSLAM::LSDmatcher m(0.8f, true);
    const int r = m.Fuse(&K, S, vp, th, rep);
    for (int i = 0; i < nml; ++i) { fusedIdx[i] = gFuseL.idx[i]; action[i] = rep[i] ? 2 : gFuseL.act[i]; }
    gFuseL = FuseLogL(); return r;
}
// the windows of their (common) projection block: the Sim3 decomposition, then as ref_line_fuse_queries; skip[k]: the candidate is already in the keyframe / in vpMatched
int ref_line_sim3_queries(const float* bounds, const float* scale8, float logScaleFactor, const float* cam, const float* Scw, const uint8_t* skip, const FuseMl* ml, int nml, float th, FuseQL* q) {
    const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, O3[3] = {0, 0, 0};
    StructureSLAM::KeyFrame K; fill_line_keyframe(K, nullptr, nullptr, 0, bounds, scale8, logScaleFactor, cam, I4, O3);
    const float fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
    cv::Mat S(4, 4, CV_32F); std::memcpy(S.data, Scw, 64);
    cv::Mat sRcw = S.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw; cv::Mat tcw = S.rowRange(0, 3).col(3) / scw; cv::Mat Ow = -Rcw.t() * tcw;
    uint8_t zero[32] = {0};
    for (int i = 0; i < nml; ++i) {
        FuseQL& Q = q[i]; Q = FuseQL();
        StructureSLAM::MapLine L; fill_fuse_line(L, ml[i], zero);
        if (L.isBad() || skip[i]) continue;
        Vector6d P = L.GetWorldPos();
        cv::Mat SP = (Mat_<float>(3, 1) << P(0), P(1), P(2)); cv::Mat EP = (Mat_<float>(3, 1) << P(3), P(4), P(5));
        const cv::Mat SPc = Rcw * SP + tcw; const cv::Mat EPc = Rcw * EP + tcw;
        const float SPcX = SPc.at<float>(0), SPcY = SPc.at<float>(1), SPcZ = SPc.at<float>(2), EPcX = EPc.at<float>(0), EPcY = EPc.at<float>(1), EPcZ = EPc.at<float>(2);
        if (SPcZ < 0.0f || EPcZ < 0.0f) continue;
        const float invz1 = 1.0f / SPcZ; const float u1 = fx * SPcX * invz1 + cx; const float v1 = fy * SPcY * invz1 + cy;
        if (u1 < K.mnMinX || u1 > K.mnMaxX) continue;
        if (v1 < K.mnMinY || v1 > K.mnMaxY) continue;
        const float invz2 = 1.0f / EPcZ; const float u2 = fx * EPcX * invz2 + cx; const float v2 = fy * EPcY * invz2 + cy;
        if (u2 < K.mnMinX || u2 > K.mnMaxX) continue;
        if (v2 < K.mnMinY || v2 > K.mnMaxY) continue;
        const float maxDistance = L.GetMaxDistanceInvariance(); const float minDistance = L.GetMinDistanceInvariance();
        const cv::Mat OM = 0.5 * (SP + EP) - Ow; const float dist = cv::norm(OM);
        if (dist < minDistance || dist > maxDistance) continue;
        Vector3d Pn = L.GetNormal(); cv::Mat pn = (Mat_<float>(3, 1) << Pn(0), Pn(1), Pn(2));
        if (OM.dot(pn) < 0.5 * dist) continue;
        const int lvl = L.PredictScale(dist, K.mfLogScaleFactor);
        Q.u1 = u1; Q.v1 = v1; Q.u2 = u2; Q.v2 = v2; Q.level = lvl; Q.valid = lvl >= 0 && lvl < 8 ? 1 : 2;
        if (Q.valid == 1) Q.radius = th * K.mvScaleFactors[lvl];
    }
    return 0;
}
// LSDmatcher::SearchBySim3 (src/LSDmatcher.cpp:685-929); arguments as ref_search_by_sim3, with key lines and map lines
int ref_line_search_by_sim3(const KeyLine* kl1, const uint8_t* ld1, int n1, const KeyLine* kl2, const uint8_t* ld2, int n2, const float* bounds, const float* scale8, float logScaleFactor,
                            const float* cam, const float* T1w, const float* T2w, float s12, const float* R12, const float* t12, const uint8_t* present1, const FuseMl* ml1, const uint8_t* mlDesc1,
                            const uint8_t* present2, const FuseMl* ml2, const uint8_t* mlDesc2, float th, int32_t* m12) {
    const float O3[3] = {0, 0, 0};
    StructureSLAM::KeyFrame K1, K2;
    fill_line_keyframe(K1, kl1, ld1, n1, bounds, scale8, logScaleFactor, cam, T1w, O3); fill_line_keyframe(K2, kl2, ld2, n2, bounds, scale8, logScaleFactor, cam, T2w, O3);
    std::vector<StructureSLAM::MapLine> pool1(n1), pool2(n2), other(1);
    K1.mvpMapLines.assign(n1, nullptr); K2.mvpMapLines.assign(n2, nullptr);
    for (int i = 0; i < n1; ++i) if (present1[i]) { fill_fuse_line(pool1[i], ml1[i], mlDesc1 + (size_t)i * 32); K1.mvpMapLines[i] = &pool1[i]; }
    for (int i = 0; i < n2; ++i) if (present2[i]) { fill_fuse_line(pool2[i], ml2[i], mlDesc2 + (size_t)i * 32); pool2[i].idxInKF2 = i; K2.mvpMapLines[i] = &pool2[i]; }
    std::vector<StructureSLAM::MapLine*> vm(n1, nullptr);
    for (int i = 0; i < n1; ++i) vm[i] = m12[i] == -1 ? nullptr : m12[i] == -2 ? &other[0] : &pool2[m12[i]];
    cv::Mat R(3, 3, CV_32F), t(3, 1, CV_32F); std::memcpy(R.data, R12, 36); std::memcpy(t.data, t12, 12);
    StructureSLAM::LSDmatcher m(0.75f, true);
    const int r = m.SearchBySim3(&K1, &K2, vm, s12, R, t, th);
    for (int i = 0; i < n1; ++i) m12[i] = !vm[i] ? -1 : vm[i] == &other[0] ? -2 : (int32_t)(vm[i] - pool2.data());
    return r;
}
int ref_line_sim3_pair_queries(int dir, const float* bounds, const float* scale8, float logScaleFactor, const float* cam, const float* T1w, const float* T2w, float s12, const float* R12v,
                               const float* t12v, const uint8_t* skip, const FuseMl* ml, int nml, float th, FuseQL* q) {
    const float O3[3] = {0, 0, 0};
    StructureSLAM::KeyFrame K1, K2; fill_line_keyframe(K1, nullptr, nullptr, 0, bounds, scale8, logScaleFactor, cam, T1w, O3); fill_line_keyframe(K2, nullptr, nullptr, 0, bounds, scale8, logScaleFactor, cam, T2w, O3);
    const float fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
    cv::Mat R1w = K1.GetRotation(), t1w = K1.GetTranslation(), R2w = K2.GetRotation(), t2w = K2.GetTranslation();
    cv::Mat R12(3, 3, CV_32F), t12(3, 1, CV_32F); std::memcpy(R12.data, R12v, 36); std::memcpy(t12.data, t12v, 12);
    cv::Mat sR12 = s12 * R12; cv::Mat sR21 = (1.0 / s12) * R12.t(); cv::Mat t21 = -sR21 * t12;
    uint8_t zero[32] = {0};
    for (int i = 0; i < nml; ++i) {
        FuseQL& Q = q[i]; Q = FuseQL();
        if (skip[i]) continue;
        StructureSLAM::MapLine L; fill_fuse_line(L, ml[i], zero);
        if (L.isBad()) continue;
        Vector6d P = L.GetWorldPos();
        cv::Mat SP = (Mat_<float>(3, 1) << P(0), P(1), P(2)); cv::Mat EP = (Mat_<float>(3, 1) << P(3), P(4), P(5));
        const cv::Mat SPa = dir == 0 ? cv::Mat(R1w * SP + t1w) : cv::Mat(R2w * SP + t2w); const cv::Mat SPb = dir == 0 ? cv::Mat(sR21 * SPa + t21) : cv::Mat(sR12 * SPa + t12);
        const cv::Mat EPa = dir == 0 ? cv::Mat(R1w * EP + t1w) : cv::Mat(R2w * EP + t2w); const cv::Mat EPb = dir == 0 ? cv::Mat(sR21 * EPa + t21) : cv::Mat(sR12 * EPa + t12);
        const float SPcX = SPb.at<float>(0), SPcY = SPb.at<float>(1), SPcZ = SPb.at<float>(2), EPcX = EPb.at<float>(0), EPcY = EPb.at<float>(1), EPcZ = EPb.at<float>(2);
        if (SPcZ < 0.0f || EPcZ < 0.0f) continue;
        StructureSLAM::KeyFrame* Kt = dir == 0 ? &K2 : &K1;
        const float invz1 = 1.0f / SPcZ; const float u1 = fx * SPcX * invz1 + cx; const float v1 = fy * SPcY * invz1 + cy;
        if (!Kt->IsInImage(u1, v1)) continue;
        const float invz2 = 1.0f / EPcZ; const float u2 = fx * EPcX * invz2 + cx; const float v2 = fy * EPcY * invz2 + cy;
        if (!Kt->IsInImage(u2, v2)) continue;
        const float maxDistance = L.GetMaxDistanceInvariance(); const float minDistance = L.GetMinDistanceInvariance();
        const float dist3D = cv::norm(0.5 * (SPb + EPb));
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int lvl = L.PredictScale(dist3D, Kt->mfLogScaleFactor);
        Q.u1 = u1; Q.v1 = v1; Q.u2 = u2; Q.v2 = v2; Q.level = lvl; Q.valid = lvl >= 0 && lvl < 8 ? 1 : 2;
        if (Q.valid == 1) Q.radius = th * Kt->mvScaleFactors[lvl];
    }
    return 0;
}
// state[i]: 0 the keyframe's slot i holds no map point, 1 a good one with stateObs[i] observations, 2 a bad one.  cam = {fx, fy, cx, cy, mbf}.
// fusedIdx[k] / action[k] for map point k: the keyframe feature it was fused to (-1) and how (0 not, 1 new observation, 2 replaced BY the keyframe's point,
// 3 it replaced the keyframe's point, 4 the slot holds a bad point: counted, nothing done); returns nFused
int ref_fuse(const cv::KeyPoint* kp, const uint8_t* desc, int n, const float* bounds, const float* scale8, const float* invSigma2_8, float logScaleFactor, const float* uright,
             const uint8_t* state, const int32_t* stateObs, const float* cam, const float* Tcw, const float* Ow, const FuseMp* mp, const uint8_t* mpDesc, int nmp, float th,
             int32_t* fusedIdx, int32_t* action) {
    StructureSLAM::Frame* F = new StructureSLAM::Frame(); StructureSLAM::KeyFrame K;
    fill_keyframe(K, *F, kp, desc, n, bounds, scale8, invSigma2_8, logScaleFactor, uright, cam, Tcw, Ow);
    std::vector<StructureSLAM::MapPoint> pool(nmp), occ(n);
    std::vector<StructureSLAM::MapPoint*> vp(nmp);
    for (int i = 0; i < nmp; ++i) { fill_fuse_point(pool[i], mp[i], mpDesc + (size_t)i * 32); vp[i] = &pool[i]; }
    K.mvpMapPoints.assign(n, nullptr);
    for (int i = 0; i < n; ++i) if (state[i]) { occ[i].nObs = stateObs[i]; occ[i].bad = state[i] == 2; K.mvpMapPoints[i] = &occ[i]; }
    gFuse.kf = &K; gFuse.pool = pool.data(); gFuse.npool = nmp; gFuse.occ = occ.data(); gFuse.nocc = n; gFuse.idx.assign(nmp, -1); gFuse.act.assign(nmp, 0);
    StructureSLAM::ORBmatcher m(0.6f, true);
    const int r = m.Fuse(&K, vp, th);
    for (int i = 0; i < nmp; ++i) { fusedIdx[i] = gFuse.idx[i]; action[i] = gFuse.act[i]; }
    gFuse = FuseLog(); delete F; return r;
}
// The window the reference's projection block (:853-894) forms for every map point, on the same stand-in objects and leaves: what a caller of
// sslam_fuse_search / the oracle's fuse_search passes as its queries.  q[k] = {u, v, ur, radius, predicted level, valid}
struct FuseQ { float u, v, ur, radius; int level, valid; };
int ref_fuse_queries(const float* bounds, const float* scale8, float logScaleFactor, const float* cam, const float* Tcw, const float* Ow, const FuseMp* mp, int nmp, float th, FuseQ* q) {
    StructureSLAM::KeyFrame K; K.mvScaleFactors.assign(scale8, scale8 + 8); K.mfLogScaleFactor = logScaleFactor; K.mnScaleLevels = 8;
    K.mnMinX = (int)bounds[0]; K.mnMaxX = (int)bounds[1]; K.mnMinY = (int)bounds[2]; K.mnMaxY = (int)bounds[3];
    K.Tcw = cv::Mat(4, 4, CV_32F); std::memcpy(K.Tcw.data, Tcw, 64); K.Ow = cv::Mat(3, 1, CV_32F); std::memcpy(K.Ow.data, Ow, 12);
    const cv::Mat Rcw = K.GetRotation(), tcw = K.GetTranslation(), Owm = K.GetCameraCenter();
    const float fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3], bf = cam[4];
    uint8_t zero[32] = {0};
    for (int i = 0; i < nmp; ++i) {
        StructureSLAM::MapPoint P; fill_fuse_point(P, mp[i], zero);
        FuseQ& Q = q[i]; Q = FuseQ();
        if (P.isBad() || P.IsInKeyFrame(&K)) continue;
        cv::Mat p3Dw = P.GetWorldPos(); cv::Mat p3Dc = Rcw * p3Dw + tcw;
        if (p3Dc.at<float>(2) < 0.0f) continue;
        const float invz = 1 / p3Dc.at<float>(2); const float x = p3Dc.at<float>(0) * invz; const float y = p3Dc.at<float>(1) * invz;
        const float u = fx * x + cx; const float v = fy * y + cy;
        if (!K.IsInImage(u, v)) continue;
        const float ur = u - bf * invz;
        const float maxDistance = P.GetMaxDistanceInvariance(); const float minDistance = P.GetMinDistanceInvariance();
        cv::Mat PO = p3Dw - Owm; const float dist3D = cv::norm(PO);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        cv::Mat Pn = P.GetNormal();
        if (PO.dot(Pn) < 0.5 * dist3D) continue;
        const int lvl = P.PredictScale(dist3D, &K);
        Q.u = u; Q.v = v; Q.ur = ur; Q.radius = th * K.mvScaleFactors[lvl]; Q.level = lvl; Q.valid = 1;
    }
    return 0;
}
// ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:1475-1602, Tracking::Relocalization's calls :1634, :1647).
// kpKF[i]: pKF->mvKeysUn (the angle of the rotation histogram); mp[i]: the keyframe's map point i (present[i] 0: NULL, 1: a point, 2: a point in sAlreadyFound; mp[i].inKF unused);
// stateCur[i2]: CurrentFrame.mvpMapPoints[i2] != NULL on entry.  assigned[i2]: the keyframe's map-point index matched to feature i2, -1 untouched, -2 occupied on entry,
// -4 matched and removed by the rotation histogram (the reference sets NULL: told apart from "untouched" through the occupied-before / after pair)
int ref_search_by_projection_reloc(const cv::KeyPoint* kp, const uint8_t* desc, int n, const float* bounds, const float* scale8, float logScaleFactor, const uint8_t* stateCur,
                                   const float* cam, const float* Tcw, const cv::KeyPoint* kpKF, const uint8_t* present, const FuseMp* mp, const uint8_t* mpDesc, int nkf,
                                   float th, int ORBdist, int checkOri, int32_t* assigned) {
    StructureSLAM::Frame* F = new StructureSLAM::Frame(); set_frame_common(*F, kp, desc, n, bounds, scale8, nullptr);
    F->fx = cam[0]; F->fy = cam[1]; F->cx = cam[2]; F->cy = cam[3]; F->mfLogScaleFactor = logScaleFactor; F->mnScaleLevels = 8;
    F->mTcw = cv::Mat(4, 4, CV_32F); std::memcpy(F->mTcw.data, Tcw, 64);
    StructureSLAM::KeyFrame K; K.N = nkf; K.mvKeysUn.assign(kpKF, kpKF + nkf);
    std::vector<StructureSLAM::MapPoint> pool(nkf), occ(1);
    std::set<StructureSLAM::MapPoint*> found;
    K.mvpMapPoints.assign(nkf, nullptr);
    for (int i = 0; i < nkf; ++i) if (present[i]) { fill_fuse_point(pool[i], mp[i], mpDesc + (size_t)i * 32); K.mvpMapPoints[i] = &pool[i]; if (present[i] == 2) found.insert(&pool[i]); }
    F->mvpMapPoints.assign(n, nullptr);
    for (int i = 0; i < n; ++i) if (stateCur[i]) F->mvpMapPoints[i] = &occ[0];
    StructureSLAM::ORBmatcher m(0.9f, checkOri != 0);
    const int r = m.SearchByProjection(*F, &K, found, th, ORBdist);
    for (int i = 0; i < n; ++i) {
        const StructureSLAM::MapPoint* p = F->mvpMapPoints[i];
        assigned[i] = !p ? -1 : p == &occ[0] ? -2 : (int32_t)(p - pool.data());
    }
    delete F; return r;
}
// the windows the reference's projection block (:1499-1530) forms: q[i] = {u, v, -, radius, predicted level, valid}
int ref_reloc_queries(const float* bounds, const float* scale8, float logScaleFactor, const float* cam, const float* Tcw, const uint8_t* present, const FuseMp* mp, int nkf, float th, FuseQ* q) {
    StructureSLAM::Frame F; F.mvScaleFactors.assign(scale8, scale8 + 8); F.mfLogScaleFactor = logScaleFactor; F.mnScaleLevels = 8;
    F.fx = cam[0]; F.fy = cam[1]; F.cx = cam[2]; F.cy = cam[3];
    StructureSLAM::Frame::mnMinX = bounds[0]; StructureSLAM::Frame::mnMaxX = bounds[1]; StructureSLAM::Frame::mnMinY = bounds[2]; StructureSLAM::Frame::mnMaxY = bounds[3];
    F.mTcw = cv::Mat(4, 4, CV_32F); std::memcpy(F.mTcw.data, Tcw, 64);
    const cv::Mat Rcw = F.mTcw.rowRange(0, 3).colRange(0, 3); const cv::Mat tcw = F.mTcw.rowRange(0, 3).col(3); const cv::Mat Ow = -Rcw.t() * tcw;
    uint8_t zero[32] = {0};
    for (int i = 0; i < nkf; ++i) {
        FuseQ& Q = q[i]; Q = FuseQ();
        if (present[i] != 1) continue;
        StructureSLAM::MapPoint P; fill_fuse_point(P, mp[i], zero);
        if (P.isBad()) continue;
        cv::Mat x3Dw = P.GetWorldPos(); cv::Mat x3Dc = Rcw * x3Dw + tcw;
        const float xc = x3Dc.at<float>(0); const float yc = x3Dc.at<float>(1); const float invzc = 1.0 / x3Dc.at<float>(2);
        const float u = F.fx * xc * invzc + F.cx; const float v = F.fy * yc * invzc + F.cy;
        if (u < StructureSLAM::Frame::mnMinX || u > StructureSLAM::Frame::mnMaxX) continue;
        if (v < StructureSLAM::Frame::mnMinY || v > StructureSLAM::Frame::mnMaxY) continue;
        cv::Mat PO = x3Dw - Ow; float dist3D = cv::norm(PO);
        const float maxDistance = P.GetMaxDistanceInvariance(); const float minDistance = P.GetMinDistanceInvariance();
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int lvl = P.PredictScale(dist3D, &F);
        Q.u = u; Q.v = v; Q.radius = th * F.mvScaleFactors[lvl]; Q.level = lvl; Q.valid = 1;
    }
    return 0;
}
// ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th) (src/ORBmatcher.cc:293-406, LoopClosing::ComputeSim3 :449 with th = 10).
// matched[idx] on entry AND exit: -1 NULL, -2 a map point that is not in vpPoints, k >= 0 vpPoints[k]
int ref_search_by_projection_sim3(const cv::KeyPoint* kp, const uint8_t* desc, int n, const float* bounds, const float* scale8, float logScaleFactor, const float* cam,
                                  const float* Scw, const FuseMp* mp, const uint8_t* mpDesc, int nmp, int th, int32_t* matched) {
    StructureSLAM::Frame* F = new StructureSLAM::Frame(); StructureSLAM::KeyFrame K;
    const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, O3[3] = {0, 0, 0}, one8[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    fill_keyframe(K, *F, kp, desc, n, bounds, scale8, one8, logScaleFactor, nullptr, cam, I4, O3);
    std::vector<StructureSLAM::MapPoint> pool(nmp), other(1);
    std::vector<StructureSLAM::MapPoint*> vp(nmp), vm(n, nullptr);
    for (int i = 0; i < nmp; ++i) { fill_fuse_point(pool[i], mp[i], mpDesc + (size_t)i * 32); vp[i] = &pool[i]; }
    for (int i = 0; i < n; ++i) vm[i] = matched[i] == -1 ? nullptr : matched[i] == -2 ? &other[0] : &pool[matched[i]];
    cv::Mat S(4, 4, CV_32F); std::memcpy(S.data, Scw, 64);
    StructureSLAM::ORBmatcher m(0.75f, true);
    const int r = m.SearchByProjection(&K, S, vp, vm, th);
    for (int i = 0; i < n; ++i) matched[i] = !vm[i] ? -1 : vm[i] == &other[0] ? -2 : (int32_t)(vm[i] - pool.data());
    delete F; return r;
}
// the windows of its projection block (:296-360): q[k] = {u, v, -, radius, predicted level, valid}; found[k]: vpPoints[k] is in vpMatched on entry
int ref_sim3_queries(const float* bounds, const float* scale8, float logScaleFactor, const float* cam, const float* Scw, const uint8_t* found, const FuseMp* mp, int nmp, float th, int fuseForm, FuseQ* q) {
    StructureSLAM::KeyFrame K; K.mvScaleFactors.assign(scale8, scale8 + 8); K.mfLogScaleFactor = logScaleFactor; K.mnScaleLevels = 8;
    K.mnMinX = (int)bounds[0]; K.mnMaxX = (int)bounds[1]; K.mnMinY = (int)bounds[2]; K.mnMaxY = (int)bounds[3];
    const float fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
    cv::Mat S(4, 4, CV_32F); std::memcpy(S.data, Scw, 64);
    cv::Mat sRcw = S.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw; cv::Mat tcw = S.rowRange(0, 3).col(3) / scw; cv::Mat Ow = -Rcw.t() * tcw;
    uint8_t zero[32] = {0};
    for (int i = 0; i < nmp; ++i) {
        FuseQ& Q = q[i]; Q = FuseQ();
        StructureSLAM::MapPoint P; fill_fuse_point(P, mp[i], zero);
        if (P.isBad() || found[i]) continue;
        cv::Mat p3Dw = P.GetWorldPos(); cv::Mat p3Dc = Rcw * p3Dw + tcw;
        if (p3Dc.at<float>(2) < 0.0) continue;
        const float invz = fuseForm ? (float)(1.0 / p3Dc.at<float>(2)) : 1 / p3Dc.at<float>(2);      // (:1024 divides in double, :336 in float)
        const float x = p3Dc.at<float>(0) * invz; const float y = p3Dc.at<float>(1) * invz;
        const float u = fx * x + cx; const float v = fy * y + cy;
        if (!K.IsInImage(u, v)) continue;
        const float maxDistance = P.GetMaxDistanceInvariance(); const float minDistance = P.GetMinDistanceInvariance();
        cv::Mat PO = p3Dw - Ow; const float dist = cv::norm(PO);
        if (dist < minDistance || dist > maxDistance) continue;
        cv::Mat Pn = P.GetNormal();
        if (PO.dot(Pn) < 0.5 * dist) continue;
        const int lvl = P.PredictScale(dist, &K);
        Q.u = u; Q.v = v; Q.radius = th * K.mvScaleFactors[lvl]; Q.level = lvl; Q.valid = 1;
    }
    return 0;
}
// ORBmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, vpPoints, th, vpReplacePoint) (src/ORBmatcher.cc:980-1103; LoopClosing::SearchAndFuse :665 with th = 4).
// state / stateObs as in ref_fuse; inKF[k]: vpPoints[k] is one of the keyframe's own map points (it sits in slot kfSlot[k]).
// fusedIdx / action as in ref_fuse (2: vpReplacePoint[k] = the keyframe's point)
int ref_fuse_sim3(const cv::KeyPoint* kp, const uint8_t* desc, int n, const float* bounds, const float* scale8, float logScaleFactor, const uint8_t* state, const int32_t* kfSlot,
                  const float* cam, const float* Scw, const FuseMp* mp, const uint8_t* mpDesc, int nmp, float th, int32_t* fusedIdx, int32_t* action) {
    StructureSLAM::Frame* F = new StructureSLAM::Frame(); StructureSLAM::KeyFrame K;
    const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, O3[3] = {0, 0, 0}, one8[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    fill_keyframe(K, *F, kp, desc, n, bounds, scale8, one8, logScaleFactor, nullptr, cam, I4, O3);
    std::vector<StructureSLAM::MapPoint> pool(nmp), occ(n);
    std::vector<StructureSLAM::MapPoint*> vp(nmp), rep(nmp, nullptr);
    for (int i = 0; i < nmp; ++i) { fill_fuse_point(pool[i], mp[i], mpDesc + (size_t)i * 32); vp[i] = &pool[i]; }
    K.mvpMapPoints.assign(n, nullptr);
    for (int i = 0; i < n; ++i) if (state[i]) { occ[i].bad = state[i] == 2; K.mvpMapPoints[i] = &occ[i]; }
    for (int k = 0; k < nmp; ++k) if (kfSlot[k] >= 0) K.mvpMapPoints[kfSlot[k]] = &pool[k];      // candidates the keyframe already observes
    gFuse.kf = &K; gFuse.pool = pool.data(); gFuse.npool = nmp; gFuse.idx.assign(nmp, -1); gFuse.act.assign(nmp, 0);
    cv::Mat S(4, 4, CV_32F); std::memcpy(S.data, Scw, 64);
    StructureSLAM::ORBmatcher m(0.8f, true);
    const int r = m.Fuse(&K, S, vp, th, rep);
    for (int i = 0; i < nmp; ++i) { fusedIdx[i] = gFuse.idx[i]; action[i] = rep[i] ? 2 : gFuse.act[i]; }
    gFuse = FuseLog(); delete F; return r;
}
// ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (src/ORBmatcher.cc:1105-1329; LoopClosing::ComputeSim3 :416 with th = 7.5).
// present1 / present2: the keyframes' map-point slots (0 NULL, 1 a point); m12[i1] on entry and exit: -1 NULL, -2 a map point pKF2 does not observe, k >= 0 pKF2's point of slot k
int ref_search_by_sim3(const cv::KeyPoint* kp1, const uint8_t* d1, int n1, const cv::KeyPoint* kp2, const uint8_t* d2, int n2, const float* bounds, const float* scale8, float logScaleFactor,
                       const float* cam, const float* T1w, const float* T2w, float s12, const float* R12, const float* t12, const uint8_t* present1, const FuseMp* mp1, const uint8_t* mpDesc1,
                       const uint8_t* present2, const FuseMp* mp2, const uint8_t* mpDesc2, float th, int32_t* m12) {
    const float O3[3] = {0, 0, 0}, one8[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    StructureSLAM::Frame* F1 = new StructureSLAM::Frame(); StructureSLAM::Frame* F2 = new StructureSLAM::Frame(); StructureSLAM::KeyFrame K1, K2;
    fill_keyframe(K1, *F1, kp1, d1, n1, bounds, scale8, one8, logScaleFactor, nullptr, cam, T1w, O3);
    fill_keyframe(K2, *F2, kp2, d2, n2, bounds, scale8, one8, logScaleFactor, nullptr, cam, T2w, O3);
    std::vector<StructureSLAM::MapPoint> pool1(n1), pool2(n2), other(1);
    K1.mvpMapPoints.assign(n1, nullptr); K2.mvpMapPoints.assign(n2, nullptr);
    for (int i = 0; i < n1; ++i) if (present1[i]) { fill_fuse_point(pool1[i], mp1[i], mpDesc1 + (size_t)i * 32); K1.mvpMapPoints[i] = &pool1[i]; }
    for (int i = 0; i < n2; ++i) if (present2[i]) { fill_fuse_point(pool2[i], mp2[i], mpDesc2 + (size_t)i * 32); pool2[i].idxInKF2 = i; K2.mvpMapPoints[i] = &pool2[i]; }
    std::vector<StructureSLAM::MapPoint*> vm(n1, nullptr);
    for (int i = 0; i < n1; ++i) vm[i] = m12[i] == -1 ? nullptr : m12[i] == -2 ? &other[0] : &pool2[m12[i]];
    cv::Mat R(3, 3, CV_32F), t(3, 1, CV_32F); std::memcpy(R.data, R12, 36); std::memcpy(t.data, t12, 12);
    StructureSLAM::ORBmatcher m(0.75f, true);
    const int r = m.SearchBySim3(&K1, &K2, vm, s12, R, t, th);
    for (int i = 0; i < n1; ++i) m12[i] = !vm[i] ? -1 : vm[i] == &other[0] ? -2 : (int32_t)(vm[i] - pool2.data());
    delete F1; delete F2; return r;
}
// the windows of its two projection blocks: dir 0 = pKF1's points into pKF2 (:1152-1196: R1w, t1w then sR21, t21), dir 1 = pKF2's into pKF1 (:1232-1276: R2w, t2w then sR12, t12);
// skip[i]: the slot is NULL or already matched
int ref_sim3_pair_queries(int dir, const float* bounds, const float* scale8, float logScaleFactor, const float* cam, const float* T1w, const float* T2w, float s12, const float* R12v, const float* t12v,
                          const uint8_t* skip, const FuseMp* mp, int nmp, float th, FuseQ* q) {
    StructureSLAM::KeyFrame K1, K2;
    for (StructureSLAM::KeyFrame* K : {&K1, &K2}) {
        K->mvScaleFactors.assign(scale8, scale8 + 8); K->mfLogScaleFactor = logScaleFactor; K->mnScaleLevels = 8;
        K->mnMinX = (int)bounds[0]; K->mnMaxX = (int)bounds[1]; K->mnMinY = (int)bounds[2]; K->mnMaxY = (int)bounds[3];
    }
    K1.Tcw = cv::Mat(4, 4, CV_32F); std::memcpy(K1.Tcw.data, T1w, 64); K2.Tcw = cv::Mat(4, 4, CV_32F); std::memcpy(K2.Tcw.data, T2w, 64);
    const float fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
    cv::Mat R1w = K1.GetRotation(), t1w = K1.GetTranslation(), R2w = K2.GetRotation(), t2w = K2.GetTranslation();
    cv::Mat R12(3, 3, CV_32F), t12(3, 1, CV_32F); std::memcpy(R12.data, R12v, 36); std::memcpy(t12.data, t12v, 12);
    cv::Mat sR12 = s12 * R12; cv::Mat sR21 = (1.0 / s12) * R12.t(); cv::Mat t21 = -sR21 * t12;
    uint8_t zero[32] = {0};
    for (int i = 0; i < nmp; ++i) {
        FuseQ& Q = q[i]; Q = FuseQ();
        if (skip[i]) continue;
        StructureSLAM::MapPoint P; fill_fuse_point(P, mp[i], zero);
        if (P.isBad()) continue;
        cv::Mat p3Dw = P.GetWorldPos();
        cv::Mat pa = dir == 0 ? cv::Mat(R1w * p3Dw + t1w) : cv::Mat(R2w * p3Dw + t2w);
        cv::Mat pb = dir == 0 ? cv::Mat(sR21 * pa + t21) : cv::Mat(sR12 * pa + t12);
        if (pb.at<float>(2) < 0.0) continue;
        const float invz = 1.0 / pb.at<float>(2); const float x = pb.at<float>(0) * invz; const float y = pb.at<float>(1) * invz;
        const float u = fx * x + cx; const float v = fy * y + cy;
        StructureSLAM::KeyFrame* Kt = dir == 0 ? &K2 : &K1;
        if (!Kt->IsInImage(u, v)) continue;
        const float maxDistance = P.GetMaxDistanceInvariance(); const float minDistance = P.GetMinDistanceInvariance();
        const float dist3D = cv::norm(pb);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int lvl = P.PredictScale(dist3D, Kt);
        Q.u = u; Q.v = v; Q.radius = th * Kt->mvScaleFactors[lvl]; Q.level = lvl; Q.valid = 1;
    }
    return 0;
}
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

int ref_search_by_projection_mappoints(const cv::KeyPoint* kp, const uint8_t* desc, int n, const float* bounds, const float* scale8, const float* uright, const uint8_t* state,
                                       const MpIn* mp, const uint8_t* mpDesc, int nmp, float th, float nnratio, int32_t* assigned) {
    Frame* F = new Frame(); set_frame_common(*F, kp, desc, n, bounds, scale8, uright);
    std::vector<StructureSLAM::MapPoint> pool(nmp), occ(2); occ[0].nObs = 1; occ[1].nObs = 0;
    std::vector<StructureSLAM::MapPoint*> vp(nmp);
    for (int i = 0; i < nmp; ++i) {
        StructureSLAM::MapPoint& q = pool[i];
        q.mbTrackInView = mp[i].inView != 0; q.bad = mp[i].bad != 0; q.mnTrackScaleLevel = mp[i].level; q.nObs = mp[i].nObs; q.mTrackViewCos = mp[i].viewCos;
        q.mTrackProjX = mp[i].projX; q.mTrackProjY = mp[i].projY; q.mTrackProjXR = mp[i].projXR; q.desc = cv::Mat(1, 32, CV_8UC1, (void*)(mpDesc + (size_t)i * 32));
        vp[i] = &q;
    }
    F->mvpMapPoints.assign(n, nullptr);
    for (int i = 0; i < n; ++i) if (state[i]) F->mvpMapPoints[i] = &occ[state[i] - 1];
    StructureSLAM::ORBmatcher m(nnratio, true);
    const int r = m.SearchByProjection(*F, vp, th);
    encode(F->mvpMapPoints, pool, occ, assigned);
    delete F; return r;
}
int ref_line_search_by_projection_maplines(const KeyLine* kl, const uint8_t* ldesc, int n, const float* scale8, const uint8_t* state,
                                           const MlIn* ml, const uint8_t* mlDesc, int nml, float th, float nnratio, int32_t* assigned) {
    Frame* F = new Frame(); F->NL = n; F->mvKeylinesUn.assign(kl, kl + n); F->mvScaleFactors.assign(scale8, scale8 + 8);
    F->mLdesc.create(n, 32, CV_8UC1); if (n) std::memcpy(F->mLdesc.data, ldesc, (size_t)n * 32);
    std::vector<StructureSLAM::MapLine> pool(nml), occ(2); occ[0].nObs = 1; occ[1].nObs = 0;
    std::vector<StructureSLAM::MapLine*> vp(nml);
    for (int i = 0; i < nml; ++i) {
        StructureSLAM::MapLine& q = pool[i];
        q.mbTrackInView = ml[i].inView != 0; q.bad = ml[i].bad != 0; q.mnTrackScaleLevel = ml[i].level; q.nObs = ml[i].nObs; q.mTrackViewCos = ml[i].viewCos;
        q.mTrackProjX1 = ml[i].x1; q.mTrackProjY1 = ml[i].y1; q.mTrackProjX2 = ml[i].x2; q.mTrackProjY2 = ml[i].y2; q.desc = cv::Mat(1, 32, CV_8UC1, (void*)(mlDesc + (size_t)i * 32));
        vp[i] = &q;
    }
    F->mvpMapLines.assign(n, nullptr);
    for (int i = 0; i < n; ++i) if (state[i]) F->mvpMapLines[i] = &occ[state[i] - 1];
    StructureSLAM::LSDmatcher m(nnratio, true);
    const int r = m.SearchByProjection(*F, vp, th);
    encode(F->mvpMapLines, pool, occ, assigned);
    delete F; return r;
}
// Tracking::TrackWithMotionModel's calls (src/Tracking.cc:1227, :1243): (CurrentFrame, LastFrame, th, bMono) for points and for lines.
// cam = {fx, fy, cx, cy, mbf, mb}; TL / TC = LastFrame.mTcw / CurrentFrame.mTcw (4 x 4 float, row major); has1[i]: 0 no map point, 1 one, outlier1[i]: mvbOutlier
int ref_track_points(const cv::KeyPoint* kp1, const uint8_t* d1, int n1, const uint8_t* has1, const uint8_t* outlier1, const int32_t* nObs1, const float* wp1,
                     const cv::KeyPoint* kp2, const uint8_t* d2, int n2, const uint8_t* state2, const float* uright2, const float* bounds, const float* scale8,
                     const float* cam, const float* TL, const float* TC, float th, int bMono, float nnratio, int32_t* assigned) {
    Frame* Last = new Frame(); Frame* Cur = new Frame();
    set_frame_common(*Cur, kp2, d2, n2, bounds, scale8, uright2); set_frame_common(*Last, kp1, d1, n1, bounds, scale8, nullptr);
    Cur->fx = cam[0]; Cur->fy = cam[1]; Cur->cx = cam[2]; Cur->cy = cam[3]; Cur->mbf = cam[4]; Cur->mb = cam[5];
    Cur->mTcw = cv::Mat(4, 4, CV_32F); Last->mTcw = cv::Mat(4, 4, CV_32F);
    std::memcpy(Cur->mTcw.data, TC, 64); std::memcpy(Last->mTcw.data, TL, 64);
    std::vector<StructureSLAM::MapPoint> pool(n1), occ(2); occ[0].nObs = 1; occ[1].nObs = 0;
    Last->mvpMapPoints.assign(n1, nullptr); Last->mvbOutlier.assign(n1, false);
    for (int i = 0; i < n1; ++i) {
        pool[i].nObs = nObs1[i]; pool[i].desc = cv::Mat(1, 32, CV_8UC1, (void*)(d1 + (size_t)i * 32)); pool[i].worldPos = cv::Mat(3, 1, CV_32F, (void*)(wp1 + 3 * (size_t)i));
        if (has1[i]) Last->mvpMapPoints[i] = &pool[i];
        Last->mvbOutlier[i] = outlier1[i] != 0;
    }
    Cur->mvpMapPoints.assign(n2, nullptr);
    for (int i = 0; i < n2; ++i) if (state2[i]) Cur->mvpMapPoints[i] = &occ[state2[i] - 1];
    StructureSLAM::ORBmatcher m(nnratio, true);
    const int r = m.SearchByProjection(*Cur, *Last, th, bMono != 0);
    encode(Cur->mvpMapPoints, pool, occ, assigned);
    delete Last; delete Cur; return r;
}
int ref_track_lines(const cv::KeyPoint* kp1, int n1kp, const KeyLine* kl1, const uint8_t* ld1, int nl1, const uint8_t* has1, const uint8_t* bad1, const uint8_t* outlier1, const int32_t* nObs1, const double* wl1,
                    const KeyLine* kl2, const uint8_t* ld2, int nl2, const uint8_t* state2, const float* bounds, const float* scale8,
                    const float* cam, const float* TL, const float* TC, float th, int bMono, float nnratio, int32_t* assigned) {
    Frame* Last = new Frame(); Frame* Cur = new Frame();
    Frame::mnMinX = bounds[0]; Frame::mnMaxX = bounds[1]; Frame::mnMinY = bounds[2]; Frame::mnMaxY = bounds[3];
    Last->mvKeys.assign(kp1, kp1 + n1kp);          // LastFrame.mvKeys[i].octave with i a LINE index: the reference's own expression (src/LSDmatcher.cpp:80)
    Last->NL = nl1; Cur->NL = nl2; Cur->mvKeylinesUn.assign(kl2, kl2 + nl2); Cur->mvScaleFactors.assign(scale8, scale8 + 8);
    Cur->mLdesc.create(nl2, 32, CV_8UC1); if (nl2) std::memcpy(Cur->mLdesc.data, ld2, (size_t)nl2 * 32);
    Cur->fx = cam[0]; Cur->fy = cam[1]; Cur->cx = cam[2]; Cur->cy = cam[3]; Cur->mbf = cam[4]; Cur->mb = cam[5];
    Cur->mTcw = cv::Mat(4, 4, CV_32F); Last->mTcw = cv::Mat(4, 4, CV_32F);
    std::memcpy(Cur->mTcw.data, TC, 64); std::memcpy(Last->mTcw.data, TL, 64);
    std::vector<StructureSLAM::MapLine> pool(nl1), occ(2); occ[0].nObs = 1; occ[1].nObs = 0;
    Last->mvpMapLines.assign(nl1, nullptr); Last->mvbLineOutlier.assign(nl1, false);
    for (int i = 0; i < nl1; ++i) {
        pool[i].nObs = nObs1[i]; pool[i].bad = bad1[i] != 0; pool[i].desc = cv::Mat(1, 32, CV_8UC1, (void*)(ld1 + (size_t)i * 32));
        for (int k = 0; k < 6; ++k) pool[i].worldPos.v[k] = wl1[6 * (size_t)i + k];
        if (has1[i]) Last->mvpMapLines[i] = &pool[i];
        Last->mvbLineOutlier[i] = outlier1[i] != 0;
    }
    Cur->mvpMapLines.assign(nl2, nullptr);
    for (int i = 0; i < nl2; ++i) if (state2[i]) Cur->mvpMapLines[i] = &occ[state2[i] - 1];
    StructureSLAM::LSDmatcher m(nnratio, true);
    const int r = m.SearchByProjection(*Cur, *Last, th, bMono != 0);
    encode(Cur->mvpMapLines, pool, occ, assigned);
    delete Last; delete Cur; return r;
}

// ---- SearchByBoW (src/ORBmatcher.cc:159-291 keyframe -> frame, :525-658 keyframe -> keyframe) over the reference's own DBoW2::FeatureVector ----
// node1[i] / node2[i]: the vocabulary node of feature i (the FeatureVector groups features by node, in ascending feature order: FeatureVector::addFeature)
static void fill_kf(StructureSLAM::KeyFrame& K, std::vector<StructureSLAM::MapPoint>& pool, const cv::KeyPoint* kp, const uint8_t* desc, int n, const int32_t* node, const uint8_t* valid) {
    K.mvKeysUn.assign(kp, kp + n); K.mDescriptors.create(n, 32, CV_8UC1); if (n) std::memcpy(K.mDescriptors.data, desc, (size_t)n * 32);
    pool.resize(n); K.mvpMapPoints.assign(n, nullptr);
    for (int i = 0; i < n; ++i) { K.mFeatVec.addFeature((unsigned)node[i], (unsigned)i); pool[i].bad = valid[i] == 2; if (valid[i]) K.mvpMapPoints[i] = &pool[i]; }      // valid: 0 none, 1 good, 2 bad
}
int ref_search_by_bow(const cv::KeyPoint* kpKF, const uint8_t* dKF, int nKF, const int32_t* nodeKF, const uint8_t* validKF,
                      const cv::KeyPoint* kpF, const uint8_t* dF, int nF, const int32_t* nodeF, float nnratio, int checkOri, int32_t* assigned) {
    StructureSLAM::KeyFrame K; std::vector<StructureSLAM::MapPoint> pool; fill_kf(K, pool, kpKF, dKF, nKF, nodeKF, validKF);
    Frame* F = new Frame(); F->N = nF; F->mvKeys.assign(kpF, kpF + nF); F->mvKeysUn = F->mvKeys;
    F->mDescriptors.create(nF, 32, CV_8UC1); if (nF) std::memcpy(F->mDescriptors.data, dF, (size_t)nF * 32);
    for (int i = 0; i < nF; ++i) F->mFeatVec.addFeature((unsigned)nodeF[i], (unsigned)i);
    std::vector<StructureSLAM::MapPoint*> m;
    StructureSLAM::ORBmatcher matcher(nnratio, checkOri != 0);
    const int r = matcher.SearchByBoW(&K, *F, m);
    for (int i = 0; i < nF; ++i) assigned[i] = m[i] ? (int32_t)(m[i] - pool.data()) : -1;
    delete F; return r;
}
int ref_search_by_bow_keyframes(const cv::KeyPoint* kp1, const uint8_t* d1, int n1, const int32_t* node1, const uint8_t* valid1,
                                const cv::KeyPoint* kp2, const uint8_t* d2, int n2, const int32_t* node2, const uint8_t* valid2, float nnratio, int checkOri, int32_t* m12) {
    StructureSLAM::KeyFrame K1, K2; std::vector<StructureSLAM::MapPoint> p1, p2; fill_kf(K1, p1, kp1, d1, n1, node1, valid1); fill_kf(K2, p2, kp2, d2, n2, node2, valid2);
    std::vector<StructureSLAM::MapPoint*> m;
    StructureSLAM::ORBmatcher matcher(nnratio, checkOri != 0);
    const int r = matcher.SearchByBoW(&K1, &K2, m);
    for (int i = 0; i < n1; ++i) m12[i] = m[i] ? (int32_t)(m[i] - p2.data()) : -1;
    return r;
}

// ---- LineSegment::ExtractLineSegment (src/ExtractLineSegment.cpp:18-69): detect, keep the 40 strongest (std::sort by response), describe, line equations ----
int ref_extract_line_segment(const uint8_t* gray, int w, int h, KeyLine* kl_out, uint8_t* ldesc_out, double* fn_out, int cap) {
    cv::Mat img(h, w, CV_8UC1, (void*)gray);
    std::vector<KeyLine> kl; cv::Mat ld; std::vector<Vector3d> fn;
    StructureSLAM::LineSegment* seg = nullptr;                  // the reference calls it through a never-constructed object (include/Frame.h:125)
    StructureSLAM::LineSegment dummy; seg = &dummy;
    seg->ExtractLineSegment(img, kl, ld, fn);
    const int n = std::min((int)kl.size(), cap);
    for (int i = 0; i < n; ++i) { kl_out[i] = kl[i]; std::memcpy(ldesc_out + (size_t)i * 32, ld.ptr(i), 32); fn_out[3 * i] = fn[i](0); fn_out[3 * i + 1] = fn[i](1); fn_out[3 * i + 2] = fn[i](2); }
    return (int)kl.size();
}

// ---- the keyframe-side line matchers: knn-2 + ratio gate (src/LSDmatcher.cpp:143-183, 286-324), knn-2 + 0.5 MAD gate (:326-362), + 0.1 MAD gate (:382-415) ----
// has1 / has2: the keyframe's line i holds a map line.  which: 0 SearchByProjection(KF, F), 1 SearchByDescriptor(KF, F), 2 SearchByDescriptor(KF, KF2), 3 SearchForTriangulation
int ref_line_keyframe_match(int which, const uint8_t* l1, int n1, const uint8_t* has1, const uint8_t* l2, int n2, const uint8_t* has2, int32_t* out, int cap) {
    StructureSLAM::KeyFrame K1, K2; Frame* F = new Frame();
    std::vector<StructureSLAM::MapLine> p1(n1), p2(n2);
    K1.mLineDescriptors.create(n1, 32, CV_8UC1); if (n1) std::memcpy(K1.mLineDescriptors.data, l1, (size_t)n1 * 32);
    K2.mLineDescriptors.create(n2, 32, CV_8UC1); if (n2) std::memcpy(K2.mLineDescriptors.data, l2, (size_t)n2 * 32);
    F->NL = n2; F->mLdesc = K2.mLineDescriptors;
    K1.mvpMapLines.assign(n1, nullptr); K2.mvpMapLines.assign(n2, nullptr);
    for (int i = 0; i < n1; ++i) if (has1[i]) K1.mvpMapLines[i] = &p1[i];
    for (int i = 0; i < n2; ++i) if (has2[i]) K2.mvpMapLines[i] = &p2[i];
    StructureSLAM::LSDmatcher m;
    std::vector<StructureSLAM::MapLine*> res; int r = 0;
    if (which == 3) {
        std::vector<std::pair<size_t, size_t> > pairs;
        r = m.SearchForTriangulation(&K1, &K2, pairs);
        for (int i = 0; i < (int)pairs.size() && i < cap / 2; ++i) { out[2 * i] = (int32_t)pairs[i].first; out[2 * i + 1] = (int32_t)pairs[i].second; }
        delete F; return (int)pairs.size() == r ? r : -1;
    }
    if (which == 0) r = m.SearchByProjection(&K1, *F, res);
    else if (which == 1) r = m.SearchByDescriptor(&K1, *F, res);
    else r = m.SearchByDescriptor(&K1, &K2, res);
    for (int i = 0; i < (int)res.size() && i < cap; ++i) out[i] = !res[i] ? -1 : which == 2 ? (int32_t)(res[i] - p2.data()) : (int32_t)(res[i] - p1.data());
    delete F; return r;
}

// ---- ORBmatcher::SearchForTriangulation + CheckDistEpipolarLine (src/ORBmatcher.cc:660-826, 140-157) ----
// free1 / free2: the keypoint holds no map point yet.  T2 = pKF2's pose (4 x 4 float), C1 = pKF1's camera centre (3 floats), cam2 = {fx, fy, cx, cy}
int ref_search_for_triangulation(const cv::KeyPoint* kp1, const uint8_t* d1, int n1, const int32_t* node1, const uint8_t* free1, const float* ur1,
                                 const cv::KeyPoint* kp2, const uint8_t* d2, int n2, const int32_t* node2, const uint8_t* free2, const float* ur2,
                                 const float* F12, const float* T2, const float* C1, const float* cam2, const float* scale8, const float* sigma8,
                                 int onlyStereo, int checkOri, int32_t* m12) {
    StructureSLAM::KeyFrame K1, K2; std::vector<StructureSLAM::MapPoint> p1, p2;
    std::vector<uint8_t> v1(n1), v2(n2); for (int i = 0; i < n1; ++i) v1[i] = free1[i] ? 0 : 1; for (int i = 0; i < n2; ++i) v2[i] = free2[i] ? 0 : 1;
    fill_kf(K1, p1, kp1, d1, n1, node1, v1.data()); fill_kf(K2, p2, kp2, d2, n2, node2, v2.data());
    K1.N = n1; K2.N = n2; K1.mvuRight.assign(ur1, ur1 + n1); K2.mvuRight.assign(ur2, ur2 + n2);
    K2.fx = cam2[0]; K2.fy = cam2[1]; K2.cx = cam2[2]; K2.cy = cam2[3];
    K2.mvScaleFactors.assign(scale8, scale8 + 8); K2.mvLevelSigma2.assign(sigma8, sigma8 + 8);
    K2.Tcw = cv::Mat(4, 4, CV_32F); std::memcpy(K2.Tcw.data, T2, 64);
    K1.Ow = cv::Mat(3, 1, CV_32F); std::memcpy(K1.Ow.data, C1, 12);
    cv::Mat F(3, 3, CV_32F); std::memcpy(F.data, F12, 36);
    std::vector<std::pair<size_t, size_t> > pairs;
    StructureSLAM::ORBmatcher m(0.6f, checkOri != 0);
    const int r = m.SearchForTriangulation(&K1, &K2, F, pairs, onlyStereo != 0);
    for (int i = 0; i < n1; ++i) m12[i] = -1;
    for (auto& pr : pairs) m12[pr.first] = (int32_t)pr.second;
    return (int)pairs.size() == r ? r : -1;
}

// ---- MapPoint / MapLine::ComputeDistinctiveDescriptors (src/MapPoint.cc:247-312, src/MapLine.cpp:246-317) ----
// the observation set: rows of `desc`, one per keyframe; bad[i]: that keyframe isBad().  The reference walks a std::map keyed by KeyFrame*, i.e. in
// ADDRESS order (which of several equally good descriptors wins depends on the allocator in the author's binary); the keyframes here are elements of one
// array, so address order = row order.  Returns the row of the chosen descriptor among the rows of keyframes that are not bad, -1 when nothing was chosen.
int ref_distinctive(int lines, const uint8_t* desc, int n, const uint8_t* bad) {
    std::vector<StructureSLAM::KeyFrame> kfs(n);
    std::map<StructureSLAM::KeyFrame*, size_t> obs;
    for (int i = 0; i < n; ++i) {
        cv::Mat& D = lines ? kfs[i].mLineDescriptors : kfs[i].mDescriptors;
        D.create(3, 32, CV_8UC1); std::memset(D.data, 0xA5, 96); std::memcpy(D.ptr(1), desc + (size_t)i * 32, 32);      // the observed feature is row 1 of that keyframe
        kfs[i].bad = bad && bad[i]; obs[&kfs[i]] = 1;
    }
    cv::Mat chosen;
    if (lines) { StructureSLAM::MapLine m; m.mObservations = obs; m.ComputeDistinctiveDescriptors(); chosen = m.mLDescriptor; }
    else { StructureSLAM::MapPoint m; m.mObservations = obs; m.ComputeDistinctiveDescriptors(); chosen = m.mDescriptor; }
    if (chosen.empty()) return -1;
    int row = 0;
    for (int i = 0; i < n; ++i) { if (kfs[i].bad) continue; if (std::memcmp(chosen.data, desc + (size_t)i * 32, 32) == 0) return row; ++row; }
    return -2;
}
}
