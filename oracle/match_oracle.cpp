// ORACLE — TEST INFRASTRUCTURE ONLY (see cvleaf.h header).
// PINNED (round 4, `make -C oracle/ref_pin pin-stub`, oracle/ref_pin/compare_slices.py): the reference's own bodies -- cut by line range out of
// src/ORBmatcher.cc, src/LSDmatcher.cpp, src/Frame.cc, src/ExtractLineSegment.cpp and compiled here against stand-in Frame / KeyFrame / MapPoint /
// MapLine types (and the reference's own vendored DBoW2::FeatureVector) -- return what this file's restatements return on the same arrays:
// DescriptorDistance, ComputeThreeMaxima, SearchForInitialization, AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea / GetLinesInArea,
// lineDescriptorMAD, LSDmatcher::SerachForInitialize, ORBmatcher::SearchByProjection(F, MapPoints) and (Cur, Last, th, bMono),
// LSDmatcher::SearchByProjection(F, MapLines), (Cur, Last, th, bMono) and (KF, F), LSDmatcher::SearchByDescriptor (KF, F) and (KF, KF),
// LSDmatcher::SearchForTriangulation over KeyFrame::lineDescriptorMAD, ORBmatcher::SearchByBoW(KF, F) and (KF, KF),
// ORBmatcher::SearchForTriangulation with CheckDistEpipolarLine, MapPoint / MapLine::ComputeDistinctiveDescriptors, the vocabulary loader + transform below against the
// reference's own vendored DBoW2 compiled whole (oracle/_ref/libref_dbow2.so; the two places where that code reads uninitialised locals are decisions D9 / D10), the
// orchestration of LineSegment::ExtractLineSegment -- and, since the end of round 4, the mapping / relocalisation / loop-closing overloads as well: ORBmatcher::Fuse x2
// (:828-978, :980-1103), SearchByProjection(Cur, KF, found, th, dist) (:1475-1602) and (KF, Scw, points, matched, th) (:293-406), SearchBySim3 (:1105-1329);
// LSDmatcher::Fuse x2 (src/LSDmatcher.cpp:417-548, 931-1063), SearchByProjection(KF, Scw, lines, matched, th) (:558-683), SearchBySim3 (:685-929); over
// KeyFrame::GetFeaturesInArea / GetLinesInArea / IsInImage and MapPoint / MapLine::PredictScale / Get*DistanceInvariance.  For those the restatements below
// (fuse_search, search_by_projection_reloc / _sim3) take the search windows; the windows are the ones the reference's own projection blocks form on the same
// stand-in objects (ref_*_queries in oracle/ref_pin/ref_slices_api.cpp).  Every public function of ORBmatcher and LSDmatcher is compared with its reference body.
// NOT pinned: the OpenCV leaves (cv::BFMatcher::knnMatch's tie-break; cv::gemm's accumulation order, Mat::dot, cv::norm and the MatExpr scalings in the pose
// algebra: the stand-ins of oracle/ref_pin/stub_cv state what they assume).
//
// CPU restatement of the reference's Hamming matchers on the hot path:
//   ORBmatcher::DescriptorDistance      src/ORBmatcher.cc:1650-1666
//   ORBmatcher::SearchForInitialization src/ORBmatcher.cc:408-523
//   ORBmatcher::ComputeThreeMaxima      src/ORBmatcher.cc:1604-1645
//   Frame::AssignFeaturesToGrid/PosInGrid/GetFeaturesInArea  src/Frame.cc:133-148,462-472,368-421
//   cv::BFMatcher(NORM_HAMMING,false).knnMatch(q,t,m,2)      (OpenCV batch_distance.cpp, A.10)
//   Frame::lineDescriptorMAD            src/Frame.cc:190-215
//   LSDmatcher::SerachForInitialize / SearchByProjection(KF,F) / SearchForTriangulation gates
//                                       src/LSDmatcher.cpp:143-183,257-284,382-415
#include "oracle.h"
#include <vector>
#include <cmath>
#include <climits>
#include <algorithm>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <map>

namespace orc {

static const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;      // src/ORBmatcher.cc:37-39
static const int GRID_COLS = 64, GRID_ROWS = 48;                      // include/Frame.h:45-46

struct KPm { float x, y, size, angle, response; int octave, class_id; };

static int descriptor_distance(const uint8_t* a, const uint8_t* b) {
    const int32_t* pa = (const int32_t*)a; const int32_t* pb = (const int32_t*)b;
    int dist = 0;
    for (int i = 0; i < 8; ++i, ++pa, ++pb) {
        unsigned v = *pa ^ *pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// knnMatch(..., 2): ascending distance, ties -> lower train index
static void knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist) {
    for (int i = 0; i < nq; ++i) {
        int bd[2] = {INT_MAX, INT_MAX}, bi[2] = {-1, -1};
        for (int j = 0; j < nt; ++j) {
            int d = descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
            if (d < bd[1]) {
                int k = 0;
                if (d < bd[0]) { bd[1] = bd[0]; bi[1] = bi[0]; k = 0; } else k = 1;
                bd[k] = d; bi[k] = j;
            }
        }
        for (int k = 0; k < 2; ++k) { idx[i * 2 + k] = bi[k]; dist[i * 2 + k] = bi[k] >= 0 ? bd[k] : -1; }
    }
}

struct FrameGrid {
    float minX, maxX, minY, maxY, invW, invH;
    std::vector<int> cells[GRID_COLS][GRID_ROWS];
    const KPm* kps; int n;
    void build(const KPm* k, int n_, const float b[4]) {
        kps = k; n = n_;
        minX = b[0]; maxX = b[1]; minY = b[2]; maxY = b[3];
        invW = (float)GRID_COLS / (maxX - minX);               // src/Frame.cc:108-109
        invH = (float)GRID_ROWS / (maxY - minY);
        for (int i = 0; i < n; ++i) {
            int px = (int)std::round((k[i].x - minX) * invW), py = (int)std::round((k[i].y - minY) * invH);
            if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) continue;
            cells[px][py].push_back(i);
        }
    }
    std::vector<int> in_area(float x, float y, float r, int minLevel, int maxLevel) const {
        std::vector<int> out;
        const int nMinCellX = std::max(0, (int)std::floor((x - minX - r) * invW));
        if (nMinCellX >= GRID_COLS) return out;
        const int nMaxCellX = std::min(GRID_COLS - 1, (int)std::ceil((x - minX + r) * invW));
        if (nMaxCellX < 0) return out;
        const int nMinCellY = std::max(0, (int)std::floor((y - minY - r) * invH));
        if (nMinCellY >= GRID_ROWS) return out;
        const int nMaxCellY = std::min(GRID_ROWS - 1, (int)std::ceil((y - minY + r) * invH));
        if (nMaxCellY < 0) return out;
        const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ++ix)
            for (int iy = nMinCellY; iy <= nMaxCellY; ++iy)
                for (int j : cells[ix][iy]) {
                    const KPm& kp = kps[j];
                    if (checkLevels) {
                        if (kp.octave < minLevel) continue;
                        if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                    }
                    const float dx = kp.x - x, dy = kp.y - y;
                    if (std::fabs(dx) < r && std::fabs(dy) < r) out.push_back(j);
                }
        return out;
    }
};

static void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; ++i) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) ind3 = -1;
}

static int search_for_initialization(const KPm* kp1, const uint8_t* d1, int n1, const KPm* kp2, const uint8_t* d2, int n2,
                                     float* prevMatched, int32_t* m12, int windowSize, float nnratio, bool checkOri,
                                     const float bounds[4]) {
    int nmatches = 0;
    for (int i = 0; i < n1; ++i) m12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> matchedDist(n2, INT_MAX), m21(n2, -1);
    FrameGrid* g = new FrameGrid();
    g->build(kp2, n2, bounds);
    for (int i1 = 0; i1 < n1; ++i1) {
        const KPm& k1 = kp1[i1];
        int level1 = k1.octave;
        if (level1 > 0) continue;
        std::vector<int> ind2 = g->in_area(prevMatched[i1 * 2], prevMatched[i1 * 2 + 1], (float)windowSize, level1, level1);
        if (ind2.empty()) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int i2 : ind2) {
            int dist = descriptor_distance(d1 + (size_t)i1 * 32, d2 + (size_t)i2 * 32);
            if (matchedDist[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (m21[bestIdx2] >= 0) { m12[m21[bestIdx2]] = -1; nmatches--; }
                m12[i1] = bestIdx2; m21[bestIdx2] = i1; matchedDist[bestIdx2] = bestDist; nmatches++;
                if (checkOri) {
                    float rot = kp1[i1].angle - kp2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i]) if (m12[idx1] >= 0) { m12[idx1] = -1; nmatches--; }
        }
    }
    for (int i1 = 0; i1 < n1; ++i1)
        if (m12[i1] >= 0) { prevMatched[i1 * 2] = kp2[m12[i1]].x; prevMatched[i1 * 2 + 1] = kp2[m12[i1]].y; }
    delete g;
    return nmatches;
}

// Frame::lineDescriptorMAD on knn-2 results (only medians are consumed, so std::sort's
// instability is harmless)
static void line_descriptor_mad(const int32_t* dist, int n, double& nn_mad, double& nn12_mad) {
    std::vector<float> a(n);
    for (int i = 0; i < n; ++i) a[i] = (float)dist[i * 2];
    std::sort(a.begin(), a.end());
    double med = a[n / 2];
    for (int i = 0; i < n; ++i) a[i] = fabsf((float)(a[i] - med));
    std::sort(a.begin(), a.end());
    nn_mad = 1.4826 * a[n / 2];
    std::vector<float> g(n);
    for (int i = 0; i < n; ++i) g[i] = (float)dist[i * 2 + 1] - (float)dist[i * 2];
    std::sort(g.begin(), g.end(), [](float x, float y) { return x > y; });
    double med12 = g[n / 2];
    for (int i = 0; i < n; ++i) g[i] = fabsf((float)(g[i] - med12));
    std::sort(g.begin(), g.end());
    nn12_mad = 1.4826 * g[n / 2];
}

// Frame::GetLinesInArea, src/Frame.cc:423-460 (linear scan, index order)
struct KLm { float angle; int class_id, octave; float pt_x, pt_y, response, size, sx, sy, ex, ey, sxo, syo, exo, eyo, len; int npix; };
static std::vector<int> lines_in_area(const KLm* kl, int n, float x1, float y1, float x2, float y2, float r, int minLevel, int maxLevel) {
    std::vector<int> out;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel > 0);
    for (int i = 0; i < n; ++i) {
        const KLm& k = kl[i];
        float distance = (float)((0.5 * (x1 + x2) - k.pt_x) * (0.5 * (x1 + x2) - k.pt_x) + (0.5 * (y1 + y2) - k.pt_y) * (0.5 * (y1 + y2) - k.pt_y));
        if (distance > r * r) continue;
        float slope = (y1 - y2) / (x1 - x2) - k.angle;
        if (slope > r * 0.01) continue;
        if (bCheckLevels) {
            if (k.octave < minLevel) continue;
            if (maxLevel >= 0 && k.octave > maxLevel) continue;
        }
        out.push_back(i);
    }
    return out;
}

struct ProjQuery { float u, v, u2, v2, radius; int minLevel, maxLevel; float angle, ur; int valid, obsPositive; };

// The window/projection matcher family, restated once:
//   kind 0, mode 0: ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th)      src/ORBmatcher.cc:45-129
//   kind 0, mode 1: ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)      src/ORBmatcher.cc:1331-1473
//   kind 1, mode 0: LSDmatcher::SearchByProjection(Frame&, vector<MapLine*>&, th)        src/LSDmatcher.cpp:185-255
//                   LSDmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)      src/LSDmatcher.cpp:22-141
// The projection / visibility tests (tracker state) stay with the caller, which passes one ProjQuery per
// map point / line: window centre (or projected endpoints), radius, level range, validity.
static int search_by_projection(int kind, int mode, const void* feats, const uint8_t* desc, int n, const float bounds[4],
                                const float* uright, const uint8_t* occupiedIn, const ProjQuery* q, const uint8_t* qdesc, int nq,
                                float nnratio, int thDist, bool checkOri, int32_t* assigned) {
    int nmatches = 0;
    std::vector<uint8_t> occ(occupiedIn, occupiedIn + n);
    for (int i = 0; i < n; ++i) assigned[i] = -1;
    const KPm* kps = (const KPm*)feats; const KLm* kls = (const KLm*)feats;
    FrameGrid* g = nullptr;
    if (kind == 0) { g = new FrameGrid(); g->build(kps, n, bounds); }
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int iq = 0; iq < nq; ++iq) {
        const ProjQuery& Q = q[iq];
        if (!Q.valid) continue;
        std::vector<int> ind = kind == 0 ? g->in_area(Q.u, Q.v, Q.radius, Q.minLevel, Q.maxLevel)
                                         : lines_in_area(kls, n, Q.u, Q.v, Q.u2, Q.v2, Q.radius, Q.minLevel, Q.maxLevel);
        if (ind.empty()) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : ind) {
            if (occ[idx]) continue;
            if (kind == 0 && uright && uright[idx] > 0) {
                const float er = std::fabs(Q.ur - uright[idx]);
                if (er > Q.radius) continue;
            }
            const int dist = descriptor_distance(qdesc + (size_t)iq * 32, desc + (size_t)idx * 32);
            const int oct = kind == 0 ? kps[idx].octave : kls[idx].octave;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = oct; bestIdx = idx; }
            else if (mode == 0 && dist < bestDist2) { bestLevel2 = oct; bestDist2 = dist; }
        }
        if (bestDist <= thDist) {
            if (mode == 0 && bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            assigned[bestIdx] = iq; occ[bestIdx] = Q.obsPositive ? 1 : 0; nmatches++;
            if (mode == 1 && checkOri) {
                float rot = Q.angle - kps[bestIdx].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx);
            }
        }
    }
    if (mode == 1 && checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int idx : rotHist[i]) { assigned[idx] = -2; nmatches--; }      // CurrentFrame.mvpMapPoints[rotHist[i][j]] = NULL (src/ORBmatcher.cc:1465, :1596): matched and removed, not "untouched"
    }
    delete g;
    return nmatches;
}

// The candidate search inside the Fuse family, one independent best-match per projected map point / line:
//   kind 0, chi2 1: ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th)          src/ORBmatcher.cc:828-960  (:897-948)
//   kind 0, chi2 0: ORBmatcher::Fuse(KeyFrame*, cv::Mat Scw, ..., th, vpReplacePoint)  src/ORBmatcher.cc:962-1103 (:1055-1080)
//   kind 1        : LSDmatcher::Fuse(KeyFrame*, const vector<MapLine*>&, th)           src/LSDmatcher.cpp:417-548 (:497-523)
// KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:610-649) has no level filter; the level test [pred-1, pred] is Fuse's own.
// The projection / visibility tests and the Replace / AddObservation bookkeeping on (bestIdx, bestDist) stay with the caller.
static void fuse_search(int kind, int chi2, const void* feats, const uint8_t* desc, int n, const float bounds[4], const float* uright,
                        const float* invLevelSigma2, const ProjQuery* q, const uint8_t* qdesc, int nq, int32_t* bestIdxOut, int32_t* bestDistOut) {
    const KPm* kps = (const KPm*)feats; const KLm* kls = (const KLm*)feats;
    FrameGrid* g = nullptr;
    if (kind == 0) { g = new FrameGrid(); g->build(kps, n, bounds); }
    for (int iq = 0; iq < nq; ++iq) {
        const ProjQuery& Q = q[iq];
        bestIdxOut[iq] = -1; bestDistOut[iq] = INT_MAX;
        if (!Q.valid) continue;
        std::vector<int> ind = kind == 0 ? g->in_area(Q.u, Q.v, Q.radius, -1, -1) : lines_in_area(kls, n, Q.u, Q.v, Q.u2, Q.v2, Q.radius, -1, -1);
        int bestDist = INT_MAX, bestIdx = -1;
        for (int idx : ind) {
            const int lvl = kind == 0 ? kps[idx].octave : kls[idx].octave;
            if (lvl < Q.minLevel || lvl > Q.maxLevel) continue;
            if (kind == 0 && chi2) {
                const float kpx = kps[idx].x, kpy = kps[idx].y;
                const float ex = Q.u - kpx, ey = Q.v - kpy;
                if (uright && uright[idx] >= 0) {
                    const float er = Q.ur - uright[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * invLevelSigma2[lvl] > 7.8) continue;
                } else {
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * invLevelSigma2[lvl] > 5.99) continue;
                }
            }
            const int dist = descriptor_distance(qdesc + (size_t)iq * 32, desc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        bestIdxOut[iq] = bestIdx; bestDistOut[iq] = bestDist;
    }
    delete g;
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&), src/ORBmatcher.cc:159-291.  The DBoW2 vocabulary
// (an LFS pointer in the reference tree) only decides WHICH features share a node; the matcher consumes the two
// FeatureVectors, passed here as CSR lists over the shared nodes in ascending node id (the merge walk of :188-253).
static int search_by_bow(const KPm* kpKF, const uint8_t* dKF, const uint8_t* validKF, const KPm* kpF, const uint8_t* dF, int nF,
                         const int32_t* ptrKF, const int32_t* ptrF, int nnodes, const int32_t* idxKF, const int32_t* idxF,
                         float nnratio, bool checkOri, int32_t* assigned) {
    int nmatches = 0;
    for (int i = 0; i < nF; ++i) assigned[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int nd = 0; nd < nnodes; ++nd) {
        for (int a = ptrKF[nd]; a < ptrKF[nd + 1]; ++a) {
            const int realIdxKF = idxKF[a];
            if (!validKF[realIdxKF]) continue;                         // !pMP || pMP->isBad()
            int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
            for (int b = ptrF[nd]; b < ptrF[nd + 1]; ++b) {
                const int realIdxF = idxF[b];
                if (assigned[realIdxF] >= 0) continue;
                const int dist = descriptor_distance(dKF + (size_t)realIdxKF * 32, dF + (size_t)realIdxF * 32);
                if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                else if (dist < bestDist2) bestDist2 = dist;
            }
            if (bestDist1 <= TH_LOW && (float)bestDist1 < nnratio * (float)bestDist2) {
                assigned[bestIdxF] = realIdxKF;
                if (checkOri) {
                    float rot = kpKF[realIdxKF].angle - kpF[bestIdxF].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(bestIdxF);
                }
                nmatches++;
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) { assigned[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

}  // namespace orc

using namespace orc;

extern "C" {

// MapPoint::ComputeDistinctiveDescriptors, src/MapPoint.cc:276-306 (MapLine: src/MapLine.cpp:280-311): N x N Hamming
// distances, per row sort and take sorted[0.5*(N-1)] (index truncated to int), first row with the smallest median.
// The MapLine variant fills the matrix with cv::norm(NORM_HAMMING), the same popcount distance.
static int distinctive_index(const uint8_t* d, int n) {
    if (n <= 0) return -1;
    std::vector<std::vector<int>> D(n, std::vector<int>(n, 0));
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) { const int v = descriptor_distance(d + (size_t)i * 32, d + (size_t)j * 32); D[i][j] = v; D[j][i] = v; }
    int bestMedian = INT_MAX, bestIdx = 0;
    for (int i = 0; i < n; ++i) {
        std::vector<int> v(D[i]);
        std::sort(v.begin(), v.end());
        const int median = v[(size_t)(0.5 * (n - 1))];
        if (median < bestMedian) { bestMedian = median; bestIdx = i; }
    }
    return bestIdx;
}
int orc_distinctive(const uint8_t* desc, const int32_t* ptr, int nsets, int32_t* best) {
    for (int s = 0; s < nsets; ++s) best[s] = distinctive_index(desc + (size_t)ptr[s] * 32, ptr[s + 1] - ptr[s]);
    return 0;
}

int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }

int orc_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist) { knn2(q, nq, t, nt, idx, dist); return 0; }

int orc_hamming_matrix(const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* D) {
    for (int i = 0; i < nq; ++i) for (int j = 0; j < nt; ++j) D[(size_t)i * nt + j] = (uint16_t)descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
    return 0;
}

int orc_search_for_initialization(const void* kp1, const uint8_t* d1, int n1, const void* kp2, const uint8_t* d2, int n2,
                                  float* prev_matched, int32_t* m12, int window, float nnratio, int check_ori, const float* bounds) {
    return search_for_initialization((const KPm*)kp1, d1, n1, (const KPm*)kp2, d2, n2, prev_matched, m12, window, nnratio, check_ori != 0, bounds);
}

// ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist),
// src/ORBmatcher.cc:1475-1602 (relocalisation), restated on its own: q[i] carries what the projection block (:1499-1530) produces for
// pKF's i-th map point (u, v, radius = th*scale[pred], maxLevel = pred, angle = pKF->mvKeysUn[i].angle, valid = passed every `continue`).
// hasMapPoint[i2] = CurrentFrame.mvpMapPoints[i2] != NULL on entry.  assigned[i2] = i as elsewhere.
int orc_search_by_projection_reloc(const void* feats, const uint8_t* desc, int n, const float* bounds, const uint8_t* hasMapPoint, const void* qv,
                                   const uint8_t* qdesc, int nq, int ORBdist, int check_ori, int32_t* assigned) {
    const KPm* kps = (const KPm*)feats; const ProjQuery* q = (const ProjQuery*)qv;
    FrameGrid g; g.build(kps, n, bounds);
    std::vector<char> mvpMapPoints(hasMapPoint, hasMapPoint + n);
    for (int i = 0; i < n; ++i) assigned[i] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < nq; ++i) {
        if (!q[i].valid) continue;
        const int nPredictedLevel = q[i].maxLevel;
        const std::vector<int> vIndices2 = g.in_area(q[i].u, q[i].v, q[i].radius, nPredictedLevel - 1, nPredictedLevel + 1);
        if (vIndices2.empty()) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            if (mvpMapPoints[i2]) continue;
            const int dist = descriptor_distance(qdesc + (size_t)i * 32, desc + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= ORBdist) {
            mvpMapPoints[bestIdx2] = 1; assigned[bestIdx2] = i; nmatches++;
            if (check_ori) {
                float rot = q[i].angle - kps[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_ori) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int idx : rotHist[i]) { assigned[idx] = -2; nmatches--; }      // CurrentFrame.mvpMapPoints[rotHist[i][j]] = NULL (src/ORBmatcher.cc:1465, :1596): matched and removed, not "untouched"
    }
    return nmatches;
}

// ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th), src/ORBmatcher.cc:293-406 (kind 0), and
// LSDmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpLines, vpMatched, th), src/LSDmatcher.cpp:558-683 (kind 1): the loop-closing
// window search.  q[i]: u, v (lines: both projected endpoints), radius, maxLevel = nPredictedLevel, valid.  matchedIn[idx] = vpMatched[idx] != NULL
// on entry; assigned[idx] = i stands for vpMatched[idx] = vpPoints[i].
int orc_search_by_projection_sim3(int kind, const void* feats, const uint8_t* desc, int n, const float* bounds, const uint8_t* matchedIn, const void* qv,
                                  const uint8_t* qdesc, int nq, int32_t* assigned) {
    const KPm* kps = (const KPm*)feats; const KLm* kls = (const KLm*)feats; const ProjQuery* q = (const ProjQuery*)qv;
    FrameGrid g; if (kind == 0) g.build(kps, n, bounds);
    std::vector<char> vpMatched(matchedIn, matchedIn + n);
    for (int i = 0; i < n; ++i) assigned[i] = -1;
    int nmatches = 0;
    for (int i = 0; i < nq; ++i) {
        if (!q[i].valid) continue;
        const int nPredictedLevel = q[i].maxLevel;
        const std::vector<int> vIndices = kind == 0 ? g.in_area(q[i].u, q[i].v, q[i].radius, -1, -1)
                                                    : lines_in_area(kls, n, q[i].u, q[i].v, q[i].u2, q[i].v2, q[i].radius, -1, -1);
        if (vIndices.empty()) continue;
        int bestDist = 256, bestIdx = -1;
        for (int idx : vIndices) {
            if (vpMatched[idx]) continue;
            const int level = kind == 0 ? kps[idx].octave : kls[idx].octave;
            if (level < nPredictedLevel - 1 || level > nPredictedLevel) continue;
            const int dist = descriptor_distance(qdesc + (size_t)i * 32, desc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) { vpMatched[bestIdx] = 1; assigned[bestIdx] = i; nmatches++; }
    }
    return nmatches;
}

int orc_search_by_projection(int kind, int mode, const void* feats, const uint8_t* desc, int n, const float* bounds, const float* uright,
                             const uint8_t* occupied, const void* q, const uint8_t* qdesc, int nq, float nnratio, int th_dist, int check_ori,
                             int32_t* assigned) {
    return search_by_projection(kind, mode, feats, desc, n, bounds, uright, occupied, (const ProjQuery*)q, qdesc, nq, nnratio, th_dist, check_ori != 0, assigned);
}

// ORBmatcher::SearchForTriangulation(KeyFrame*, KeyFrame*, cv::Mat F12, vector<pair<size_t,size_t>>&, bOnlyStereo),
// src/ORBmatcher.cc:660-826, with CheckDistEpipolarLine :140-157.  FeatureVectors as CSR lists over the shared nodes
// (ascending node id), free1/free2 = "no MapPoint yet" (:704-708, :725-729).  The reference never sets vbMatched2, so the
// idx1 queries are independent; `dist>bestDist` lets a later candidate with an EQUAL distance replace the current one.
static int search_for_triangulation(const KPm* kp1, const uint8_t* d1, const float* ur1, const uint8_t* free1, int n1,
                                    const KPm* kp2, const uint8_t* d2, const float* ur2, const uint8_t* free2, int n2,
                                    const int32_t* ptr1, const int32_t* ptr2, int nnodes, const int32_t* idx1v, const int32_t* idx2v,
                                    const float* F12, float ex, float ey, const float* scaleFactors2, const float* levelSigma2_2,
                                    bool onlyStereo, bool checkOri, int32_t* m12) {
    int nmatches = 0;
    for (int i = 0; i < n1; ++i) m12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int nd = 0; nd < nnodes; ++nd) {
        for (int a = ptr1[nd]; a < ptr1[nd + 1]; ++a) {
            const int i1 = idx1v[a];
            if (!free1[i1]) continue;
            const bool bStereo1 = ur1 && ur1[i1] >= 0;
            if (onlyStereo && !bStereo1) continue;
            const KPm& k1 = kp1[i1];
            int bestDist = TH_LOW, bestIdx2 = -1;
            for (int b = ptr2[nd]; b < ptr2[nd + 1]; ++b) {
                const int i2 = idx2v[b];
                if (!free2[i2]) continue;
                const bool bStereo2 = ur2 && ur2[i2] >= 0;
                if (onlyStereo && !bStereo2) continue;
                const int dist = descriptor_distance(d1 + (size_t)i1 * 32, d2 + (size_t)i2 * 32);
                if (dist > TH_LOW || dist > bestDist) continue;
                const KPm& k2 = kp2[i2];
                if (!bStereo1 && !bStereo2) {
                    const float distex = ex - k2.x, distey = ey - k2.y;
                    if (distex * distex + distey * distey < 100 * scaleFactors2[k2.octave]) continue;
                }
                // CheckDistEpipolarLine
                const float la = k1.x * F12[0] + k1.y * F12[3] + F12[6];
                const float lb = k1.x * F12[1] + k1.y * F12[4] + F12[7];
                const float lc = k1.x * F12[2] + k1.y * F12[5] + F12[8];
                const float num = la * k2.x + lb * k2.y + lc;
                const float den = la * la + lb * lb;
                if (den == 0) continue;
                const float dsqr = num * num / den;
                if (dsqr < 3.84 * levelSigma2_2[k2.octave]) { bestIdx2 = i2; bestDist = dist; }
            }
            if (bestIdx2 >= 0) {
                m12[i1] = bestIdx2; nmatches++;
                if (checkOri) {
                    float rot = k1.angle - kp2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int i1 : rotHist[i]) { m12[i1] = -1; nmatches--; }
    }
    return nmatches;
}

int orc_search_for_triangulation(const void* kp1, const uint8_t* d1, const float* ur1, const uint8_t* free1, int n1,
                                 const void* kp2, const uint8_t* d2, const float* ur2, const uint8_t* free2, int n2,
                                 const int32_t* ptr1, const int32_t* ptr2, int nnodes, const int32_t* idx1, const int32_t* idx2,
                                 const float* F12, float ex, float ey, const float* scale_factors2, const float* level_sigma2_2,
                                 int only_stereo, int check_ori, int32_t* m12) {
    return search_for_triangulation((const KPm*)kp1, d1, ur1, free1, n1, (const KPm*)kp2, d2, ur2, free2, n2, ptr1, ptr2, nnodes, idx1, idx2,
                                    F12, ex, ey, scale_factors2, level_sigma2_2, only_stereo != 0, check_ori != 0, m12);
}

// TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup), Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1216-1259
// (FORB::distance = Hamming), on a vocabulary given as CSR children lists in DBoW2's node numbering.
int orc_bow_transform(int nnodes, int levels, const int32_t* childPtr, const int32_t* children, const uint8_t* nodeDesc, const int32_t* wordId,
                      const double* weight, const uint8_t* feat, int n, int levelsup, int32_t* wordOut, double* weightOut, int32_t* nodeOut) {
    (void)nnodes;
    for (int i = 0; i < n; ++i) {
        const int nid_level = levels - levelsup;
        int nid = 0;                       // root when nid_level <= 0 (:1227)
        int final_id = 0, current_level = 0;
        do {
            ++current_level;
            const int c0 = childPtr[final_id], c1 = childPtr[final_id + 1];
            final_id = children[c0];
            double best_d = (double)descriptor_distance(feat + (size_t)i * 32, nodeDesc + (size_t)final_id * 32);
            for (int c = c0 + 1; c < c1; ++c) {
                const int id = children[c];
                const double d = (double)descriptor_distance(feat + (size_t)i * 32, nodeDesc + (size_t)id * 32);
                if (d < best_d) { best_d = d; final_id = id; }
            }
            if (current_level == nid_level) nid = final_id;
        } while (childPtr[final_id + 1] > childPtr[final_id]);
        wordOut[i] = wordId[final_id]; weightOut[i] = weight[final_id]; nodeOut[i] = nid;
    }
    return 0;
}

// ---- DBoW2 vocabulary text file + BowVector / FeatureVector (test infrastructure like the rest of this file)
// TemplatedVocabulary::loadFromTextFile, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1423, read with the same stream
// operations.  One decision where the reference is undefined: its `while(!f.eof())` loop runs once more on the empty string after the
// final newline and appends a node from uninitialised locals; here an empty line ends nothing and adds nothing.
namespace {
struct OrcVocab { int k = 0, L = 0, scoring = 0, weighting = 0; std::vector<int32_t> parent, wordId; std::vector<std::vector<int32_t>> children; std::vector<uint8_t> desc; std::vector<double> weight; };
OrcVocab g_vocab;
}
int orc_vocab_load_text(const char* path, int* k, int* L, int* scoring, int* weighting, int* nnodes, int* nwords) {
    std::ifstream f(path);
    if (!f.is_open()) return -1;
    OrcVocab v;
    std::string s;
    std::getline(f, s);
    std::stringstream ss; ss << s;
    int n1 = -1, n2 = -1; v.k = -1; v.L = -1;
    ss >> v.k; ss >> v.L; ss >> n1; ss >> n2;
    if (v.k < 0 || v.k > 20 || v.L < 1 || v.L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) return -2;
    v.scoring = n1; v.weighting = n2;
    v.parent.assign(1, -1); v.wordId.assign(1, 0); v.children.resize(1); v.desc.assign(32, 0); v.weight.assign(1, 0.0);
    int words = 0;
    while (!f.eof()) {
        std::string snode;
        std::getline(f, snode);
        std::stringstream ssnode; ssnode << snode;
        int pid;
        if (!(ssnode >> pid)) continue;
        const int nid = (int)v.parent.size();
        if (pid < 0 || pid >= nid) return -3;
        v.parent.push_back(pid); v.children.resize(nid + 1); v.children[pid].push_back(nid);
        int nIsLeaf = 0; ssnode >> nIsLeaf;
        v.desc.resize(v.desc.size() + 32, 0);
        for (int i = 0; i < 32; ++i) { int n; ssnode >> n; if (!ssnode.fail()) v.desc[(size_t)nid * 32 + i] = (unsigned char)n; }      // FORB::fromString, FORB.cpp:120-135
        double w = 0; ssnode >> w; v.weight.push_back(w);
        if (nIsLeaf > 0) v.wordId.push_back(words++); else v.wordId.push_back(0);
    }
    g_vocab = v;
    *k = v.k; *L = v.L; *scoring = v.scoring; *weighting = v.weighting; *nnodes = (int)v.parent.size(); *nwords = words;
    return 0;
}
// the loaded tree as the CSR arrays orc_bow_transform / sslam_vocab_create take
int orc_vocab_arrays(int32_t* childPtr, int32_t* children, uint8_t* nodeDesc, int32_t* wordId, double* weight) {
    const OrcVocab& v = g_vocab; const int n = (int)v.parent.size();
    int c = 0; childPtr[0] = 0;
    for (int i = 0; i < n; ++i) { for (int ch : v.children[i]) children[c++] = ch; childPtr[i + 1] = c; }
    std::memcpy(nodeDesc, v.desc.data(), v.desc.size()); std::memcpy(wordId, v.wordId.data(), 4 * (size_t)n); std::memcpy(weight, v.weight.data(), 8 * (size_t)n);
    return 0;
}
// TemplatedVocabulary::transform(features, v, fv, levelsup), TemplatedVocabulary.h:1126-1208 with BowVector.cpp:34-85 and
// FeatureVector.cpp:30-44; the two maps come back flattened in key order.
int orc_compute_bow(int nnodes, int levels, const int32_t* childPtr, const int32_t* children, const uint8_t* nodeDesc, const int32_t* wordId, const double* weight,
                    int weighting, int scoring, const uint8_t* feat, int n, int levelsup,
                    int32_t* bowWord, double* bowValue, int32_t* nbow, int32_t* fvNode, int32_t* fvPtr, int32_t* fvFeat, int32_t* nfv) {
    std::vector<int32_t> w(n), nd(n); std::vector<double> wt(n);
    orc_bow_transform(nnodes, levels, childPtr, children, nodeDesc, wordId, weight, feat, n, levelsup, w.data(), wt.data(), nd.data());
    std::map<int32_t, double> v; std::map<int32_t, std::vector<unsigned>> fv;
    const bool must = scoring != 5, l2 = scoring == 1;          // ScoringObject.h:74-89
    if (weighting == 0 || weighting == 1) {                      // TF_IDF, TF
        for (int i = 0; i < n; ++i) if (wt[i] > 0) {
            auto vit = v.lower_bound(w[i]);
            if (vit != v.end() && !(v.key_comp()(w[i], vit->first))) vit->second += wt[i]; else v.insert(vit, std::make_pair(w[i], wt[i]));
            fv[nd[i]].push_back((unsigned)i);
        }
        if (!v.empty() && !must) { const double ndv = (double)v.size(); for (auto& kv : v) kv.second /= ndv; }
    } else {                                                      // IDF, BINARY
        for (int i = 0; i < n; ++i) if (wt[i] > 0) {
            auto vit = v.lower_bound(w[i]);
            if (vit == v.end() || v.key_comp()(w[i], vit->first)) v.insert(vit, std::make_pair(w[i], wt[i]));
            fv[nd[i]].push_back((unsigned)i);
        }
    }
    if (must) {
        double norm = 0.0;
        if (!l2) { for (auto& kv : v) norm += std::fabs(kv.second); }
        else { for (auto& kv : v) norm += kv.second * kv.second; norm = std::sqrt(norm); }
        if (norm > 0.0) for (auto& kv : v) kv.second /= norm;
    }
    int j = 0; for (auto& kv : v) { bowWord[j] = kv.first; bowValue[j] = kv.second; ++j; }
    *nbow = j;
    int a = 0, c = 0; fvPtr[0] = 0;
    for (auto& kv : fv) { fvNode[a] = kv.first; for (unsigned idx : kv.second) fvFeat[c++] = (int32_t)idx; fvPtr[++a] = c; }
    *nfv = a;
    return 0;
}

int orc_fuse_search(int kind, int chi2, const void* feats, const uint8_t* desc, int n, const float* bounds, const float* uright,
                    const float* inv_level_sigma2, const void* q, const uint8_t* qdesc, int nq, int32_t* best_idx, int32_t* best_dist) {
    fuse_search(kind, chi2, feats, desc, n, bounds, uright, inv_level_sigma2, (const ProjQuery*)q, qdesc, nq, best_idx, best_dist);
    return 0;
}

// ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12), src/ORBmatcher.cc:525-658.
// matches12[idx1] = idx2 stands for vpMatches12[idx1] = vpMapPoints2[idx2].
int orc_search_by_bow_keyframes(const void* kp1v, const uint8_t* d1, const uint8_t* valid1, int n1, const void* kp2v, const uint8_t* d2, const uint8_t* valid2, int n2,
                                const int32_t* ptr1, const int32_t* ptr2, int nnodes, const int32_t* idx1s, const int32_t* idx2s, float nnratio, int check_ori,
                                int32_t* matches12) {
    const KPm* kp1 = (const KPm*)kp1v; const KPm* kp2 = (const KPm*)kp2v;
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    std::vector<char> vbMatched2((size_t)n2, 0);
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0;
    for (int nd = 0; nd < nnodes; ++nd) {
        for (int a = ptr1[nd]; a < ptr1[nd + 1]; ++a) {
            const int idx1 = idx1s[a];
            if (!valid1[idx1]) continue;                                    // :561-566
            int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
            for (int b = ptr2[nd]; b < ptr2[nd + 1]; ++b) {
                const int idx2 = idx2s[b];
                if (vbMatched2[idx2] || !valid2[idx2]) continue;            // :576-580
                const int dist = descriptor_distance(d1 + (size_t)idx1 * 32, d2 + (size_t)idx2 * 32);
                if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                else if (dist < bestDist2) bestDist2 = dist;
            }
            if (bestDist1 < TH_LOW) {
                if ((float)bestDist1 < nnratio * (float)bestDist2) {
                    matches12[idx1] = bestIdx2; vbMatched2[bestIdx2] = 1;
                    if (check_ori) {
                        float rot = kp1[idx1].angle - kp2[bestIdx2].angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back(idx1);
                    }
                    nmatches++;
                }
            }
        }
    }
    if (check_ori) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) { matches12[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

int orc_search_by_bow(const void* kpKF, const uint8_t* dKF, const uint8_t* validKF, const void* kpF, const uint8_t* dF, int nF,
                      const int32_t* ptrKF, const int32_t* ptrF, int nnodes, const int32_t* idxKF, const int32_t* idxF, float nnratio, int check_ori,
                      int32_t* assigned) {
    return search_by_bow((const KPm*)kpKF, dKF, validKF, (const KPm*)kpF, dF, nF, ptrKF, ptrF, nnodes, idxKF, idxF, nnratio, check_ori != 0, assigned);
}

// Frame::GetFeaturesInArea / GetLinesInArea on their own (src/Frame.cc:368-421, 423-460): what oracle/ref_pin's slices are compared with
int orc_features_in_area(const void* kps, int n, const float* bounds, float x, float y, float r, int minLevel, int maxLevel, int32_t* out) {
    FrameGrid* g = new FrameGrid(); g->build((const KPm*)kps, n, bounds);
    std::vector<int> v = g->in_area(x, y, r, minLevel, maxLevel);
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    delete g; return (int)v.size();
}
int orc_lines_in_area(const void* kls, int n, float x1, float y1, float x2, float y2, float r, int minLevel, int maxLevel, int32_t* out) {
    std::vector<int> v = lines_in_area((const KLm*)kls, n, x1, y1, x2, y2, r, minLevel, maxLevel);
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return (int)v.size();
}

// LSDmatcher gates.  Degenerate inputs (n1==0 or n2<2) are UB in the reference
// (src/LSDmatcher.cpp:167); defined here (and in the HIP library) as "0 matches".
int orc_line_match(const uint8_t* l1, int n1, const uint8_t* l2, int n2, double gate_scale, int ratio_mode,
                   int32_t* pairs, int cap, double* nn_mad, double* nn12_mad) {
    *nn_mad = 0; *nn12_mad = 0;
    if (n1 <= 0 || n2 < 2) return 0;
    std::vector<int32_t> idx(n1 * 2), dist(n1 * 2);
    knn2(l1, n1, l2, n2, idx.data(), dist.data());
    line_descriptor_mad(dist.data(), n1, *nn_mad, *nn12_mad);
    double th = *nn12_mad * gate_scale;
    const float minRatio = 1.0f / 1.5f;
    int n = 0;
    for (int i = 0; i < n1; ++i) {
        bool ok;
        if (ratio_mode) { double r = (float)dist[i * 2] / (float)dist[i * 2 + 1]; ok = r < minRatio; }
        else { double g = (float)dist[i * 2 + 1] - (float)dist[i * 2]; ok = g > th; }
        if (ok) { if (n < cap) { pairs[n * 2] = i; pairs[n * 2 + 1] = idx[i * 2]; } ++n; }
    }
    return n;
}

}  // extern "C"
