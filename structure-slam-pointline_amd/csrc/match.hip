// Hamming matchers for MI355X (gfx950): brute-force knn-2 (BFMatcher semantics),
// dense distance matrix, ORBmatcher::SearchForInitialization and the LSDmatcher
// knn2+MAD gates — device side of reference src/ORBmatcher.cc:408-523,1604-1666,
// src/LSDmatcher.cpp:143-183,257-284,364-415, src/Frame.cc:133-148,190-215,368-472.
//
// The path is integer/bitwise: v_bcnt popcounts on 8x u32 XORs, wave-wide min
// reductions with the tie-break order carried in the key; no MFMA.
#include "common.h"
#include <algorithm>

using namespace sslam;

namespace {

constexpr int TH_LOW = 50;
constexpr int HISTO_LENGTH = 30;
constexpr int GRID_COLS = 64, GRID_ROWS = 48;

__device__ __forceinline__ int hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// one wave per query; lanes stride the train set (coalesced 32 B per lane).
// key = dist<<32 | idx so the minimum resolves ties to the lower train index (A.10).
__device__ __forceinline__ void wave_knn2(const uint8_t* __restrict__ qd, const uint8_t* __restrict__ t, int nt,
                                          unsigned long long& best, unsigned long long& second) {
    const int lane = threadIdx.x & 63;
    const uint4 q0 = ((const uint4*)qd)[0], q1 = ((const uint4*)qd)[1];
    unsigned long long b = ~0ull, s = ~0ull;
    for (int j = lane; j < nt; j += 64) {
        const uint4* tp = (const uint4*)(t + (size_t)j * 32);
        unsigned long long k = ((unsigned long long)hamming256(q0, q1, tp[0], tp[1]) << 32) | (unsigned)j;
        if (k < b) { s = b; b = k; } else if (k < s) s = k;
    }
    best = wave_min_u64(b);
    unsigned long long c = (b == best) ? s : b;
    second = wave_min_u64(c);
}

__global__ __launch_bounds__(256) void k_knn2(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t, int nt,
                                              int* __restrict__ idx, int* __restrict__ dist) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wave >= nq) return;
    unsigned long long b, s;
    wave_knn2(q + (size_t)wave * 32, t, nt, b, s);
    if ((threadIdx.x & 63) == 0) {
        idx[wave * 2] = b == ~0ull ? -1 : (int)(unsigned)b;
        dist[wave * 2] = b == ~0ull ? -1 : (int)(b >> 32);
        idx[wave * 2 + 1] = s == ~0ull ? -1 : (int)(unsigned)s;
        dist[wave * 2 + 1] = s == ~0ull ? -1 : (int)(s >> 32);
    }
}

// batch form: frame f uses q + f*cap*32 (nq[f] rows) against t + f*cap*32 (nt[f] rows).  The kernel is bound by L2
// traffic on the train rows, so each wave scores FOUR queries against every train row it loads.
__global__ __launch_bounds__(256) void k_knn2_batch(const uint8_t* __restrict__ q, const int* __restrict__ nq, const uint8_t* __restrict__ t,
                                                    const int* __restrict__ nt, int cap, int* __restrict__ idx, int* __restrict__ dist) {
    const int f = blockIdx.y, lane = threadIdx.x & 63;
    const int q0i = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 4;
    const int nqf = nq[f], ntf = nt[f];
    if (q0i >= nqf) return;
    const uint8_t* qb = q + (size_t)f * cap * 32;
    const uint8_t* tb = t + (size_t)f * cap * 32;
    uint4 qa[4], qc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int qi = min(q0i + k, nqf - 1);
        qa[k] = ((const uint4*)(qb + (size_t)qi * 32))[0]; qc[k] = ((const uint4*)(qb + (size_t)qi * 32))[1];
    }
    unsigned long long b[4] = {~0ull, ~0ull, ~0ull, ~0ull}, s[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    for (int j = lane; j < ntf; j += 64) {
        const uint4* tp = (const uint4*)(tb + (size_t)j * 32);
        const uint4 t0 = tp[0], t1 = tp[1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long kk = ((unsigned long long)hamming256(qa[k], qc[k], t0, t1) << 32) | (unsigned)j;
            if (kk < b[k]) { s[k] = b[k]; b[k] = kk; } else if (kk < s[k]) s[k] = kk;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long best = wave_min_u64(b[k]);
        const unsigned long long second = wave_min_u64(b[k] == best ? s[k] : b[k]);
        if (lane == 0 && q0i + k < nqf) {
            const size_t o = ((size_t)f * cap + q0i + k) * 2;
            idx[o] = best == ~0ull ? -1 : (int)(unsigned)best;
            dist[o] = best == ~0ull ? -1 : (int)(best >> 32);
            idx[o + 1] = second == ~0ull ? -1 : (int)(unsigned)second;
            dist[o + 1] = second == ~0ull ? -1 : (int)(second >> 32);
        }
    }
}

__global__ void k_hamming_matrix(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t, int nt, unsigned short* __restrict__ D) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= nt || i >= nq) return;
    const uint4* qp = (const uint4*)(q + (size_t)i * 32);
    const uint4* tp = (const uint4*)(t + (size_t)j * 32);
    D[(size_t)i * nt + j] = (unsigned short)hamming256(qp[0], qp[1], tp[0], tp[1]);
}

// ---------------------------------------------------------------- SearchForInitialization
// One wave per frame pair.  The i1 loop is sequential (vMatchedDistance / un-match
// semantics); the candidate scan of each step is wave-parallel.  The
// GetFeaturesInArea order (grid cell x-major, then y, then index — D.4) is carried in
// the reduction key so "first strictly smaller distance wins" is reproduced.
struct SfiArgs {
    const sslam_keypoint* kp1; const uint8_t* d1; const int* n1;
    const sslam_keypoint* kp2; const uint8_t* d2; const int* n2;
    int cap, n1s, n2s;             // n1s/n2s used when n1/n2 pointers are null
    float* prevMatched; int* m12; int* nmatches;
    int* scratch;                  // per pair: matchedDist[cap], m21[cap], cand[cap], key[cap], bin[cap]
    int window; float nnratio; int checkOri;
    float minX, maxX, minY, maxY;
};

__global__ __launch_bounds__(64) void k_search_init(SfiArgs A) {
    const int p = xcd_mix_frame(blockIdx.x, gridDim.x), lane = threadIdx.x;      // one wave per frame pair: see xcd_mix_frame
    const int n1 = A.n1 ? A.n1[p] : A.n1s, n2 = A.n2 ? A.n2[p] : A.n2s;
    const sslam_keypoint* kp1 = A.kp1 + (size_t)p * A.cap;
    const sslam_keypoint* kp2 = A.kp2 + (size_t)p * A.cap;
    const uint8_t* d1 = A.d1 + (size_t)p * A.cap * 32;
    const uint8_t* d2 = A.d2 + (size_t)p * A.cap * 32;
    float* pm = A.prevMatched + (size_t)p * A.cap * 2;
    int* m12 = A.m12 + (size_t)p * A.cap;
    int* matchedDist = A.scratch + (size_t)p * A.cap * 5;
    int* m21 = matchedDist + A.cap;
    int* cand = m21 + A.cap;       // compact list of F2 level-0, in-grid keypoints
    int* ckey = cand + A.cap;      // their GetFeaturesInArea order key
    int* binOf = ckey + A.cap;     // rotation bin of i1 (or -1)
    __shared__ int hist[HISTO_LENGTH];

    const float invW = __fdiv_rn((float)GRID_COLS, __fsub_rn(A.maxX, A.minX));
    const float invH = __fdiv_rn((float)GRID_ROWS, __fsub_rn(A.maxY, A.minY));
    for (int i = lane; i < n1; i += 64) { m12[i] = -1; binOf[i] = -1; }
    for (int i = lane; i < n2; i += 64) { matchedDist[i] = 0x7FFFFFFF; m21[i] = -1; }
    if (lane < HISTO_LENGTH) hist[lane] = 0;
    // candidates of level 0 that sit in the 64x48 grid (PosInGrid, src/Frame.cc:462-472)
    int nc = 0;
    for (int j0 = 0; j0 < n2; j0 += 64) {
        int j = j0 + lane;
        bool ok = false; int key = 0;
        if (j < n2) {
            const sslam_keypoint k = kp2[j];
            int px = (int)roundf(__fmul_rn(__fsub_rn(k.x, A.minX), invW));
            int py = (int)roundf(__fmul_rn(__fsub_rn(k.y, A.minY), invH));
            ok = k.octave == 0 && px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS;     // level1 == 0 -> minLevel=maxLevel=0
            key = ((px * GRID_ROWS + py) << 19) | j;
        }
        unsigned long long m = __ballot(ok);
        if (ok) { int o = nc + mbcnt(m); cand[o] = j; ckey[o] = key; }
        nc += __popcll(m);
    }
    __syncthreads();
    int nmatches = 0;
    const float r = (float)A.window;
    for (int i1 = 0; i1 < n1; ++i1) {
        const int level1 = kp1[i1].octave;
        if (level1 > 0) continue;
        const float cx = pm[i1 * 2], cy = pm[i1 * 2 + 1];
        const uint4 q0 = ((const uint4*)(d1 + (size_t)i1 * 32))[0], q1 = ((const uint4*)(d1 + (size_t)i1 * 32))[1];
        unsigned long long b = ~0ull; unsigned s2 = 0x7FFFFFFFu;    // best key (dist<<32|orderkey), second-best distance
        for (int c = lane; c < nc; c += 64) {
            const int j = cand[c];
            const float dx = __fsub_rn(kp2[j].x, cx), dy = __fsub_rn(kp2[j].y, cy);
            if (!(fabsf(dx) < r && fabsf(dy) < r)) continue;
            const uint4* tp = (const uint4*)(d2 + (size_t)j * 32);
            const int dist = hamming256(q0, q1, tp[0], tp[1]);
            if (matchedDist[j] <= dist) continue;
            unsigned long long k = ((unsigned long long)dist << 32) | (unsigned)ckey[c];
            if (k < b) { if (b != ~0ull) s2 = min(s2, (unsigned)(b >> 32)); b = k; }
            else s2 = min(s2, (unsigned)dist);
        }
        const unsigned long long best = wave_min_u64(b);
        if (best == ~0ull) continue;                   // vIndices2 empty, or every candidate suppressed: bestDist stays INT_MAX
        unsigned other = (b == best) ? s2 : min(s2, (unsigned)(b >> 32));
        if (b == ~0ull) other = 0x7FFFFFFFu;
        const unsigned second = wave_min_u32(other);
        const int bestDist = (int)(best >> 32);
        const int bestIdx2 = (int)(best & 0x7FFFF);
        if (bestDist <= TH_LOW && (float)bestDist < __fmul_rn((float)(int)second, A.nnratio)) {
            const int prev = m21[bestIdx2];
            if (prev >= 0) { if (lane == 0) m12[prev] = -1; nmatches--; }
            if (lane == 0) { m12[i1] = bestIdx2; m21[bestIdx2] = i1; matchedDist[bestIdx2] = bestDist; }
            nmatches++;
            if (A.checkOri) {
                float rot = __fsub_rn(kp1[i1].angle, kp2[bestIdx2].angle);
                if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                int bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
                if (bin == HISTO_LENGTH) bin = 0;
                if (lane == 0) { binOf[i1] = bin; hist[bin]++; }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (A.checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;      // ComputeThreeMaxima
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            const int s = hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
        int removed = 0;
        for (int i0 = 0; i0 < n1; i0 += 64) {
            int i = i0 + lane;
            bool rm = false;
            if (i < n1) {
                int bn = binOf[i];
                rm = bn >= 0 && bn != ind1 && bn != ind2 && bn != ind3 && m12[i] >= 0;
                if (rm) m12[i] = -1;
            }
            removed += __popcll(__ballot(rm));
        }
        nmatches -= removed;
    }
    __syncthreads();
    for (int i = lane; i < n1; i += 64) {
        int m = m12[i];
        if (m >= 0) { pm[i * 2] = kp2[m].x; pm[i * 2 + 1] = kp2[m].y; }
    }
    if (lane == 0) A.nmatches[p] = nmatches;
}


// ---------------------------------------------------------------- projection-window matchers
// ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th)  src/ORBmatcher.cc:45-129   (kind 0, mode 0)
// ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)  src/ORBmatcher.cc:1331-1473 (kind 0, mode 1)
// LSDmatcher::SearchByProjection(Frame&, vector<MapLine*>&, th) / (Frame&, const Frame&, ...) src/LSDmatcher.cpp:185-255,22-141 (kind 1, mode 0)
// One wave per problem.  Queries are consumed sequentially (an accepted match occupies its keypoint for
// the later queries); the candidate scan of each query is wave-parallel, with the reference's candidate
// order (GetFeaturesInArea cell order / GetLinesInArea index order) carried in the reduction key.
struct ProjArgs {
    int kind, mode;
    const void* feats; const uint8_t* desc; int n;
    float minX, maxX, minY, maxY;
    const float* uright; const uint8_t* occIn;
    const sslam_proj_query* q; const uint8_t* qdesc; int nq;
    float nnratio; int thDist, checkOri;
    int* assigned; int* nmatches;
    int* scratch;          // occ[n], key[n], qbin[nq], qidx[nq]
};

__global__ __launch_bounds__(64) void k_search_proj(ProjArgs A) {
    const int lane = threadIdx.x;
    const int n = A.n, nq = A.nq;
    int* occ = A.scratch; int* key = occ + n; int* qbin = key + n; int* qidx = qbin + nq;
    const sslam_keypoint* kps = (const sslam_keypoint*)A.feats;
    const sslam_keyline* kls = (const sslam_keyline*)A.feats;
    __shared__ int hist[HISTO_LENGTH];
    if (lane < HISTO_LENGTH) hist[lane] = 0;
    const float invW = __fdiv_rn((float)GRID_COLS, __fsub_rn(A.maxX, A.minX));
    const float invH = __fdiv_rn((float)GRID_ROWS, __fsub_rn(A.maxY, A.minY));
    for (int i = lane; i < n; i += 64) {
        occ[i] = A.occIn ? (int)A.occIn[i] : 0;
        A.assigned[i] = -1;
        int k = i;                                  // lines: GetLinesInArea scans in index order
        if (A.kind == 0) {
            const int px = (int)roundf(__fmul_rn(__fsub_rn(kps[i].x, A.minX), invW));
            const int py = (int)roundf(__fmul_rn(__fsub_rn(kps[i].y, A.minY), invH));
            k = (px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS) ? (((px * GRID_ROWS + py) << 19) | i) : -1;
        }
        key[i] = k;
    }
    for (int i = lane; i < nq; i += 64) qbin[i] = -1;
    __syncthreads();
    int nmatches = 0;
    for (int iq = 0; iq < nq; ++iq) {
        const sslam_proj_query Q = A.q[iq];
        if (!Q.valid) continue;
        const uint4 q0 = ((const uint4*)(A.qdesc + (size_t)iq * 32))[0], q1 = ((const uint4*)(A.qdesc + (size_t)iq * 32))[1];
        unsigned long long b = ~0ull, s = ~0ull;
        bool any = false;
        for (int i = lane; i < n; i += 64) {
            const int k = key[i];
            if (k < 0) continue;
            int oct;
            if (A.kind == 0) {
                const sslam_keypoint kp = kps[i];
                oct = kp.octave;
                if (Q.min_level > 0 || Q.max_level >= 0) {
                    if (oct < Q.min_level) continue;
                    if (Q.max_level >= 0 && oct > Q.max_level) continue;
                }
                const float dx = __fsub_rn(kp.x, Q.u), dy = __fsub_rn(kp.y, Q.v);
                if (!(fabsf(dx) < Q.radius && fabsf(dy) < Q.radius)) continue;
            } else {
                const sslam_keyline kl = kls[i];
                oct = kl.octave;
                const double mxp = 0.5 * (double)__fadd_rn(Q.u, Q.u2) - (double)kl.pt_x, myp = 0.5 * (double)__fadd_rn(Q.v, Q.v2) - (double)kl.pt_y;
                const float distance = (float)(mxp * mxp + myp * myp);
                if (distance > __fmul_rn(Q.radius, Q.radius)) continue;
                const float slope = __fsub_rn(__fdiv_rn(__fsub_rn(Q.v, Q.v2), __fsub_rn(Q.u, Q.u2)), kl.angle);
                if ((double)slope > (double)Q.radius * 0.01) continue;
                if (Q.min_level > 0 || Q.max_level > 0) {
                    if (oct < Q.min_level) continue;
                    if (Q.max_level >= 0 && oct > Q.max_level) continue;
                }
            }
            any = true;                              // vIndices non-empty
            if (occ[i]) continue;
            if (A.kind == 0 && A.uright) {
                const float ur = A.uright[i];
                if (ur > 0 && fabsf(__fsub_rn(Q.ur, ur)) > Q.radius) continue;
            }
            const uint4* tp = (const uint4*)(A.desc + (size_t)i * 32);
            const unsigned long long kk = ((unsigned long long)hamming256(q0, q1, tp[0], tp[1]) << 32) | (unsigned)k;
            if (kk < b) { s = b; b = kk; } else if (kk < s) s = kk;
        }
        if (!__ballot(any)) continue;
        const unsigned long long best = wave_min_u64(b);
        const unsigned long long second = wave_min_u64(b == best ? s : b);
        int bestDist = 256, bestLevel = -1, bestIdx = -1, bestDist2 = 256, bestLevel2 = -1;
        if (best != ~0ull && (int)(best >> 32) < 256) {
            bestDist = (int)(best >> 32); bestIdx = (int)(best & 0x7FFFF);
            bestLevel = A.kind == 0 ? kps[bestIdx].octave : kls[bestIdx].octave;
        }
        if (A.mode == 0 && second != ~0ull && (int)(second >> 32) < 256) {
            bestDist2 = (int)(second >> 32);
            const int i2 = (int)(second & 0x7FFFF);
            bestLevel2 = A.kind == 0 ? kps[i2].octave : kls[i2].octave;
        }
        if (bestDist <= A.thDist && bestIdx >= 0) {
            if (A.mode == 0 && bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(A.nnratio, (float)bestDist2)) continue;
            ++nmatches;
            if (lane == 0) { A.assigned[bestIdx] = iq; occ[bestIdx] = Q.obs_positive ? 1 : 0; }
            if (A.mode == 1 && A.checkOri) {
                float rot = __fsub_rn(Q.angle, kps[bestIdx].angle);
                if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                int bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
                if (bin == HISTO_LENGTH) bin = 0;
                if (lane == 0) { qbin[iq] = bin; qidx[iq] = bestIdx; hist[bin]++; }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (A.mode == 1 && A.checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            const int c = hist[i];
            if (c > max1) { max3 = max2; max2 = max1; max1 = c; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (c > max2) { max3 = max2; max2 = c; ind3 = ind2; ind2 = i; }
            else if (c > max3) { max3 = c; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
        int removed = 0;
        for (int i0 = 0; i0 < nq; i0 += 64) {
            const int i = i0 + lane;
            bool rm = false;
            if (i < nq) { const int bn = qbin[i]; rm = bn >= 0 && bn != ind1 && bn != ind2 && bn != ind3; if (rm) A.assigned[qidx[i]] = -1; }
            removed += __popcll(__ballot(rm));
        }
        nmatches -= removed;
    }
    if (lane == 0) *A.nmatches = nmatches;
}


// ---------------------------------------------------------------- SearchByBoW(KeyFrame*, Frame&)
// src/ORBmatcher.cc:159-291.  One wave: the shared vocabulary nodes are walked in ascending id, the keyframe features
// of a node sequentially (a matched frame feature is skipped by the later ones), the node's frame features lane-parallel
// with the list position in the reduction key (first strictly smaller distance wins).
struct BowArgs {
    const sslam_keypoint* kpKF; const uint8_t* dKF; const uint8_t* validKF;
    const sslam_keypoint* kpF; const uint8_t* dF; int nF;
    const int* ptrKF; const int* ptrF; int nnodes; const int* idxKF; const int* idxF;
    float nnratio; int checkOri; int* assigned; int* nmatches; int* qbin;   // qbin[nF]: rotation bin recorded for a frame feature
};

// A frame feature belongs to one vocabulary node, so the only order dependence of SearchByBoW -- a frame feature that is
// already matched is skipped (:216-217) -- stays inside a node: one wave per node walks that node's keyframe features in order
// (grid = 1 replays all nodes in order, used when a caller's lists share a feature between nodes).  The rotation histogram
// only needs counts; k_bow_finish builds it, prunes and counts.  assigned / qbin arrive as -1, *nmatches as 0.
__global__ __launch_bounds__(64) void k_search_bow(BowArgs A) {
    const int lane = threadIdx.x;
    for (int nd = blockIdx.x; nd < A.nnodes; nd += gridDim.x) {
        const int f0 = A.ptrF[nd], f1 = A.ptrF[nd + 1];
        for (int a = A.ptrKF[nd]; a < A.ptrKF[nd + 1]; ++a) {
            const int ik = A.idxKF[a];
            if (!A.validKF[ik]) continue;
            const uint4 q0 = ((const uint4*)(A.dKF + (size_t)ik * 32))[0], q1 = ((const uint4*)(A.dKF + (size_t)ik * 32))[1];
            unsigned long long b = ~0ull, s = ~0ull;
            for (int p = f0 + lane; p < f1; p += 64) {
                const int jf = A.idxF[p];
                if (A.assigned[jf] >= 0) continue;
                const uint4* tp = (const uint4*)(A.dF + (size_t)jf * 32);
                const unsigned long long kk = ((unsigned long long)hamming256(q0, q1, tp[0], tp[1]) << 32) | (unsigned)(p - f0);
                if (kk < b) { s = b; b = kk; } else if (kk < s) s = kk;
            }
            const unsigned long long best = wave_min_u64(b);
            const unsigned long long second = wave_min_u64(b == best ? s : b);
            int bestDist1 = 256, bestDist2 = 256, bestIdxF = -1;
            if (best != ~0ull && (int)(best >> 32) < 256) { bestDist1 = (int)(best >> 32); bestIdxF = A.idxF[f0 + (int)(unsigned)best]; }
            if (second != ~0ull && (int)(second >> 32) < 256) bestDist2 = (int)(second >> 32);
            if (bestDist1 <= TH_LOW && (float)bestDist1 < __fmul_rn(A.nnratio, (float)bestDist2)) {
                if (lane == 0) {
                    if (A.checkOri) {
                        float rot = __fsub_rn(A.kpKF[ik].angle, A.kpF[bestIdxF].angle);
                        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                        int bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
                        if (bin == HISTO_LENGTH) bin = 0;
                        A.qbin[bestIdxF] = bin;
                    }
                    A.assigned[bestIdxF] = ik;
                }
                __syncthreads();          // the next keyframe feature of this node must see the assignment
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_bow_finish(int* __restrict__ assigned, const int* __restrict__ qbin, int nF, int checkOri, int* __restrict__ nmatches) {
    __shared__ int hist[HISTO_LENGTH];
    __shared__ int keep[3];
    __shared__ int total;
    const int t = threadIdx.x;
    if (t < HISTO_LENGTH) hist[t] = 0;
    if (t == 0) total = 0;
    __syncthreads();
    if (checkOri) {
        for (int i = t; i < nF; i += 256) if (assigned[i] >= 0) atomicAdd(&hist[qbin[i]], 1);
        __syncthreads();
        if (t == 0) {
            int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
            for (int i = 0; i < HISTO_LENGTH; ++i) {
                const int c = hist[i];
                if (c > max1) { max3 = max2; max2 = max1; max1 = c; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (c > max2) { max3 = max2; max2 = c; ind3 = ind2; ind2 = i; }
                else if (c > max3) { max3 = c; ind3 = i; }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
            keep[0] = ind1; keep[1] = ind2; keep[2] = ind3;
        }
        __syncthreads();
    }
    int cnt = 0;
    for (int i = t; i < nF; i += 256) {
        if (assigned[i] < 0) continue;
        if (checkOri) { const int bn = qbin[i]; if (bn != keep[0] && bn != keep[1] && bn != keep[2]) { assigned[i] = -1; continue; } }
        ++cnt;
    }
    atomicAdd(&total, cnt);
    __syncthreads();
    if (t == 0) *nmatches = total;
}

// ---------------------------------------------------------------- line matching
// One 256-thread workgroup per frame pair: knn-2 of n1 query LBD descriptors against
// n2 train descriptors, Frame::lineDescriptorMAD (medians via LDS bitonic sorts), then
// the MAD-gap or ratio gate, pairs emitted in query order.
constexpr int LM_MAX = 1024;

__device__ void lds_sort_asc(float* a, int P2) {
    for (int k = 2; k <= P2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P2; i += blockDim.x) {
                int ixj = i ^ j;
                if (ixj > i) {
                    float x = a[i], y = a[ixj];
                    bool up = (i & k) == 0;
                    if ((x > y) == up) { a[i] = y; a[ixj] = x; }
                }
            }
            __syncthreads();
        }
}

__global__ __launch_bounds__(256) void k_line_match(const uint8_t* __restrict__ l1, const int* __restrict__ n1p, int n1s,
                                                    const uint8_t* __restrict__ l2, const int* __restrict__ n2p, int n2s, int cap,
                                                    double gateScale, int ratioMode, int* __restrict__ pairs, int* __restrict__ npairs,
                                                    double* __restrict__ madOut) {
    __shared__ int bd[LM_MAX], sd[LM_MAX], bi[LM_MAX];
    __shared__ float srt[LM_MAX];
    __shared__ int wcount[4];
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n1 = n1p ? n1p[p] : n1s, n2 = n2p ? n2p[p] : n2s;
    const uint8_t* q = l1 + (size_t)p * cap * 32;
    const uint8_t* t = l2 + (size_t)p * cap * 32;
    int* out = pairs + (size_t)p * cap * 2;
    if (n1 <= 0 || n2 < 2 || n1 > LM_MAX) {      // degenerate (UB in the reference, src/LSDmatcher.cpp:167): defined as 0 matches
        if (tid == 0) { npairs[p] = 0; if (madOut) { madOut[p * 2] = 0; madOut[p * 2 + 1] = 0; } }
        return;
    }
    for (int i = wv; i < n1; i += 4) {
        unsigned long long b, s;
        wave_knn2(q + (size_t)i * 32, t, n2, b, s);
        if (lane == 0) { bd[i] = (int)(b >> 32); bi[i] = (int)(unsigned)b; sd[i] = (int)(s >> 32); }
    }
    __syncthreads();
    int P2 = 1; while (P2 < n1) P2 <<= 1;
    const float INF = 3.0e38f;
    // NN distance MAD
    for (int i = tid; i < P2; i += 256) srt[i] = i < n1 ? (float)bd[i] : INF;
    __syncthreads();
    lds_sort_asc(srt, P2);
    const double med = srt[n1 / 2];
    __syncthreads();
    for (int i = tid; i < P2; i += 256) srt[i] = i < n1 ? fabsf((float)((double)(float)bd[i] - med)) : INF;
    __syncthreads();
    lds_sort_asc(srt, P2);
    const double nnMad = 1.4826 * (double)srt[n1 / 2];
    __syncthreads();
    // NN12 gap MAD: median of the gaps sorted DESCENDING = ascending element n1-1-n1/2
    for (int i = tid; i < P2; i += 256) srt[i] = i < n1 ? __fsub_rn((float)sd[i], (float)bd[i]) : INF;
    __syncthreads();
    lds_sort_asc(srt, P2);
    const double med12 = srt[n1 - 1 - n1 / 2];
    __syncthreads();
    for (int i = tid; i < P2; i += 256) srt[i] = i < n1 ? fabsf((float)((double)__fsub_rn((float)sd[i], (float)bd[i]) - med12)) : INF;
    __syncthreads();
    lds_sort_asc(srt, P2);
    const double nn12Mad = 1.4826 * (double)srt[n1 / 2];
    const double th = nn12Mad * gateScale;
    const float minRatio = 1.0f / 1.5f;
    __syncthreads();
    // gate + ordered compaction
    int base = 0;
    for (int i0 = 0; i0 < n1; i0 += 256) {
        int i = i0 + tid;
        bool ok = false;
        if (i < n1) {
            if (ratioMode) ok = (double)__fdiv_rn((float)bd[i], (float)sd[i]) < (double)minRatio;
            else ok = (double)__fsub_rn((float)sd[i], (float)bd[i]) > th;
        }
        unsigned long long m = __ballot(ok);
        if (lane == 0) wcount[wv] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wv; ++w) off += wcount[w];
        if (ok) { int o = off + mbcnt(m); if (o < cap) { out[o * 2] = i; out[o * 2 + 1] = bi[i]; } }
        base += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
    }
    if (tid == 0) { npairs[p] = base; if (madOut) { madOut[p * 2] = nnMad; madOut[p * 2 + 1] = nn12Mad; } }
}

}  // namespace

// =============================================================== host side
static hipStream_t pick(sslam_ctx* c, void* s) { return s ? (hipStream_t)s : c->stream; }

// ------------------------------------------------------------------ projection matchers, low-latency form
// Same semantics as k_search_proj, built for the single-frame call.  The frame's features are sorted by their
// GetFeaturesInArea order (grid column, row, index: one bitonic sort in LDS) and staged in LDS in that order (position,
// level, angle, occupancy, descriptors), so a query only scans the sorted range of the grid columns its window touches.
// Sixteen waves evaluate sixteen consecutive queries speculatively against the occupancy at the start of the round, each
// wave also taking its own accept / reject decision; wave 0 then commits them in query order and re-evaluates a query only
// if an earlier query of the same round occupied its best or second-best feature (removing any other candidate cannot
// change best / second, so the check is exact).  Global memory is touched once per round (the next round's queries are
// prefetched); the one-wave kernel above spends ~7 us per query on dependent global loads.
constexpr int PROJ_WAVES = 16;
constexpr int PROJ_MAXN = 2048;            // features that fit: 64 B each in sorted order + the sort keys
struct ProjLds { float* px; float* py; float* ang; int* oct; float* ur; int* occ; unsigned* ord; uint4* desc; int* colStart; };
struct ProjDecision { int acc, bestP, secondP, bin; };

// best / second-best candidate of one query (keys: dist | sorted position | level; the position is unique, so the level bits
// below it never take part in a comparison)
__device__ __forceinline__ void proj_scan(const ProjArgs& A, const ProjLds& S, const sslam_proj_query& Q, const uint4& q0, const uint4& q1, int lane,
                                          float invW, unsigned long long& best, unsigned long long& second, bool& anyOut) {
    unsigned long long b = ~0ull, s = ~0ull;
    bool any = false;
    int p0 = 0, p1 = A.n;
    if (A.kind == 0) {        // grid columns the window can touch (KeyFrame/Frame::GetFeaturesInArea's own cell range)
        const int c0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.u, A.minX), Q.radius), invW)));
        const int c1 = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.u, A.minX), Q.radius), invW)));
        if (c0 >= GRID_COLS || c1 < 0) { p0 = p1 = 0; }
        else { p0 = S.colStart[c0]; p1 = S.colStart[c1 + 1]; }
    }
    for (int p = p0 + lane; p < p1; p += 64) {
        const int oct = S.oct[p];
        if (A.kind == 0) {
            if (Q.min_level > 0 || Q.max_level >= 0) {
                if (oct < Q.min_level) continue;
                if (Q.max_level >= 0 && oct > Q.max_level) continue;
            }
            const float dx = __fsub_rn(S.px[p], Q.u), dy = __fsub_rn(S.py[p], Q.v);
            if (!(fabsf(dx) < Q.radius && fabsf(dy) < Q.radius)) continue;
        } else {
            const double mxp = 0.5 * (double)__fadd_rn(Q.u, Q.u2) - (double)S.px[p], myp = 0.5 * (double)__fadd_rn(Q.v, Q.v2) - (double)S.py[p];
            const float distance = (float)(mxp * mxp + myp * myp);
            if (distance > __fmul_rn(Q.radius, Q.radius)) continue;
            const float slope = __fsub_rn(__fdiv_rn(__fsub_rn(Q.v, Q.v2), __fsub_rn(Q.u, Q.u2)), S.ang[p]);
            if ((double)slope > (double)Q.radius * 0.01) continue;
            if (Q.min_level > 0 || Q.max_level > 0) {
                if (oct < Q.min_level) continue;
                if (Q.max_level >= 0 && oct > Q.max_level) continue;
            }
        }
        any = true;                              // vIndices non-empty
        if (S.occ[p]) continue;
        if (A.kind == 0 && A.uright) {
            const float ur = S.ur[p];
            if (ur > 0 && fabsf(__fsub_rn(Q.ur, ur)) > Q.radius) continue;
        }
        const unsigned long long kk = ((unsigned long long)hamming256(q0, q1, S.desc[2 * p], S.desc[2 * p + 1]) << 35) | ((unsigned long long)(unsigned)p << 4) | (unsigned)(oct & 15);
        if (kk < b) { s = b; b = kk; } else if (kk < s) s = kk;
    }
    anyOut = __ballot(any) != 0;
    best = wave_min_u64(b);
    second = wave_min_u64(b == best ? s : b);
}

// the reference's accept / reject logic on (best, second): thresholds, same-level ratio test (mode 0), rotation bin (mode 1)
__device__ __forceinline__ ProjDecision proj_decide(const ProjArgs& A, const ProjLds& S, float qAngle, unsigned long long b, unsigned long long s2) {
    ProjDecision D; D.acc = 0; D.bestP = -1; D.secondP = -2; D.bin = -1;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1;
    if (b != ~0ull) D.bestP = (int)((b >> 4) & 0x7FFFFFFF);
    if (s2 != ~0ull) D.secondP = (int)((s2 >> 4) & 0x7FFFFFFF);
    if (b != ~0ull && (int)(b >> 35) < 256) { bestDist = (int)(b >> 35); bestLevel = (int)(b & 15); }
    if (A.mode == 0 && s2 != ~0ull && (int)(s2 >> 35) < 256) { bestDist2 = (int)(s2 >> 35); bestLevel2 = (int)(s2 & 15); }
    if (bestDist <= A.thDist && b != ~0ull && (int)(b >> 35) < 256) {
        if (A.mode == 0 && bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(A.nnratio, (float)bestDist2)) return D;
        D.acc = 1;
        if (A.mode == 1 && A.checkOri) {
            float rot = __fsub_rn(qAngle, S.ang[D.bestP]);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
            if (bin == HISTO_LENGTH) bin = 0;
            D.bin = bin;
        }
    }
    return D;
}

__global__ __launch_bounds__(PROJ_WAVES * 64) void k_search_proj_lds(ProjArgs A) {
    extern __shared__ __align__(16) uint8_t dyn[];
    constexpr int NT = PROJ_WAVES * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = A.n, nq = A.nq;
    int* key2 = A.scratch + n; int* qbin = key2 + n; int* qidx = qbin + nq;      // same scratch layout as k_search_proj
    int N2 = 64; while (N2 < n) N2 <<= 1;
    ProjLds S;
    S.desc = (uint4*)dyn;
    S.px = (float*)(S.desc + 2 * (size_t)n); S.py = S.px + n; S.ang = S.py + n; S.ur = S.ang + n;
    S.oct = (int*)(S.ur + n); S.occ = S.oct + n; S.ord = (unsigned*)(S.occ + n); S.colStart = (int*)(S.ord + N2);
    __shared__ int rAcc[PROJ_WAVES], rBestP[PROJ_WAVES], rSecondP[PROJ_WAVES], rBin[PROJ_WAVES], rObs[PROJ_WAVES];
    __shared__ int hist[HISTO_LENGTH];
    __shared__ int sh_nmatches;
    const sslam_keypoint* kps = (const sslam_keypoint*)A.feats;
    const sslam_keyline* kls = (const sslam_keyline*)A.feats;
    if (tid < HISTO_LENGTH) hist[tid] = 0;
    if (tid == 0) sh_nmatches = 0;
    const float invW = __fdiv_rn((float)GRID_COLS, __fsub_rn(A.maxX, A.minX));
    const float invH = __fdiv_rn((float)GRID_ROWS, __fsub_rn(A.maxY, A.minY));
    // sort keys: (grid column * ROWS + grid row) << 19 | index; features outside the grid (and the padding) sort last
    for (int i = tid; i < N2; i += NT) {
        unsigned k = 0xFFFFFFFFu;
        if (i < n) {
            A.assigned[i] = -1;
            if (A.kind == 0) {
                const int gx = (int)roundf(__fmul_rn(__fsub_rn(kps[i].x, A.minX), invW));
                const int gy = (int)roundf(__fmul_rn(__fsub_rn(kps[i].y, A.minY), invH));
                if (gx >= 0 && gx < GRID_COLS && gy >= 0 && gy < GRID_ROWS) k = ((unsigned)(gx * GRID_ROWS + gy) << 19) | (unsigned)i;
            } else k = (unsigned)i;                 // lines: GetLinesInArea scans in index order
        }
        S.ord[i] = k;
    }
    for (int i = tid; i < nq; i += NT) qbin[i] = -1;
    __syncthreads();
    if (A.kind == 0) {
        for (int k = 2; k <= N2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < N2; i += NT) {
                    const int l = i ^ j;
                    if (l > i) {
                        const unsigned a = S.ord[i], b2 = S.ord[l];
                        const bool up = (i & k) == 0;
                        if ((a > b2) == up) { S.ord[i] = b2; S.ord[l] = a; }
                    }
                }
                __syncthreads();
            }
    }
    // stage the features in sorted order; A.n shrinks to the features that are in the grid (the rest can never be a candidate)
    int nIn = n;
    if (A.kind == 0) {
        int c = 0;
        for (int i = tid; i < n; i += NT) c += S.ord[i] != 0xFFFFFFFFu ? 1 : 0;
        c = wave_sum(c);
        if (lane == 0) atomicAdd(&sh_nmatches, c);
        __syncthreads();
        nIn = sh_nmatches;
        __syncthreads();
        if (tid == 0) sh_nmatches = 0;
    }
    for (int p = tid; p < nIn; p += NT) {
        const int i = (int)(S.ord[p] & 0x7FFFFu);
        if (A.kind == 0) { const sslam_keypoint kp = kps[i]; S.px[p] = kp.x; S.py[p] = kp.y; S.ang[p] = kp.angle; S.oct[p] = kp.octave; }
        else { const sslam_keyline kl = kls[i]; S.px[p] = kl.pt_x; S.py[p] = kl.pt_y; S.ang[p] = kl.angle; S.oct[p] = kl.octave; }
        S.occ[p] = A.occIn ? (int)A.occIn[i] : 0;
        S.ur[p] = A.uright ? A.uright[i] : -1.f;
        S.desc[2 * p] = ((const uint4*)A.desc)[2 * i]; S.desc[2 * p + 1] = ((const uint4*)A.desc)[2 * i + 1];
    }
    if (A.kind == 0) {        // colStart[c] = first sorted position whose grid column is >= c
        for (int p = tid; p <= nIn; p += NT) {
            const int colPrev = p == 0 ? -1 : (int)(S.ord[p - 1] >> 19) / GRID_ROWS;
            const int colCur = p == nIn ? GRID_COLS : (int)(S.ord[p] >> 19) / GRID_ROWS;
            for (int c = colPrev + 1; c <= colCur; ++c) S.colStart[c] = p;
        }
    }
    __syncthreads();
    ProjArgs B = A; B.n = nIn;
    // this wave's query of the first round
    sslam_proj_query Q; uint4 q0, q1;
    Q.valid = 0; Q.obs_positive = 0; Q.angle = 0.f;
    q0 = q1 = make_uint4(0, 0, 0, 0);
    if (wave < nq) { Q = A.q[wave]; q0 = ((const uint4*)(A.qdesc + (size_t)wave * 32))[0]; q1 = ((const uint4*)(A.qdesc + (size_t)wave * 32))[1]; }
    for (int base = 0; base < nq; base += PROJ_WAVES) {
        const int iq = base + wave;
        // prefetch the next round's query while this one is evaluated
        sslam_proj_query Qn; uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0;
        Qn.valid = 0; Qn.obs_positive = 0; Qn.angle = 0.f;
        const int iqn = iq + PROJ_WAVES;
        if (iqn < nq) { Qn = A.q[iqn]; n0 = ((const uint4*)(A.qdesc + (size_t)iqn * 32))[0]; n1 = ((const uint4*)(A.qdesc + (size_t)iqn * 32))[1]; }
        ProjDecision D; D.acc = 0; D.bestP = -1; D.secondP = -2; D.bin = -1;
        if (iq < nq && Q.valid) {
            unsigned long long best, second; bool any;
            proj_scan(B, S, Q, q0, q1, lane, invW, best, second, any);
            if (any) D = proj_decide(B, S, Q.angle, best, second);
        }
        if (lane == 0) { rAcc[wave] = D.acc; rBestP[wave] = D.bestP; rSecondP[wave] = D.secondP; rBin[wave] = D.bin; rObs[wave] = Q.obs_positive; }
        __syncthreads();
        if (wave == 0) {
            // lane w holds what wave w decided; the serial loop broadcasts with v_readlane, and lane t remembers the t-th
            // feature occupied during this round
            const int cnt = min(PROJ_WAVES, nq - base);
            const int myAcc = lane < cnt ? rAcc[lane] : 0, myBP = lane < cnt ? rBestP[lane] : -1, mySP = lane < cnt ? rSecondP[lane] : -2;
            const int myBin = lane < cnt ? rBin[lane] : -1, myObs = lane < cnt ? rObs[lane] : 0;
            int myTaken = -3, nTaken = 0, accepted = 0;
            for (int w = 0; w < cnt; ++w) {
                int acc = __builtin_amdgcn_readlane(myAcc, w), bP = __builtin_amdgcn_readlane(myBP, w), bin = __builtin_amdgcn_readlane(myBin, w);
                const int sP = __builtin_amdgcn_readlane(mySP, w);
                const int jq = base + w;
                if (nTaken > 0 && __ballot(lane < nTaken && (myTaken == bP || myTaken == sP))) {      // rare: re-evaluate against the updated occupancy
                    const sslam_proj_query Qw = A.q[jq];
                    const uint4 w0 = ((const uint4*)(A.qdesc + (size_t)jq * 32))[0], w1 = ((const uint4*)(A.qdesc + (size_t)jq * 32))[1];
                    unsigned long long b, s2; bool any2;
                    proj_scan(B, S, Qw, w0, w1, lane, invW, b, s2, any2);
                    ProjDecision R; R.acc = 0; R.bestP = -1; R.bin = -1;
                    if (any2) R = proj_decide(B, S, Qw.angle, b, s2);
                    acc = R.acc; bP = R.bestP; bin = R.bin;
                }
                if (!acc) continue;
                ++accepted;
                const int obsPositive = __builtin_amdgcn_readlane(myObs, w);
                if (lane == 0) {
                    const int fi = (int)(S.ord[bP] & 0x7FFFFu);
                    A.assigned[fi] = jq;
                    if (obsPositive) S.occ[bP] = 1;
                    if (A.mode == 1 && A.checkOri) { qbin[jq] = bin; qidx[jq] = fi; hist[bin]++; }
                }
                if (obsPositive) { if (lane == nTaken) myTaken = bP; ++nTaken; }
            }
            if (lane == 0) sh_nmatches += accepted;
        }
        __syncthreads();
        Q = Qn; q0 = n0; q1 = n1;
    }
    if (wave == 0) {
        int nmatches = sh_nmatches;
        if (A.mode == 1 && A.checkOri) {
            int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
            for (int i = 0; i < HISTO_LENGTH; ++i) {
                const int c = hist[i];
                if (c > max1) { max3 = max2; max2 = max1; max1 = c; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (c > max2) { max3 = max2; max2 = c; ind3 = ind2; ind2 = i; }
                else if (c > max3) { max3 = c; ind3 = i; }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
            int removed = 0;
            for (int i0 = 0; i0 < nq; i0 += 64) {
                const int i = i0 + lane;
                bool rm = false;
                if (i < nq) { const int bn = qbin[i]; rm = bn >= 0 && bn != ind1 && bn != ind2 && bn != ind3; if (rm) A.assigned[qidx[i]] = -1; }
                removed += __popcll(__ballot(rm));
            }
            nmatches -= removed;
        }
        if (lane == 0) *A.nmatches = nmatches;
    }
}

// ------------------------------------------------------------------ Fuse: independent best match per projected point / line
// ORBmatcher::Fuse (src/ORBmatcher.cc:897-948 with the chi-square gates, :1055-1080 without) and LSDmatcher::Fuse
// (src/LSDmatcher.cpp:497-523): every query keeps the candidate with the smallest Hamming distance, the first one in
// KeyFrame::GetFeaturesInArea / GetLinesInArea order on ties.  Queries do not interact, so one wave takes one query and scans
// the keyframe's features lane-parallel; the candidate order travels in the low bits of the min-reduction key.
struct FuseArgs {
    int kind, chi2;
    const void* feats; const uint8_t* desc; int n;
    float minX, maxX, minY, maxY;
    const float* uright; const float* invSigma2; int nlevels;
    const sslam_proj_query* q; const uint8_t* qdesc; int nq;
    int* bestIdx; int* bestDist;
};
__global__ __launch_bounds__(64) void k_fuse_search(FuseArgs A) {
    const int lane = threadIdx.x;
    const sslam_keypoint* kps = (const sslam_keypoint*)A.feats;
    const sslam_keyline* kls = (const sslam_keyline*)A.feats;
    const float invW = __fdiv_rn((float)GRID_COLS, __fsub_rn(A.maxX, A.minX));
    const float invH = __fdiv_rn((float)GRID_ROWS, __fsub_rn(A.maxY, A.minY));
    for (int iq = blockIdx.x; iq < A.nq; iq += gridDim.x) {
        const sslam_proj_query Q = A.q[iq];
        unsigned long long b = ~0ull;
        if (Q.valid) {
            const uint4 q0 = ((const uint4*)(A.qdesc + (size_t)iq * 32))[0], q1 = ((const uint4*)(A.qdesc + (size_t)iq * 32))[1];
            for (int i = lane; i < A.n; i += 64) {
                int key = i, lvl;
                if (A.kind == 0) {
                    const sslam_keypoint kp = kps[i];
                    const int px = (int)roundf(__fmul_rn(__fsub_rn(kp.x, A.minX), invW));
                    const int py = (int)roundf(__fmul_rn(__fsub_rn(kp.y, A.minY), invH));
                    if (!(px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS)) continue;      // not in the grid at all
                    key = ((px * GRID_ROWS + py) << 19) | i;
                    const float dx = __fsub_rn(kp.x, Q.u), dy = __fsub_rn(kp.y, Q.v);
                    if (!(fabsf(dx) < Q.radius && fabsf(dy) < Q.radius)) continue;
                    lvl = kp.octave;
                    if (lvl < Q.min_level || lvl > Q.max_level) continue;
                    if (A.chi2) {
                        const float ex = __fsub_rn(Q.u, kp.x), ey = __fsub_rn(Q.v, kp.y);
                        const float inv = (lvl >= 0 && lvl < A.nlevels) ? A.invSigma2[lvl] : 0.f;
                        const float ur = A.uright ? A.uright[i] : -1.f;
                        if (ur >= 0) {
                            const float er = __fsub_rn(Q.ur, ur);
                            const float e2 = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(er, er));
                            if ((double)__fmul_rn(e2, inv) > 7.8) continue;
                        } else {
                            const float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                            if ((double)__fmul_rn(e2, inv) > 5.99) continue;
                        }
                    }
                } else {
                    const sslam_keyline kl = kls[i];
                    const double mxp = 0.5 * (double)__fadd_rn(Q.u, Q.u2) - (double)kl.pt_x, myp = 0.5 * (double)__fadd_rn(Q.v, Q.v2) - (double)kl.pt_y;
                    const float distance = (float)(mxp * mxp + myp * myp);
                    if (distance > __fmul_rn(Q.radius, Q.radius)) continue;
                    const float slope = __fsub_rn(__fdiv_rn(__fsub_rn(Q.v, Q.v2), __fsub_rn(Q.u, Q.u2)), kl.angle);
                    if ((double)slope > (double)Q.radius * 0.01) continue;
                    lvl = kl.octave;
                    if (lvl < Q.min_level || lvl > Q.max_level) continue;
                }
                const uint4* tp = (const uint4*)(A.desc + (size_t)i * 32);
                const unsigned long long kk = ((unsigned long long)hamming256(q0, q1, tp[0], tp[1]) << 32) | (unsigned)key;
                b = kk < b ? kk : b;
            }
        }
        b = wave_min_u64(b);
        if (lane == 0) {
            A.bestIdx[iq] = b == ~0ull ? -1 : (int)(b & 0x7FFFF);
            A.bestDist[iq] = b == ~0ull ? 0x7fffffff : (int)(b >> 32);
        }
    }
}

// ------------------------------------------------------------------ ORBmatcher::SearchForTriangulation
// src/ORBmatcher.cc:660-826 (+ CheckDistEpipolarLine :140-157).  The reference never sets vbMatched2, so every keyframe-1
// feature is an independent query over the keyframe-2 features of its vocabulary node: one wave per query, candidates
// lane-parallel.  `dist > bestDist` (not >=) lets a later candidate with an equal distance win, hence the inverted position
// in the min-reduction key.  k_tri_finish applies the rotation-histogram pruning and counts.
struct TriArgs {
    const sslam_keypoint* kp1; const uint8_t* d1; const float* ur1; const uint8_t* free1; int n1;
    const sslam_keypoint* kp2; const uint8_t* d2; const float* ur2; const uint8_t* free2;
    const int* ptr1; const int* ptr2; int nnodes; const int* idx1; const int* idx2; const int* nodeOf; int total1;
    float F[9]; float ex, ey; const float* scale2; const float* sigma2_2; int nlevels;
    int onlyStereo, checkOri;
    int* m12; int* qbin; int* nmatches;
};
__global__ __launch_bounds__(64) void k_tri_search(TriArgs A) {
    const int lane = threadIdx.x;
    for (int a = blockIdx.x; a < A.total1; a += gridDim.x) {
        const int i1 = A.idx1[a];
        if (!A.free1[i1]) continue;
        const bool st1 = A.ur1 && A.ur1[i1] >= 0;
        if (A.onlyStereo && !st1) continue;
        const int nd = A.nodeOf[a];
        const int f0 = A.ptr2[nd], f1 = A.ptr2[nd + 1];
        const sslam_keypoint k1 = A.kp1[i1];
        const uint4 q0 = ((const uint4*)(A.d1 + (size_t)i1 * 32))[0], q1 = ((const uint4*)(A.d1 + (size_t)i1 * 32))[1];
        const float la = __fadd_rn(__fadd_rn(__fmul_rn(k1.x, A.F[0]), __fmul_rn(k1.y, A.F[3])), A.F[6]);
        const float lb = __fadd_rn(__fadd_rn(__fmul_rn(k1.x, A.F[1]), __fmul_rn(k1.y, A.F[4])), A.F[7]);
        const float lc = __fadd_rn(__fadd_rn(__fmul_rn(k1.x, A.F[2]), __fmul_rn(k1.y, A.F[5])), A.F[8]);
        const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
        unsigned long long b = ~0ull;
        for (int p = f0 + lane; p < f1; p += 64) {
            const int i2 = A.idx2[p];
            if (!A.free2[i2]) continue;
            const bool st2 = A.ur2 && A.ur2[i2] >= 0;
            if (A.onlyStereo && !st2) continue;
            const uint4* tp = (const uint4*)(A.d2 + (size_t)i2 * 32);
            const int dist = hamming256(q0, q1, tp[0], tp[1]);
            if (dist > TH_LOW) continue;
            const sslam_keypoint k2 = A.kp2[i2];
            const int oct = min(max(k2.octave, 0), A.nlevels - 1);
            if (!st1 && !st2) {
                const float dx = __fsub_rn(A.ex, k2.x), dy = __fsub_rn(A.ey, k2.y);
                if (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) < __fmul_rn(100.f, A.scale2[oct])) continue;
            }
            const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, k2.x), __fmul_rn(lb, k2.y)), lc);
            if (den == 0.f) continue;
            const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
            if (!((double)dsqr < 3.84 * (double)A.sigma2_2[oct])) continue;
            const unsigned long long kk = ((unsigned long long)dist << 32) | (unsigned)(0x7FFFFFFF - (p - f0));
            b = kk < b ? kk : b;
        }
        b = wave_min_u64(b);
        if (b != ~0ull && lane == 0) {
            const int i2 = A.idx2[f0 + (0x7FFFFFFF - (int)(unsigned)b)];
            A.m12[i1] = i2;
            if (A.checkOri) {
                float rot = __fsub_rn(k1.angle, A.kp2[i2].angle);
                if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                int bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
                if (bin == HISTO_LENGTH) bin = 0;
                A.qbin[i1] = bin;
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_tri_finish(int* __restrict__ m12, const int* __restrict__ qbin, int n1, int checkOri, int* __restrict__ nmatches) {
    __shared__ int hist[HISTO_LENGTH];
    __shared__ int keep[3];
    __shared__ int total;
    const int t = threadIdx.x;
    if (t < HISTO_LENGTH) hist[t] = 0;
    if (t == 0) total = 0;
    __syncthreads();
    if (checkOri) {
        for (int i = t; i < n1; i += 256) if (m12[i] >= 0) atomicAdd(&hist[qbin[i]], 1);
        __syncthreads();
        if (t == 0) {
            int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
            for (int i = 0; i < HISTO_LENGTH; ++i) {
                const int c = hist[i];
                if (c > max1) { max3 = max2; max2 = max1; max1 = c; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (c > max2) { max3 = max2; max2 = c; ind3 = ind2; ind2 = i; }
                else if (c > max3) { max3 = c; ind3 = i; }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
            keep[0] = ind1; keep[1] = ind2; keep[2] = ind3;
        }
        __syncthreads();
    }
    int cnt = 0;
    for (int i = t; i < n1; i += 256) {
        if (m12[i] < 0) continue;
        if (checkOri) { const int bn = qbin[i]; if (bn != keep[0] && bn != keep[1] && bn != keep[2]) { m12[i] = -1; continue; } }
        ++cnt;
    }
    atomicAdd(&total, cnt);
    __syncthreads();
    if (t == 0) *nmatches = total;
}

// ------------------------------------------------------------------ DBoW2 vocabulary descent (Frame::ComputeBoW)
// TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup), Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1216-1259:
// from the root, move to the child with the smallest Hamming distance (the FIRST such child: `d < best_d`) until a leaf;
// remember the node passed at level L - levelsup.  Features are independent: one lane per feature.
__global__ __launch_bounds__(256) void k_bow_transform(const uint8_t* __restrict__ feat, int n, const int* __restrict__ childPtr, const int* __restrict__ children,
                                                       const uint8_t* __restrict__ nodeDesc, const int* __restrict__ wordId, const double* __restrict__ weight,
                                                       int nidLevel, int* __restrict__ wordOut, double* __restrict__ weightOut, int* __restrict__ nodeOut) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 q0 = ((const uint4*)(feat + (size_t)i * 32))[0], q1 = ((const uint4*)(feat + (size_t)i * 32))[1];
    int node = 0, level = 0, nid = 0;
    while (childPtr[node + 1] > childPtr[node]) {
        ++level;
        const int c0 = childPtr[node], c1 = childPtr[node + 1];
        int best = children[c0];
        const uint4* bp = (const uint4*)(nodeDesc + (size_t)best * 32);
        int bestD = hamming256(q0, q1, bp[0], bp[1]);
        for (int c = c0 + 1; c < c1; ++c) {
            const int id = children[c];
            const uint4* tp = (const uint4*)(nodeDesc + (size_t)id * 32);
            const int d = hamming256(q0, q1, tp[0], tp[1]);
            if (d < bestD) { bestD = d; best = id; }
        }
        node = best;
        if (level == nidLevel) nid = node;
    }
    wordOut[i] = wordId[node]; weightOut[i] = weight[node]; nodeOut[i] = nid;
}

// ------------------------------------------------------------------ distinctive descriptor of an observation set
// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:247-312) == MapLine::ComputeDistinctiveDescriptors
// (src/MapLine.cpp:246-317): all-pairs Hamming distances of the N observed descriptors, per row the median
// `sorted[int(0.5*(N-1))]`, and the FIRST row with the smallest median wins.  One wave per set: the descriptors sit in LDS,
// lane i owns row i (rows i+64, ... in turn); the k-th smallest of a row is found by bisection on the value
// (distances are integers in [0,256]: nine counting passes) instead of sorting.
constexpr int DISTINCT_MAXN = 1024;
__global__ __launch_bounds__(64) void k_distinctive(const uint8_t* __restrict__ desc, const int32_t* __restrict__ ptr, int nsets, int32_t* __restrict__ best) {
    __shared__ __align__(16) unsigned d[DISTINCT_MAXN * 8];
    const int lane = threadIdx.x;
    for (int sIdx = blockIdx.x; sIdx < nsets; sIdx += gridDim.x) {
        const int beg = ptr[sIdx], n = ptr[sIdx + 1] - beg;
        if (n <= 0) { if (lane == 0) best[sIdx] = -1; continue; }
        __syncthreads();
        for (int i = lane; i < n * 8; i += 64) d[i] = ((const unsigned*)(desc + (size_t)beg * 32))[i];
        __syncthreads();
        const int k = (int)(0.5 * (double)(n - 1));              // index of the median in the sorted row
        unsigned long long bestKey = ~0ull;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            int lo = 0, hi = 256;                                  // smallest v with #{j : dist(i,j) <= v} >= k+1
            if (i < n) {
                unsigned a[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) a[w] = d[i * 8 + w];
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    int cnt = 0;
                    for (int j = 0; j < n; ++j) {
                        int dist = 0;
#pragma unroll
                        for (int w = 0; w < 8; ++w) dist += __popc(a[w] ^ d[j * 8 + w]);
                        cnt += dist <= mid ? 1 : 0;
                    }
                    if (cnt >= k + 1) hi = mid; else lo = mid + 1;
                }
                const unsigned long long key = ((unsigned long long)(unsigned)lo << 32) | (unsigned)i;
                bestKey = key < bestKey ? key : bestKey;
            }
        }
        bestKey = wave_min_u64(bestKey);
        if (lane == 0) best[sIdx] = (int)(unsigned)bestKey;
    }
}

extern "C" int sslam_hamming_knn2_dev(sslam_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_idx, int32_t* d_dist, void* stream) {
    if (!ctx || nq < 0 || nt < 0 || (nq > 0 && (!d_q || !d_idx || !d_dist))) { set_error("sslam_hamming_knn2_dev: invalid arguments"); return SSLAM_ERR_INVALID; }
    if (nq == 0) return SSLAM_OK;
    SSLAM_HIP(hipSetDevice(ctx->device));
    { sslam::ProfScope _ps(ctx, "k_knn2", pick(ctx, stream)); hipLaunchKernelGGL(k_knn2, dim3((nq + 3) / 4), dim3(256), 0, pick(ctx, stream), d_q, nq, d_t, nt, d_idx, d_dist); }
    SSLAM_HIP(hipGetLastError());
    return SSLAM_OK;
}

extern "C" int sslam_hamming_knn2_batch_dev(sslam_ctx* ctx, const uint8_t* d_q, const int32_t* d_nq, const uint8_t* d_t, const int32_t* d_nt,
                                            int cap, int nframes, int32_t* d_idx, int32_t* d_dist, void* stream) {
    if (!ctx || !d_q || !d_nq || !d_t || !d_nt || !d_idx || !d_dist || cap <= 0 || nframes <= 0) { set_error("sslam_hamming_knn2_batch_dev: invalid arguments"); return SSLAM_ERR_INVALID; }
    SSLAM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    { sslam::ProfScope _ps(ctx, "k_knn2_batch", st); hipLaunchKernelGGL(k_knn2_batch, dim3((cap + 15) / 16, nframes), dim3(256), 0, st, d_q, d_nq, d_t, d_nt, cap, d_idx, d_dist); }
    SSLAM_HIP(hipGetLastError());
    return SSLAM_OK;
}

extern "C" int sslam_hamming_knn2(sslam_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist) {
    if (!ctx || nq < 0 || nt < 0 || (nq > 0 && (!q || !idx || !dist)) || (nt > 0 && !t)) { set_error("sslam_hamming_knn2: invalid arguments"); return SSLAM_ERR_INVALID; }
    if (nq == 0) return SSLAM_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = ctx->scratch[0].ensure((size_t)nq * 32))) return rc;
    if ((rc = ctx->scratch[1].ensure((size_t)std::max(nt, 1) * 32))) return rc;
    if ((rc = ctx->scratch[2].ensure((size_t)nq * 16))) return rc;
    hipStream_t st = ctx->stream;
    SSLAM_HIP(hipMemcpyAsync(ctx->scratch[0].p, q, (size_t)nq * 32, hipMemcpyHostToDevice, st));
    if (nt) SSLAM_HIP(hipMemcpyAsync(ctx->scratch[1].p, t, (size_t)nt * 32, hipMemcpyHostToDevice, st));
    int* di = ctx->scratch[2].as<int>();
    if ((rc = sslam_hamming_knn2_dev(ctx, ctx->scratch[0].as<uint8_t>(), nq, ctx->scratch[1].as<uint8_t>(), nt, di, di + (size_t)nq * 2, st))) return rc;
    SSLAM_HIP(hipMemcpyAsync(idx, di, (size_t)nq * 8, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(dist, di + (size_t)nq * 2, (size_t)nq * 8, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    return SSLAM_OK;
}

extern "C" int sslam_hamming_matrix(sslam_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* D) {
    if (!ctx || nq < 0 || nt < 0 || ((nq > 0 && nt > 0) && (!q || !t || !D))) { set_error("sslam_hamming_matrix: invalid arguments"); return SSLAM_ERR_INVALID; }
    if (nq == 0 || nt == 0) return SSLAM_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = ctx->scratch[0].ensure((size_t)nq * 32))) return rc;
    if ((rc = ctx->scratch[1].ensure((size_t)nt * 32))) return rc;
    if ((rc = ctx->scratch[2].ensure((size_t)nq * nt * 2))) return rc;
    hipStream_t st = ctx->stream;
    SSLAM_HIP(hipMemcpyAsync(ctx->scratch[0].p, q, (size_t)nq * 32, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(ctx->scratch[1].p, t, (size_t)nt * 32, hipMemcpyHostToDevice, st));
    { sslam::ProfScope _ps(ctx, "k_hamming_matrix", st); hipLaunchKernelGGL(k_hamming_matrix, dim3((nt + 255) / 256, nq), dim3(256), 0, st, ctx->scratch[0].as<uint8_t>(), nq,
                       ctx->scratch[1].as<uint8_t>(), nt, ctx->scratch[2].as<unsigned short>()); }
    SSLAM_HIP(hipGetLastError());
    SSLAM_HIP(hipMemcpyAsync(D, ctx->scratch[2].p, (size_t)nq * nt * 2, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    return SSLAM_OK;
}

extern "C" int sslam_orb_search_for_initialization_batch_dev(sslam_ctx* ctx,
        const sslam_keypoint* d_kp1, const uint8_t* d_desc1, const int32_t* d_n1,
        const sslam_keypoint* d_kp2, const uint8_t* d_desc2, const int32_t* d_n2,
        int cap, int npairs, float* d_prev, int32_t* d_m12, int32_t* d_nm,
        int window, float nnratio, int checkOri, const float bounds[4], void* stream) {
    if (!ctx || !d_kp1 || !d_desc1 || !d_kp2 || !d_desc2 || !d_n1 || !d_n2 || !d_prev || !d_m12 || !d_nm || cap <= 0 || cap >= (1 << 19) || npairs <= 0 || !bounds) {
        set_error("sslam_orb_search_for_initialization_batch_dev: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    SSLAM_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = ctx->scratch[3].ensure(sizeof(int) * 5 * (size_t)cap * npairs))) return rc;
    SfiArgs A;
    A.kp1 = d_kp1; A.d1 = d_desc1; A.n1 = d_n1; A.kp2 = d_kp2; A.d2 = d_desc2; A.n2 = d_n2;
    A.cap = cap; A.n1s = 0; A.n2s = 0; A.prevMatched = d_prev; A.m12 = d_m12; A.nmatches = d_nm;
    A.scratch = ctx->scratch[3].as<int>(); A.window = window; A.nnratio = nnratio; A.checkOri = checkOri;
    A.minX = bounds[0]; A.maxX = bounds[1]; A.minY = bounds[2]; A.maxY = bounds[3];
    { sslam::ProfScope _ps(ctx, "k_search_init", pick(ctx, stream)); hipLaunchKernelGGL(k_search_init, dim3(npairs), dim3(64), 0, pick(ctx, stream), A); }
    SSLAM_HIP(hipGetLastError());
    return SSLAM_OK;
}

extern "C" int sslam_orb_search_for_initialization(sslam_ctx* ctx,
        const sslam_keypoint* kp1, const uint8_t* desc1, int n1, const sslam_keypoint* kp2, const uint8_t* desc2, int n2,
        float* prev_matched, int32_t* matches12, int window, float nnratio, int checkOri, const float bounds[4], int* nmatches_out) {
    if (!ctx || n1 < 0 || n2 < 0 || !nmatches_out || !bounds || (n1 > 0 && (!kp1 || !desc1 || !prev_matched || !matches12)) || (n2 > 0 && (!kp2 || !desc2))) {
        set_error("sslam_orb_search_for_initialization: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    *nmatches_out = 0;
    if (n1 == 0) return SSLAM_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    const int cap = std::max(std::max(n1, n2), 1);
    if (cap >= (1 << 19)) { set_error("too many keypoints"); return SSLAM_ERR_UNSUPPORTED; }
    hipStream_t st = ctx->stream;
    const size_t kb = sizeof(sslam_keypoint) * (size_t)cap, db = 32 * (size_t)cap;
    int rc;
    // layout: kp1 | kp2 | d1 | d2 | prev | m12 | n1,n2,nm
    size_t total = 2 * kb + 2 * db + 8 * (size_t)cap + 4 * (size_t)cap + 64;
    if ((rc = ctx->scratch[4].ensure(total))) return rc;
    uint8_t* base = ctx->scratch[4].as<uint8_t>();
    sslam_keypoint* dk1 = (sslam_keypoint*)base; sslam_keypoint* dk2 = (sslam_keypoint*)(base + kb);
    uint8_t* dd1 = base + 2 * kb; uint8_t* dd2 = dd1 + db;
    float* dpm = (float*)(dd2 + db); int* dm12 = (int*)((uint8_t*)dpm + 8 * (size_t)cap); int* dn = dm12 + cap;
    int hn[3] = {n1, n2, 0};
    SSLAM_HIP(hipMemcpyAsync(dk1, kp1, sizeof(sslam_keypoint) * (size_t)n1, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(dd1, desc1, 32 * (size_t)n1, hipMemcpyHostToDevice, st));
    if (n2) {
        SSLAM_HIP(hipMemcpyAsync(dk2, kp2, sizeof(sslam_keypoint) * (size_t)n2, hipMemcpyHostToDevice, st));
        SSLAM_HIP(hipMemcpyAsync(dd2, desc2, 32 * (size_t)n2, hipMemcpyHostToDevice, st));
    }
    SSLAM_HIP(hipMemcpyAsync(dpm, prev_matched, 8 * (size_t)n1, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(dn, hn, sizeof(hn), hipMemcpyHostToDevice, st));
    if ((rc = sslam_orb_search_for_initialization_batch_dev(ctx, dk1, dd1, dn, dk2, dd2, dn + 1, cap, 1, dpm, dm12, dn + 2, window, nnratio, checkOri, bounds, st))) return rc;
    SSLAM_HIP(hipMemcpyAsync(prev_matched, dpm, 8 * (size_t)n1, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(matches12, dm12, 4 * (size_t)n1, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(hn, dn, sizeof(hn), hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    *nmatches_out = hn[2];
    return SSLAM_OK;
}

extern "C" int sslam_line_match_batch_dev(sslam_ctx* ctx, const uint8_t* d_l1, const int32_t* d_n1, const uint8_t* d_l2, const int32_t* d_n2,
                                          int cap, int npf, double gate_scale, int ratio_mode, int32_t* d_pairs, int32_t* d_npairs, void* stream) {
    if (!ctx || !d_l1 || !d_l2 || !d_n1 || !d_n2 || !d_pairs || !d_npairs || cap <= 0 || npf <= 0) { set_error("sslam_line_match_batch_dev: invalid arguments"); return SSLAM_ERR_INVALID; }
    SSLAM_HIP(hipSetDevice(ctx->device));
    { sslam::ProfScope _ps(ctx, "k_line_match", pick(ctx, stream)); hipLaunchKernelGGL(k_line_match, dim3(npf), dim3(256), 0, pick(ctx, stream), d_l1, d_n1, 0, d_l2, d_n2, 0, cap, gate_scale, ratio_mode, d_pairs, d_npairs, (double*)nullptr); }
    SSLAM_HIP(hipGetLastError());
    return SSLAM_OK;
}

extern "C" int sslam_line_match(sslam_ctx* ctx, const uint8_t* l1, int n1, const uint8_t* l2, int n2, double gate_scale, int ratio_mode,
                                int32_t* pairs_out, int cap, int* npairs_out, double* nn_mad_out, double* nn12_mad_out) {
    if (!ctx || n1 < 0 || n2 < 0 || !npairs_out || (n1 > 0 && (!l1 || !pairs_out)) || (n2 > 0 && !l2)) { set_error("sslam_line_match: invalid arguments"); return SSLAM_ERR_INVALID; }
    *npairs_out = 0;
    if (nn_mad_out) *nn_mad_out = 0;
    if (nn12_mad_out) *nn12_mad_out = 0;
    if (n1 == 0 || n2 < 2) return SSLAM_OK;       // degenerate: defined as no matches
    if (n1 > LM_MAX) { set_error("sslam_line_match: n1=%d exceeds %d", n1, LM_MAX); return SSLAM_ERR_UNSUPPORTED; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int c = std::max(n1, n2);
    int rc;
    size_t total = 64 * (size_t)c + 8 * (size_t)c + 64;
    if ((rc = ctx->scratch[5].ensure(total))) return rc;
    uint8_t* base = ctx->scratch[5].as<uint8_t>();
    uint8_t* d1 = base; uint8_t* d2 = base + 32 * (size_t)c;
    int* dp = (int*)(d2 + 32 * (size_t)c); int* dn = dp + 2 * (size_t)c; double* dm = (double*)(dn + 4);
    SSLAM_HIP(hipMemcpyAsync(d1, l1, 32 * (size_t)n1, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(d2, l2, 32 * (size_t)n2, hipMemcpyHostToDevice, st));
    { sslam::ProfScope _ps(ctx, "k_line_match", st); hipLaunchKernelGGL(k_line_match, dim3(1), dim3(256), 0, st, d1, (const int*)nullptr, n1, d2, (const int*)nullptr, n2, c, gate_scale, ratio_mode, dp, dn, dm); }
    SSLAM_HIP(hipGetLastError());
    int np = 0; double mads[2] = {0, 0};
    std::vector<int> hp(2 * (size_t)c);
    SSLAM_HIP(hipMemcpyAsync(&np, dn, sizeof(int), hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(mads, dm, sizeof(mads), hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(hp.data(), dp, 8 * (size_t)c, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    *npairs_out = np;
    if (nn_mad_out) *nn_mad_out = mads[0];
    if (nn12_mad_out) *nn12_mad_out = mads[1];
    if (np > cap) { set_error("sslam_line_match: %d pairs exceed capacity %d", np, cap); return SSLAM_ERR_CAPACITY; }
    memcpy(pairs_out, hp.data(), 8 * (size_t)np);
    return SSLAM_OK;
}

// shared body of the projection matchers: features already on the device (d_feats / d_desc / d_uright), per-call inputs staged here
static int search_proj_core(sslam_ctx* ctx, int kind, int mode, const void* d_feats, const uint8_t* d_desc, int n, const float bounds[4],
                            const float* d_uright, const uint8_t* occupied, const sslam_proj_query* queries, const uint8_t* qdesc, int nq,
                            float nnratio, int th_dist, int check_orientation, int32_t* assigned_out, int* nmatches_out) {
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t oO = 0, oQ = oO + al((size_t)n), oQD = oQ + al(sizeof(sslam_proj_query) * (size_t)nq), oA = oQD + al(32 * (size_t)nq), oN = oA + al(4 * (size_t)n),
           oS = oN + 256, total = oS + al(4 * (2 * (size_t)n + 2 * (size_t)nq));
    int rc;
    if ((rc = ctx->scratch[6].ensure(total))) return rc;
    uint8_t* B = ctx->scratch[6].as<uint8_t>();
    if (occupied) SSLAM_HIP(hipMemcpyAsync(B + oO, occupied, (size_t)n, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oQ, queries, sizeof(sslam_proj_query) * (size_t)nq, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oQD, qdesc, 32 * (size_t)nq, hipMemcpyHostToDevice, st));
    ProjArgs A;
    A.kind = kind; A.mode = mode; A.feats = (const uint8_t*)d_feats; A.desc = d_desc; A.n = n;
    A.minX = bounds[0]; A.maxX = bounds[1]; A.minY = bounds[2]; A.maxY = bounds[3];
    A.uright = d_uright; A.occIn = occupied ? B + oO : nullptr;
    A.q = (const sslam_proj_query*)(B + oQ); A.qdesc = B + oQD; A.nq = nq; A.nnratio = nnratio; A.thDist = th_dist; A.checkOri = check_orientation;
    A.assigned = (int*)(B + oA); A.nmatches = (int*)(B + oN); A.scratch = (int*)(B + oS);
    if (n <= PROJ_MAXN) {       // the frame fits in LDS: sixteen speculative queries per round
        int n2 = 64; while (n2 < n) n2 <<= 1;
        const size_t lds = (size_t)n * (32 + 6 * 4) + (size_t)n2 * 4 + (GRID_COLS + 2) * 4 + 64;
        if (lds > 48 * 1024) SSLAM_HIP(hipFuncSetAttribute((const void*)k_search_proj_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        sslam::ProfScope _ps(ctx, "k_search_proj_lds", st);
        hipLaunchKernelGGL(k_search_proj_lds, dim3(1), dim3(PROJ_WAVES * 64), lds, st, A);
    } else { sslam::ProfScope _ps(ctx, "k_search_proj", st); hipLaunchKernelGGL(k_search_proj, dim3(1), dim3(64), 0, st, A); }
    SSLAM_HIP(hipGetLastError());
    SSLAM_HIP(hipMemcpyAsync(assigned_out, B + oA, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(nmatches_out, B + oN, sizeof(int), hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    return SSLAM_OK;
}

extern "C" int sslam_search_by_projection(sslam_ctx* ctx, int kind, int mode, const void* feats, const uint8_t* desc, int n, const float bounds[4],
                                          const float* uright, const uint8_t* occupied, const sslam_proj_query* queries, const uint8_t* qdesc, int nq,
                                          float nnratio, int th_dist, int check_orientation, int32_t* assigned_out, int* nmatches_out) {
    if (!ctx || (kind != 0 && kind != 1) || (mode != 0 && mode != 1) || (kind == 1 && mode == 1) || n < 0 || nq < 0 || !nmatches_out || !bounds ||
        (n > 0 && (!feats || !desc || !assigned_out)) || (nq > 0 && (!queries || !qdesc)) || n >= (1 << 19)) {
        set_error("sslam_search_by_projection: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    *nmatches_out = 0;
    for (int i = 0; i < n; ++i) assigned_out[i] = -1;
    if (n == 0 || nq == 0) return SSLAM_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t fsz = kind == 0 ? sizeof(sslam_keypoint) : sizeof(sslam_keyline);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t oF = 0, oD = oF + al(fsz * n), oU = oD + al(32 * (size_t)n), total = oU + al(4 * (size_t)n);
    int rc;
    if ((rc = ctx->scratch[7].ensure(total))) return rc;
    uint8_t* B = ctx->scratch[7].as<uint8_t>();
    SSLAM_HIP(hipMemcpyAsync(B + oF, feats, fsz * n, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oD, desc, 32 * (size_t)n, hipMemcpyHostToDevice, st));
    if (uright) SSLAM_HIP(hipMemcpyAsync(B + oU, uright, 4 * (size_t)n, hipMemcpyHostToDevice, st));
    return search_proj_core(ctx, kind, mode, B + oF, B + oD, n, bounds, uright ? (const float*)(B + oU) : nullptr, occupied, queries, qdesc, nq,
                            nnratio, th_dist, check_orientation, assigned_out, nmatches_out);
}

// ---- device-resident frames
extern "C" int sslam_frame_upload(sslam_ctx* ctx, int kind, const void* feats, const uint8_t* desc, int n, const float* uright, const float bounds[4],
                                  sslam_frame** out) {
    if (!ctx || !out || (kind != 0 && kind != 1) || n < 0 || n >= (1 << 19) || !bounds || (n > 0 && (!feats || !desc))) {
        set_error("sslam_frame_upload: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    sslam_frame* f = new sslam_frame();
    f->ctx = ctx; f->kind = kind; f->n = n; f->hasUright = uright != nullptr;
    for (int i = 0; i < 4; ++i) f->bounds[i] = bounds[i];
    const size_t fsz = kind == 0 ? sizeof(sslam_keypoint) : sizeof(sslam_keyline);
    int rc = SSLAM_OK;
    if ((rc = f->feats.ensure(std::max<size_t>(fsz * n, 256))) || (rc = f->desc.ensure(std::max<size_t>(32 * (size_t)n, 256))) ||
        (uright && (rc = f->uright.ensure(std::max<size_t>(4 * (size_t)n, 256))))) { sslam_frame_destroy(f); return rc; }
    if (n > 0) {
        SSLAM_HIP(hipMemcpyAsync(f->feats.p, feats, fsz * n, hipMemcpyHostToDevice, ctx->stream));
        SSLAM_HIP(hipMemcpyAsync(f->desc.p, desc, 32 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        if (uright) SSLAM_HIP(hipMemcpyAsync(f->uright.p, uright, 4 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        SSLAM_HIP(hipStreamSynchronize(ctx->stream));
    }
    *out = f;
    return SSLAM_OK;
}

// adopt device buffers the extractors already hold (sslam_frame_from_orb / sslam_frame_from_lines): device-to-device snapshot
int sslam_frame_from_device(sslam_ctx* ctx, int kind, const void* d_feats, const uint8_t* d_desc, int n, const float bounds[4], sslam_frame** out) {
    SSLAM_HIP(hipSetDevice(ctx->device));
    sslam_frame* f = new sslam_frame();
    f->ctx = ctx; f->kind = kind; f->n = n;
    for (int i = 0; i < 4; ++i) f->bounds[i] = bounds[i];
    const size_t fsz = kind == 0 ? sizeof(sslam_keypoint) : sizeof(sslam_keyline);
    int rc;
    if ((rc = f->feats.ensure(std::max<size_t>(fsz * n, 256))) || (rc = f->desc.ensure(std::max<size_t>(32 * (size_t)n, 256)))) { sslam_frame_destroy(f); return rc; }
    if (n > 0) {
        SSLAM_HIP(hipMemcpyAsync(f->feats.p, d_feats, fsz * n, hipMemcpyDeviceToDevice, ctx->stream));
        SSLAM_HIP(hipMemcpyAsync(f->desc.p, d_desc, 32 * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
        SSLAM_HIP(hipStreamSynchronize(ctx->stream));
    }
    *out = f;
    return SSLAM_OK;
}

extern "C" void sslam_frame_destroy(sslam_frame* f) {
    if (!f) return;
    if (f->ctx) (void)hipSetDevice(f->ctx->device);
    f->feats.release(); f->desc.release(); f->uright.release();
    delete f;
}

extern "C" int sslam_frame_count(const sslam_frame* f) { return f ? f->n : 0; }

extern "C" int sslam_search_by_projection_frame(sslam_ctx* ctx, const sslam_frame* frame, int mode, const uint8_t* occupied,
                                                const sslam_proj_query* queries, const uint8_t* qdesc, int nq,
                                                float nnratio, int th_dist, int check_orientation, int32_t* assigned_out, int* nmatches_out) {
    if (!ctx || !frame || frame->ctx != ctx || (mode != 0 && mode != 1) || (frame->kind == 1 && mode == 1) || nq < 0 || !nmatches_out ||
        (frame->n > 0 && !assigned_out) || (nq > 0 && (!queries || !qdesc))) {
        set_error("sslam_search_by_projection_frame: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    *nmatches_out = 0;
    for (int i = 0; i < frame->n; ++i) assigned_out[i] = -1;
    if (frame->n == 0 || nq == 0) return SSLAM_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    return search_proj_core(ctx, frame->kind, mode, frame->feats.p, frame->desc.as<uint8_t>(), frame->n, frame->bounds,
                            frame->hasUright ? frame->uright.as<float>() : nullptr, occupied, queries, qdesc, nq, nnratio, th_dist, check_orientation,
                            assigned_out, nmatches_out);
}

extern "C" int sslam_hamming_knn2_frames(sslam_ctx* ctx, const sslam_frame* q, const sslam_frame* t, int32_t* idx, int32_t* dist) {
    if (!ctx || !q || !t || q->ctx != ctx || t->ctx != ctx || (q->n > 0 && (!idx || !dist))) { set_error("sslam_hamming_knn2_frames: invalid arguments"); return SSLAM_ERR_INVALID; }
    if (q->n == 0) return SSLAM_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int rc;
    if ((rc = ctx->scratch[6].ensure(16 * (size_t)q->n))) return rc;
    int32_t* dI = ctx->scratch[6].as<int32_t>();
    int32_t* dD = dI + 2 * (size_t)q->n;
    if ((rc = sslam_hamming_knn2_dev(ctx, q->desc.as<uint8_t>(), q->n, t->desc.as<uint8_t>(), t->n, dI, dD, (void*)st))) return rc;
    SSLAM_HIP(hipMemcpyAsync(idx, dI, 8 * (size_t)q->n, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(dist, dD, 8 * (size_t)q->n, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    return SSLAM_OK;
}

extern "C" int sslam_orb_search_by_bow(sslam_ctx* ctx, const sslam_keypoint* kf_kp, const uint8_t* kf_desc, const uint8_t* kf_valid, int nkf,
                                       const sslam_keypoint* f_kp, const uint8_t* f_desc, int nf, const int32_t* node_kf_ptr, const int32_t* node_f_ptr,
                                       int nnodes, const int32_t* kf_idx, const int32_t* f_idx, float nnratio, int check_orientation,
                                       int32_t* assigned_out, int* nmatches_out) {
    if (!ctx || nkf < 0 || nf < 0 || nnodes < 0 || !nmatches_out || (nf > 0 && !assigned_out) ||
        (nnodes > 0 && (!node_kf_ptr || !node_f_ptr || !kf_idx || !f_idx || !kf_kp || !kf_desc || !kf_valid || !f_kp || !f_desc))) {
        set_error("sslam_orb_search_by_bow: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    *nmatches_out = 0;
    for (int i = 0; i < nf; ++i) assigned_out[i] = -1;
    if (nnodes == 0 || nf == 0 || nkf == 0) return SSLAM_OK;
    const int nk = node_kf_ptr[nnodes], nfi = node_f_ptr[nnodes];
    if (node_kf_ptr[0] != 0 || node_f_ptr[0] != 0 || nk < 0 || nfi < 0) { set_error("sslam_orb_search_by_bow: invalid node offsets"); return SSLAM_ERR_INVALID; }
    for (int i = 0; i < nk; ++i) if (kf_idx[i] < 0 || kf_idx[i] >= nkf) { set_error("sslam_orb_search_by_bow: keyframe feature index out of range"); return SSLAM_ERR_INVALID; }
    for (int i = 0; i < nfi; ++i) if (f_idx[i] < 0 || f_idx[i] >= nf) { set_error("sslam_orb_search_by_bow: frame feature index out of range"); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t ks = sizeof(sslam_keypoint);
    size_t o[16]; size_t off = 0; int k = 0;
    auto take = [&](size_t b) { o[k++] = off; off += al(b); };
    take(ks * nkf); take(32 * (size_t)nkf); take((size_t)nkf); take(ks * nf); take(32 * (size_t)nf);
    take(4 * (size_t)(nnodes + 1)); take(4 * (size_t)(nnodes + 1)); take(4 * (size_t)std::max(nk, 1)); take(4 * (size_t)std::max(nfi, 1));
    take(4 * (size_t)nf); take(256); take(4 * (size_t)nf);
    int rc;
    if ((rc = ctx->scratch[7].ensure(off))) return rc;
    uint8_t* B = ctx->scratch[7].as<uint8_t>();
    const void* src[9] = {kf_kp, kf_desc, kf_valid, f_kp, f_desc, node_kf_ptr, node_f_ptr, kf_idx, f_idx};
    const size_t len[9] = {ks * nkf, 32 * (size_t)nkf, (size_t)nkf, ks * nf, 32 * (size_t)nf, 4 * (size_t)(nnodes + 1), 4 * (size_t)(nnodes + 1), 4 * (size_t)nk, 4 * (size_t)nfi};
    for (int i = 0; i < 9; ++i) if (len[i]) SSLAM_HIP(hipMemcpyAsync(B + o[i], src[i], len[i], hipMemcpyHostToDevice, st));
    BowArgs A;
    A.kpKF = (const sslam_keypoint*)(B + o[0]); A.dKF = B + o[1]; A.validKF = B + o[2]; A.kpF = (const sslam_keypoint*)(B + o[3]); A.dF = B + o[4]; A.nF = nf;
    A.ptrKF = (const int*)(B + o[5]); A.ptrF = (const int*)(B + o[6]); A.nnodes = nnodes; A.idxKF = (const int*)(B + o[7]); A.idxF = (const int*)(B + o[8]);
    A.nnratio = nnratio; A.checkOri = check_orientation; A.assigned = (int*)(B + o[9]); A.nmatches = (int*)(B + o[10]); A.qbin = (int*)(B + o[11]);
    SSLAM_HIP(hipMemsetAsync(A.assigned, 0xFF, 4 * (size_t)nf, st));
    SSLAM_HIP(hipMemsetAsync(A.qbin, 0xFF, 4 * (size_t)nf, st));
    bool disjoint = true;                     // DBoW2 puts a feature under exactly one node; if a caller's lists do not, replay in order
    {
        std::vector<uint8_t> seen((size_t)nf, 0);
        for (int i = 0; i < nfi && disjoint; ++i) { if (seen[f_idx[i]]) disjoint = false; seen[f_idx[i]] = 1; }
    }
    { sslam::ProfScope _ps(ctx, "k_search_bow", st); hipLaunchKernelGGL(k_search_bow, dim3(disjoint ? std::min(nnodes, 4096) : 1), dim3(64), 0, st, A); }
    { sslam::ProfScope _ps(ctx, "k_bow_finish", st); hipLaunchKernelGGL(k_bow_finish, dim3(1), dim3(256), 0, st, A.assigned, A.qbin, nf, check_orientation, A.nmatches); }
    SSLAM_HIP(hipGetLastError());
    SSLAM_HIP(hipMemcpyAsync(assigned_out, B + o[9], 4 * (size_t)nf, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(nmatches_out, B + o[10], sizeof(int), hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    return SSLAM_OK;
}


// MapPoint / MapLine ::ComputeDistinctiveDescriptors for `nsets` observation sets at once: set s owns descriptor rows
// ptr[s] .. ptr[s+1] of `desc`; best_out[s] = index (inside the set) of the descriptor with the least median distance to
// the rest, first such row on ties, -1 for an empty set.
extern "C" int sslam_distinctive_descriptors(sslam_ctx* ctx, const uint8_t* desc, const int32_t* ptr, int nsets, int32_t* best_out) {
    if (!ctx || nsets < 0 || (nsets > 0 && (!ptr || !best_out))) { set_error("sslam_distinctive_descriptors: invalid arguments"); return SSLAM_ERR_INVALID; }
    if (nsets == 0) return SSLAM_OK;
    const int total = ptr[nsets];
    if (ptr[0] != 0 || total < 0 || (total > 0 && !desc)) { set_error("sslam_distinctive_descriptors: invalid set offsets"); return SSLAM_ERR_INVALID; }
    for (int s2 = 0; s2 < nsets; ++s2) {
        const int n = ptr[s2 + 1] - ptr[s2];
        if (n < 0) { set_error("sslam_distinctive_descriptors: offsets must be non-decreasing"); return SSLAM_ERR_INVALID; }
        if (n > DISTINCT_MAXN) { set_error("sslam_distinctive_descriptors: a set of %d descriptors exceeds the supported %d", n, DISTINCT_MAXN); return SSLAM_ERR_UNSUPPORTED; }
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t oD = 0, oP = oD + al(32 * (size_t)std::max(total, 1)), oB = oP + al(4 * (size_t)(nsets + 1)), tot = oB + al(4 * (size_t)nsets);
    int rc;
    if ((rc = ctx->scratch[6].ensure(tot))) return rc;
    uint8_t* B = ctx->scratch[6].as<uint8_t>();
    if (total > 0) SSLAM_HIP(hipMemcpyAsync(B + oD, desc, 32 * (size_t)total, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oP, ptr, 4 * (size_t)(nsets + 1), hipMemcpyHostToDevice, st));
    { sslam::ProfScope _ps(ctx, "k_distinctive", st); hipLaunchKernelGGL(k_distinctive, dim3(std::min(nsets, 4096)), dim3(64), 0, st, B + oD, (const int32_t*)(B + oP), nsets, (int32_t*)(B + oB)); }
    SSLAM_HIP(hipGetLastError());
    SSLAM_HIP(hipMemcpyAsync(best_out, B + oB, 4 * (size_t)nsets, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    return SSLAM_OK;
}

// The candidate search of ORBmatcher::Fuse (both overloads) and LSDmatcher::Fuse on a device-resident keyframe: per query the
// feature with the smallest descriptor distance inside the window (first in GetFeaturesInArea / GetLinesInArea order on ties),
// -1 / INT_MAX when the window holds no admissible feature.  The caller applies `bestDist <= TH_LOW` and the Replace /
// AddObservation bookkeeping in query order, exactly as the reference does after its inner loop.
extern "C" int sslam_fuse_search(sslam_ctx* ctx, const sslam_frame* kf, int chi2_mode, const float* inv_level_sigma2, int nlevels,
                                 const sslam_proj_query* queries, const uint8_t* qdesc, int nq, int32_t* best_idx_out, int32_t* best_dist_out) {
    if (!ctx || !kf || kf->ctx != ctx || (chi2_mode != 0 && chi2_mode != 1) || nq < 0 || (nq > 0 && (!queries || !qdesc || !best_idx_out || !best_dist_out)) ||
        (chi2_mode == 1 && (kf->kind != 0 || !inv_level_sigma2 || nlevels <= 0 || nlevels > 64))) {
        set_error("sslam_fuse_search: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    for (int i = 0; i < nq; ++i) { best_idx_out[i] = -1; best_dist_out[i] = 0x7fffffff; }
    if (nq == 0 || kf->n == 0) return SSLAM_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t oQ = 0, oQD = oQ + al(sizeof(sslam_proj_query) * (size_t)nq), oS = oQD + al(32 * (size_t)nq), oI = oS + 256, oD = oI + al(4 * (size_t)nq),
                 total = oD + al(4 * (size_t)nq);
    int rc;
    if ((rc = ctx->scratch[6].ensure(total))) return rc;
    uint8_t* B = ctx->scratch[6].as<uint8_t>();
    SSLAM_HIP(hipMemcpyAsync(B + oQ, queries, sizeof(sslam_proj_query) * (size_t)nq, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oQD, qdesc, 32 * (size_t)nq, hipMemcpyHostToDevice, st));
    if (chi2_mode) SSLAM_HIP(hipMemcpyAsync(B + oS, inv_level_sigma2, sizeof(float) * (size_t)nlevels, hipMemcpyHostToDevice, st));
    FuseArgs A;
    A.kind = kf->kind; A.chi2 = chi2_mode; A.feats = kf->feats.p; A.desc = kf->desc.as<uint8_t>(); A.n = kf->n;
    A.minX = kf->bounds[0]; A.maxX = kf->bounds[1]; A.minY = kf->bounds[2]; A.maxY = kf->bounds[3];
    A.uright = kf->hasUright ? kf->uright.as<float>() : nullptr; A.invSigma2 = (const float*)(B + oS); A.nlevels = nlevels;
    A.q = (const sslam_proj_query*)(B + oQ); A.qdesc = B + oQD; A.nq = nq; A.bestIdx = (int*)(B + oI); A.bestDist = (int*)(B + oD);
    { sslam::ProfScope _ps(ctx, "k_fuse_search", st); hipLaunchKernelGGL(k_fuse_search, dim3(std::min(nq, 8192)), dim3(64), 0, st, A); }
    SSLAM_HIP(hipGetLastError());
    SSLAM_HIP(hipMemcpyAsync(best_idx_out, B + oI, 4 * (size_t)nq, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(best_dist_out, B + oD, 4 * (size_t)nq, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    return SSLAM_OK;
}

// ORBmatcher::SearchForTriangulation on two device-resident keyframes (their mvuRight travel with the handles).
extern "C" int sslam_orb_search_for_triangulation(sslam_ctx* ctx, const sslam_frame* kf1, const sslam_frame* kf2, const uint8_t* free1, const uint8_t* free2,
                                                  const int32_t* node_kf1_ptr, const int32_t* node_kf2_ptr, int nnodes, const int32_t* kf1_idx, const int32_t* kf2_idx,
                                                  const float F12[9], float ex, float ey, const float* scale_factors2, const float* level_sigma2_2, int nlevels,
                                                  int only_stereo, int check_orientation, int32_t* matches12_out, int* nmatches_out) {
    if (!ctx || !kf1 || !kf2 || kf1->ctx != ctx || kf2->ctx != ctx || kf1->kind != 0 || kf2->kind != 0 || nnodes < 0 || !nmatches_out || !F12 ||
        !scale_factors2 || !level_sigma2_2 || nlevels <= 0 || nlevels > 64 || (kf1->n > 0 && (!free1 || !matches12_out)) || (kf2->n > 0 && !free2) ||
        (nnodes > 0 && (!node_kf1_ptr || !node_kf2_ptr || !kf1_idx || !kf2_idx))) {
        set_error("sslam_orb_search_for_triangulation: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    *nmatches_out = 0;
    const int n1 = kf1->n, n2 = kf2->n;
    for (int i = 0; i < n1; ++i) matches12_out[i] = -1;
    if (n1 == 0 || n2 == 0 || nnodes == 0) return SSLAM_OK;
    const int total1 = node_kf1_ptr[nnodes], total2 = node_kf2_ptr[nnodes];
    if (node_kf1_ptr[0] != 0 || node_kf2_ptr[0] != 0 || total1 < 0 || total2 < 0) { set_error("sslam_orb_search_for_triangulation: invalid node offsets"); return SSLAM_ERR_INVALID; }
    for (int i = 0; i < total1; ++i) if (kf1_idx[i] < 0 || kf1_idx[i] >= n1) { set_error("sslam_orb_search_for_triangulation: keyframe-1 index out of range"); return SSLAM_ERR_INVALID; }
    for (int i = 0; i < total2; ++i) if (kf2_idx[i] < 0 || kf2_idx[i] >= n2) { set_error("sslam_orb_search_for_triangulation: keyframe-2 index out of range"); return SSLAM_ERR_INVALID; }
    if (total1 == 0 || total2 == 0) return SSLAM_OK;
    std::vector<int32_t> nodeOf((size_t)total1);
    for (int nd = 0; nd < nnodes; ++nd) for (int a = node_kf1_ptr[nd]; a < node_kf1_ptr[nd + 1]; ++a) nodeOf[a] = nd;
    std::lock_guard<std::mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al(bytes); return o; };
    const size_t oF1 = take(n1), oF2 = take(n2), oP1 = take(4 * (size_t)(nnodes + 1)), oP2 = take(4 * (size_t)(nnodes + 1)), oI1 = take(4 * (size_t)total1),
                 oI2 = take(4 * (size_t)total2), oNO = take(4 * (size_t)total1), oSF = take(4 * (size_t)nlevels), oSG = take(4 * (size_t)nlevels),
                 oM = take(4 * (size_t)n1), oQB = take(4 * (size_t)n1), oN = take(4);
    int rc;
    if ((rc = ctx->scratch[6].ensure(off))) return rc;
    uint8_t* B = ctx->scratch[6].as<uint8_t>();
    SSLAM_HIP(hipMemcpyAsync(B + oF1, free1, n1, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oF2, free2, n2, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oP1, node_kf1_ptr, 4 * (size_t)(nnodes + 1), hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oP2, node_kf2_ptr, 4 * (size_t)(nnodes + 1), hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oI1, kf1_idx, 4 * (size_t)total1, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oI2, kf2_idx, 4 * (size_t)total2, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oNO, nodeOf.data(), 4 * (size_t)total1, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oSF, scale_factors2, 4 * (size_t)nlevels, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oSG, level_sigma2_2, 4 * (size_t)nlevels, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemsetAsync(B + oM, 0xFF, 4 * (size_t)n1, st));
    TriArgs A;
    A.kp1 = kf1->feats.as<sslam_keypoint>(); A.d1 = kf1->desc.as<uint8_t>(); A.ur1 = kf1->hasUright ? kf1->uright.as<float>() : nullptr; A.free1 = B + oF1; A.n1 = n1;
    A.kp2 = kf2->feats.as<sslam_keypoint>(); A.d2 = kf2->desc.as<uint8_t>(); A.ur2 = kf2->hasUright ? kf2->uright.as<float>() : nullptr; A.free2 = B + oF2;
    A.ptr1 = (const int*)(B + oP1); A.ptr2 = (const int*)(B + oP2); A.nnodes = nnodes; A.idx1 = (const int*)(B + oI1); A.idx2 = (const int*)(B + oI2);
    A.nodeOf = (const int*)(B + oNO); A.total1 = total1;
    for (int i = 0; i < 9; ++i) A.F[i] = F12[i];
    A.ex = ex; A.ey = ey; A.scale2 = (const float*)(B + oSF); A.sigma2_2 = (const float*)(B + oSG); A.nlevels = nlevels;
    A.onlyStereo = only_stereo; A.checkOri = check_orientation;
    A.m12 = (int*)(B + oM); A.qbin = (int*)(B + oQB); A.nmatches = (int*)(B + oN);
    { sslam::ProfScope _ps(ctx, "k_tri_search", st); hipLaunchKernelGGL(k_tri_search, dim3(std::min(total1, 8192)), dim3(64), 0, st, A); }
    { sslam::ProfScope _ps(ctx, "k_tri_finish", st); hipLaunchKernelGGL(k_tri_finish, dim3(1), dim3(256), 0, st, A.m12, A.qbin, n1, check_orientation, A.nmatches); }
    SSLAM_HIP(hipGetLastError());
    SSLAM_HIP(hipMemcpyAsync(matches12_out, B + oM, 4 * (size_t)n1, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(nmatches_out, B + oN, sizeof(int), hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    return SSLAM_OK;
}

// ---- DBoW2 vocabulary
extern "C" int sslam_vocab_create(sslam_ctx* ctx, int nnodes, int levels, const int32_t* child_ptr, const int32_t* children, const uint8_t* node_desc,
                                  const int32_t* word_id, const double* weight, sslam_vocab** out) {
    if (!ctx || !out || nnodes < 1 || levels < 1 || !child_ptr || !node_desc || !word_id || !weight) { set_error("sslam_vocab_create: invalid arguments"); return SSLAM_ERR_INVALID; }
    const int nch = child_ptr[nnodes];
    if (child_ptr[0] != 0 || nch < 0 || (nch > 0 && !children)) { set_error("sslam_vocab_create: invalid child offsets"); return SSLAM_ERR_INVALID; }
    for (int i = 0; i < nnodes; ++i) if (child_ptr[i + 1] < child_ptr[i]) { set_error("sslam_vocab_create: child offsets must be non-decreasing"); return SSLAM_ERR_INVALID; }
    for (int i = 0; i < nch; ++i) if (children[i] <= 0 || children[i] >= nnodes) { set_error("sslam_vocab_create: child id out of range"); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    sslam_vocab* v = new sslam_vocab();
    v->ctx = ctx; v->nnodes = nnodes; v->levels = levels;
    int rc;
    if ((rc = v->childPtr.ensure(4 * (size_t)(nnodes + 1))) || (rc = v->children.ensure(std::max<size_t>(4 * (size_t)nch, 256))) || (rc = v->desc.ensure(32 * (size_t)nnodes)) ||
        (rc = v->wordId.ensure(4 * (size_t)nnodes)) || (rc = v->weight.ensure(8 * (size_t)nnodes))) { sslam_vocab_destroy(v); return rc; }
    hipStream_t st = ctx->stream;
    SSLAM_HIP(hipMemcpyAsync(v->childPtr.p, child_ptr, 4 * (size_t)(nnodes + 1), hipMemcpyHostToDevice, st));
    if (nch > 0) SSLAM_HIP(hipMemcpyAsync(v->children.p, children, 4 * (size_t)nch, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(v->desc.p, node_desc, 32 * (size_t)nnodes, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(v->wordId.p, word_id, 4 * (size_t)nnodes, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(v->weight.p, weight, 8 * (size_t)nnodes, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    *out = v;
    return SSLAM_OK;
}

extern "C" void sslam_vocab_destroy(sslam_vocab* v) {
    if (!v) return;
    if (v->ctx) (void)hipSetDevice(v->ctx->device);
    v->childPtr.release(); v->children.release(); v->desc.release(); v->wordId.release(); v->weight.release();
    delete v;
}

static int bow_core(sslam_ctx* ctx, const sslam_vocab* v, const uint8_t* d_desc, int n, int levelsup, int32_t* word_out, double* weight_out, int32_t* node_out) {
    hipStream_t st = ctx->stream;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t oW = 0, oV = oW + al(4 * (size_t)n), oN = oV + al(8 * (size_t)n), total = oN + al(4 * (size_t)n);
    int rc;
    if ((rc = ctx->scratch[6].ensure(total))) return rc;
    uint8_t* B = ctx->scratch[6].as<uint8_t>();
    { sslam::ProfScope _ps(ctx, "k_bow_transform", st);
      hipLaunchKernelGGL(k_bow_transform, dim3((n + 255) / 256), dim3(256), 0, st, d_desc, n, v->childPtr.as<int>(), v->children.as<int>(), v->desc.as<uint8_t>(),
                         v->wordId.as<int>(), v->weight.as<double>(), v->levels - levelsup, (int*)(B + oW), (double*)(B + oV), (int*)(B + oN)); }
    SSLAM_HIP(hipGetLastError());
    SSLAM_HIP(hipMemcpyAsync(word_out, B + oW, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(weight_out, B + oV, 8 * (size_t)n, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(node_out, B + oN, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    return SSLAM_OK;
}

extern "C" int sslam_bow_transform_frame(sslam_ctx* ctx, const sslam_vocab* vocab, const sslam_frame* frame, int levelsup,
                                         int32_t* word_out, double* weight_out, int32_t* node_out) {
    if (!ctx || !vocab || !frame || vocab->ctx != ctx || frame->ctx != ctx || levelsup < 0 || (frame->n > 0 && (!word_out || !weight_out || !node_out))) {
        set_error("sslam_bow_transform_frame: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    if (frame->n == 0) return SSLAM_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    return bow_core(ctx, vocab, frame->desc.as<uint8_t>(), frame->n, levelsup, word_out, weight_out, node_out);
}

extern "C" int sslam_bow_transform(sslam_ctx* ctx, const sslam_vocab* vocab, const uint8_t* desc, int n, int levelsup,
                                   int32_t* word_out, double* weight_out, int32_t* node_out) {
    if (!ctx || !vocab || vocab->ctx != ctx || n < 0 || levelsup < 0 || (n > 0 && (!desc || !word_out || !weight_out || !node_out))) {
        set_error("sslam_bow_transform: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    if (n == 0) return SSLAM_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = ctx->scratch[7].ensure(32 * (size_t)n))) return rc;
    SSLAM_HIP(hipMemcpyAsync(ctx->scratch[7].p, desc, 32 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    return bow_core(ctx, vocab, ctx->scratch[7].as<uint8_t>(), n, levelsup, word_out, weight_out, node_out);
}
