// Matchers whose queries do not interact: Fuse / Sim3 candidate search, SearchForTriangulation, DBoW2 vocabulary descent,
// distinctive-descriptor selection.
// Part of match.hip (included there, inside its anonymous namespace: one translation unit).  Not a standalone header.
#pragma once

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
