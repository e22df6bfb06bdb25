// Brute-force Hamming kernels: constants, 256-bit popcount distance, knn-2 (BFMatcher semantics), distance matrix,
// knn-2 + MAD gates for lines.
// Part of match.hip (included there, inside its anonymous namespace: one translation unit).  Not a standalone header.
#pragma once

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
    if (ntf <= 65536) {
        // 32-bit keys (dist << 16 | train index; all keys of a query are distinct): best and second-best are three integer min / max per
        // pair -- b' = min(b, k), s' = min(s, max(b, k)) -- instead of 64-bit compares and selects
        unsigned b[4] = {~0u, ~0u, ~0u, ~0u}, s[4] = {~0u, ~0u, ~0u, ~0u};
        for (int j = lane; j < ntf; j += 64) {
            const uint4* tp = (const uint4*)(tb + (size_t)j * 32);
            const uint4 t0 = tp[0], t1 = tp[1];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned kk = ((unsigned)hamming256(qa[k], qc[k], t0, t1) << 16) | (unsigned)j;
                s[k] = min(s[k], max(b[k], kk));
                b[k] = min(b[k], kk);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned best = wave_min_u32(b[k]);
            const unsigned second = wave_min_u32(b[k] == best ? s[k] : b[k]);
            if (lane == 0 && q0i + k < nqf) {
                const size_t o = ((size_t)f * cap + q0i + k) * 2;
                idx[o] = best == ~0u ? -1 : (int)(best & 0xFFFFu);
                dist[o] = best == ~0u ? -1 : (int)(best >> 16);
                idx[o + 1] = second == ~0u ? -1 : (int)(second & 0xFFFFu);
                dist[o + 1] = second == ~0u ? -1 : (int)(second >> 16);
            }
        }
        return;
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

// ---------------------------------------------------------------- knn-2 of a batch on the matrix cores
// The dense best / second-best search of a frame pair is 10^6 descriptor pairs; as xor + popcount it is 20 vector instructions per pair
// and lane (k_knn2_batch: 0.43 M wave-instructions per frame, 6 % of a step that is bound by vector issue, docs/history/DESIGN_rounds_1-4.md 5f).  The same
// distances as a product of +-64 matrices (int8): sum_k a_k b_k = 4096 (256 - 2 h), h the Hamming distance -- exact in the i32
// accumulator of v_mfma_i32_32x32x32_i8.  One wave holds 32 (or 64) queries as the B operand (column j = lane & 31; 8 k-steps x 16 bytes
// per lane, expanded once from the 256 bits) and streams the train rows through the A operand in tiles of 32; each lane then owns ONE
// query and 16 train rows per tile (C/D layout: row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)), so best and second-best are a per-lane
// running max / min over keys -- a handful of vector instructions per pair instead of 20, and the multiply-adds run on an otherwise
// idle unit.  The accumulator is started at 4096 * 256 + (31 - row-in-tile) (a constant vector), so that
//     key = acc + 32 * (127 - tile) = 4096 (512 - 2 h) + 32 (127 - tile) + (31 - row-in-tile)              (one add)
// orders by distance first and by train index second exactly like BFMatcher's (dist, index) comparison (A.10); a key is > 0, 0 = none.
// 7 bits of tile: up to 4096 train rows per frame (sslam_hamming_knn2_batch_dev keeps the popcount form beyond).
// The k index a byte of the A / B operands stands for is whatever the hardware assigns to (lane >> 5, byte): both operands are filled
// by the same rule (descriptor bit 32 s + 16 (lane >> 5) + byte in k-step s), which is all a dot product needs.
// k_knn2_expand writes the train side in operand order: frame f, tile T, k-step s = 64 lanes x 16 bytes, contiguous.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int KNN_MFMA_MAX_TILES = 128;
__device__ __forceinline__ unsigned spread4(unsigned n) { return __umul24(n, 0x00204081u) & 0x01010101u; }      // bit i of a nibble -> byte i (0 / 1)
__device__ __forceinline__ int pm64(unsigned nibble) { const unsigned m = spread4(nibble); return (int)((m << 7) ^ 0xC0C0C0C0u); }      // bit 1 -> +64 (0x40), bit 0 -> -64 (0xC0)

__global__ __launch_bounds__(64) void k_knn2_expand(const uint8_t* __restrict__ t, const int* __restrict__ nt, int cap, int tilesCap, uint8_t* __restrict__ out) {
    const int T = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    const int ntf = nt[f];
    if (T * 32 >= ntf) return;                     // tiles past the frame's rows are never read
    const int row = T * 32 + (lane & 31), h = lane >> 5;
    uint4 d0 = make_uint4(0, 0, 0, 0), d1 = d0;
    const bool valid = row < ntf;
    if (valid) { const uint4* tp = (const uint4*)(t + ((size_t)f * cap + row) * 32); d0 = tp[0]; d1 = tp[1]; }
    const unsigned w[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    v4i* o = (v4i*)(out + ((size_t)f * tilesCap + T) * 8 * 1024) + lane;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const unsigned bits = (w[s] >> (16 * h)) & 0xFFFFu;
        v4i v;
#pragma unroll
        for (int d = 0; d < 4; ++d) v[d] = valid ? pm64((bits >> (4 * d)) & 15u) : 0;      // (rows past the end: 0, and masked again below)
        o[s * 64] = v;
    }
}

// QS = query sets of 32 per wave: 2 halves the passes over the frame's train rows (the kernel's L2 traffic)
template <int QS>
__global__ __launch_bounds__(64) void k_knn2_mfma(const uint8_t* __restrict__ q, const int* __restrict__ nq, const uint8_t* __restrict__ texp,
                                                  const int* __restrict__ nt, int cap, int tilesCap, int qblocks, int nframes,
                                                  int* __restrict__ idx, int* __restrict__ dist) {
    // every query block of a frame on the same XCD (workgroups are dealt round-robin over the eight): the frame's expanded train rows
    // (256 KB) are fetched into one L2 instead of eight
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3, qf = kk / qblocks, qb = kk - qf * qblocks, f = qf * 8 + xcd;
    if (f >= nframes) return;
    const int nqf = nq[f], ntf = nt[f], lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int q0 = qb * 32 * QS;
    if (q0 >= nqf) return;
    v4i Bf[QS][8];
#pragma unroll
    for (int u = 0; u < QS; ++u) {
        const uint4* qp = (const uint4*)(q + ((size_t)f * cap + min(q0 + 32 * u + j, nqf - 1)) * 32);
        const uint4 d0 = qp[0], d1 = qp[1];
        const unsigned w[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const unsigned bits = (w[s] >> (16 * h)) & 0xFFFFu;
#pragma unroll
            for (int d = 0; d < 4; ++d) Bf[u][s][d] = pm64((bits >> (4 * d)) & 15u);
        }
    }
    v16i cin;
#pragma unroll
    for (int r = 0; r < 16; ++r) cin[r] = 4096 * 256 + 31 - ((r & 3) + 8 * (r >> 2));
    unsigned b[QS], s2[QS];
#pragma unroll
    for (int u = 0; u < QS; ++u) { b[u] = 0; s2[u] = 0; }
    const int ntiles = (ntf + 31) >> 5;
    const v4i* A = (const v4i*)(texp + (size_t)f * tilesCap * 8 * 1024) + lane;
    auto load_tile = [&](int T, v4i (&a)[8]) {
#pragma unroll
        for (int s = 0; s < 8; ++s) a[s] = A[((size_t)T * 8 + s) * 64];
    };
    auto do_tile = [&](int T, const v4i (&a)[8]) {
        const unsigned tk = 32u * (unsigned)(KNN_MFMA_MAX_TILES - 1 - T);
        const bool partial = T == ntiles - 1 && (ntf & 31);      // the last, partial tile: rows past the end are no candidates
#pragma unroll
        for (int u = 0; u < QS; ++u) {
            v16i acc = cin;
#pragma unroll
            for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], Bf[u][s], acc, 0, 0, 0);
            if (partial) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = T * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const unsigned key = row < ntf ? (unsigned)acc[r] + tk : 0u;
                    s2[u] = max(s2[u], min(b[u], key)); b[u] = max(b[u], key);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned key = (unsigned)acc[r] + tk;
                    s2[u] = max(s2[u], min(b[u], key)); b[u] = max(b[u], key);
                }
            }
        }
    };
    v4i a0[8], a1[8];
    if (ntiles > 0) load_tile(0, a0);
    for (int T = 0; T < ntiles; T += 2) {                   // two tiles per trip: the next tile's rows are in flight while this one is multiplied
        if (T + 1 < ntiles) load_tile(T + 1, a1);
        do_tile(T, a0);
        if (T + 2 < ntiles) load_tile(T + 2, a0);
        if (T + 1 < ntiles) do_tile(T + 1, a1);
    }
    // per lane: the best two of its 16 rows per tile.  To (512 - 2 h) << 16 | (0xFFFF - train index), then the two lanes of a query merge
    auto canon = [&](unsigned key) -> unsigned {
        if (key == 0u) return 0u;
        const unsigned T = (unsigned)(KNN_MFMA_MAX_TILES - 1) - ((key >> 5) & 127u);
        const unsigned ti = T * 32u + (31u - (key & 31u)) + 4u * (unsigned)h;
        return ((key >> 12) << 16) | (0xFFFFu - ti);
    };
#pragma unroll
    for (int u = 0; u < QS; ++u) {
        const unsigned cb = canon(b[u]), cs = canon(s2[u]);
        const unsigned ob = (unsigned)__shfl_xor((int)cb, 32, 64), os = (unsigned)__shfl_xor((int)cs, 32, 64);
        const unsigned best = max(cb, ob), second = max(min(cb, ob), max(cs, os));
        const int qi = q0 + 32 * u + j;
        if (h == 0 && qi < nqf) {
            const size_t o = ((size_t)f * cap + qi) * 2;
            idx[o] = best ? (int)(0xFFFFu - (best & 0xFFFFu)) : -1;
            dist[o] = best ? (int)((512u - (best >> 16)) >> 1) : -1;
            idx[o + 1] = second ? (int)(0xFFFFu - (second & 0xFFFFu)) : -1;
            dist[o + 1] = second ? (int)((512u - (second >> 16)) >> 1) : -1;
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
