// KeyLine fill + top-N, LBD pre-blur + Sobel (fused register sliding window), LBD band descriptor.
// Part of lines.hip (included there, inside its anonymous namespace: one translation unit, so device helpers are shared
// without relocatable device code).  Not a standalone header.
#pragma once

// ------------------------------------------------------------------ KeyLine fill + top-N (LSDDetector::detectImpl, ExtractLineSegment :42-51)
// bitonic sort of P2 (a power of two) 64-bit keys by the 256 threads of the workgroup, ascending
template <class KeyPtr>
__device__ __forceinline__ void keyline_sort(KeyPtr keys, int P2, int tid) {
    for (int k = 2; k <= P2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P2; i += 256) {
                int ixj = i ^ j;
                if (ixj > i) {
                    unsigned long long a = keys[i], c = keys[ixj];
                    bool up = (i & k) == 0;
                    if ((a > c) == up) { keys[i] = c; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
}

// The sort keys of up to KL_LDS accepted segments live in LDS (8 KB); a frame with more of them -- MAX_SEG is the hard limit -- sorts in
// the frame's workspace instead (the NFA states are dead by now).  Rounds 1-3 declared `__shared__ keys[MAX_SEG]`, 64 KB per workgroup:
// with the point branch's workgroups holding LDS on every CU, the line stream's 12 288 workgroups of this kernel could only trickle in
// (0.45 ms alone, 18.6 ms under the point branch).
constexpr int KL_LDS = 1024;
__global__ __launch_bounds__(256) void k_keylines(uint8_t* __restrict__ ws, LsdPlan P, int maxLines,
                                                  sslam_keyline* __restrict__ klOut, double* __restrict__ fnOut,
                                                  int* __restrict__ counts, int cap) {
    __shared__ unsigned long long keysL[KL_LDS];
    const int b = blockIdx.x, tid = threadIdx.x;
    uint8_t* base = ws + (size_t)b * P.frameBytes;
    float4* seg = (float4*)(base + P.offSeg);
    Misc* misc = (Misc*)(base + P.offMisc);
    sslam_keyline* klw = (sslam_keyline*)(base + P.offKl);
    // ordered compaction of the candidates the NFA stage accepted (seed order == the reference's emission order)
    __shared__ int wcnt[4];
    __shared__ int nAcc;
    {
        const int* flag = (const int*)(base + P.offFlag);
        const int nCand = misc->nCand;
        const int lane = tid & 63, wv = tid >> 6;
        int basePos = 0;
        for (int i0 = 0; i0 < nCand; i0 += 256) {
            const int i = i0 + tid;
            const bool ok = i < nCand && flag[i] != 0;
            const float4 v = ok ? seg[i] : make_float4(0, 0, 0, 0);
            const unsigned long long m = __ballot(ok);
            if (lane == 0) wcnt[wv] = __popcll(m);
            __syncthreads();                       // also orders the reads of seg[i0..i0+256) before the writes below (dst <= src)
            int off = basePos;
            for (int q = 0; q < wv; ++q) off += wcnt[q];
            if (ok) seg[off + mbcnt(m)] = v;
            basePos += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            __syncthreads();
        }
        if (tid == 0) { nAcc = basePos; misc->nSeg = basePos; }
        __syncthreads();
    }
    const int n = nAcc;
    static_assert(sizeof(NfaState) >= sizeof(unsigned long long), "the overflow sort buffer reuses the NFA states");
    unsigned long long* keysG = (unsigned long long*)(base + P.offNfa);      // MAX_SEG keys fit: sizeof(NfaState) >= 8
    const bool inLds = n <= KL_LDS;
    for (int i = tid; i < n; i += 256) {
        float4 s = seg[i];
        float e0 = s.x, e1 = s.y, e2 = s.z, e3 = s.w;
        const float W = (float)P.w, H = (float)P.h;           // checkLineExtremes
        if (e0 < 0) e0 = 0; if (e0 >= W) e0 = W - 1.0f;
        if (e2 < 0) e2 = 0; if (e2 >= W) e2 = W - 1.0f;
        if (e1 < 0) e1 = 0; if (e1 >= H) e1 = H - 1.0f;
        if (e3 < 0) e3 = 0; if (e3 >= H) e3 = H - 1.0f;
        sslam_keyline k;
        k.startPointX = e0; k.startPointY = e1; k.endPointX = e2; k.endPointY = e3;
        k.sPointInOctaveX = e0; k.sPointInOctaveY = e1; k.ePointInOctaveX = e2; k.ePointInOctaveY = e3;
        const double ddx = (double)__fsub_rn(e0, e2), ddy = (double)__fsub_rn(e1, e3);
        k.lineLength = (float)sqrt(ddx * ddx + ddy * ddy);
        const int ax = cv_roundf(e0), ay = cv_roundf(e1), bx = cv_roundf(e2), by = cv_roundf(e3);
        k.numOfPixels = max(abs(bx - ax), abs(by - ay)) + 1;
        k.angle = (float)atan2((double)__fsub_rn(e3, e1), (double)__fsub_rn(e2, e0));     // D5
        k.class_id = i; k.octave = 0;
        k.size = __fmul_rn(__fsub_rn(e2, e0), __fsub_rn(e3, e1));
        k.response = __fdiv_rn(k.lineLength, (float)max(P.w, P.h));
        k.pt_x = __fdiv_rn(__fadd_rn(e2, e0), 2.f); k.pt_y = __fdiv_rn(__fadd_rn(e3, e1), 2.f);
        klw[i] = k;
        const unsigned long long key = ((unsigned long long)(~__float_as_uint(k.response)) << 32) | (unsigned)i;   // response >= 0: descending response, ascending index (D3 stable)
        if (inLds) keysL[i] = key; else keysG[i] = key;
    }
    __syncthreads();
    int nOut = n;
    const bool doSort = n > maxLines;
    if (doSort) {
        int P2 = 1; while (P2 < n) P2 <<= 1;
        if (inLds) {
            for (int i = n + tid; i < P2; i += 256) keysL[i] = ~0ull;
            __syncthreads();
            keyline_sort(keysL, P2, tid);
        } else {
            for (int i = n + tid; i < P2; i += 256) keysG[i] = ~0ull;
            __syncthreads();
            keyline_sort(keysG, P2, tid);
        }
        nOut = maxLines;
    }
    nOut = min(nOut, cap);
    for (int i = tid; i < nOut; i += 256) {
        const int src = doSort ? (int)(unsigned)(inLds ? keysL[i] : keysG[i]) : i;
        sslam_keyline k = klw[src];
        if (doSort) k.class_id = i;
        klOut[(size_t)b * cap + i] = k;
        // the line's direction for k_lbd (D5: cos / sin in double, rounded to float), once per line by ONE lane here instead of by all 64 lanes of the line's wave there:
        // 200 of a line's ~2 100 vector instructions (the same expressions, the same bits)
        ((float2*)(base + P.offLbdDir))[i] = make_float2((float)cos((double)k.angle), (float)sin((double)k.angle));
        // line equation sp x ep, normalised by its first two components (ExtractLineSegment :56-68), fp64
        const double sx = k.startPointX, sy = k.startPointY, ex = k.endPointX, ey = k.endPointY;
        const double l0 = __dsub_rn(sy, ey), l1 = __dsub_rn(ex, sx), l2 = __dsub_rn(__dmul_rn(sx, ey), __dmul_rn(sy, ex));
        const double nrm = sqrt(__dadd_rn(__dmul_rn(l0, l0), __dmul_rn(l1, l1)));
        double* f = fnOut + ((size_t)b * cap + i) * 3;
        f[0] = l0 / nrm; f[1] = l1 / nrm; f[2] = l2 / nrm;
    }
    if (tid == 0) { counts[b] = nOut; misc->nKl = nOut; }
}

// ------------------------------------------------------------------ LBD front: 5x5 sigma-1 blur + Sobel 3x3 -> s16, fused
// BinaryDescriptor::computeGaussianPyramid (GaussianBlur 5x5, sigma 1) + cv::Sobel(CV_16S, ksize 3), both BORDER_REFLECT_101.
// The blurred image never reaches HBM: a 70x22 source tile (reflect-101) -> 66x22 horizontal pass -> 66x18 blurred tile in
// LDS -> 64x16 dx/dy.  (A symmetric kernel with reflect-101 borders commutes with the reflection, so evaluating the blur at the
// one-pixel Sobel halo outside the image from the reflected source IS the blurred value at the reflected pixel.)
__global__ __launch_bounds__(256) void k_blur_sobel(const uint8_t* __restrict__ src, size_t spitch, size_t sframe, int w, int h,
                                                    unsigned* __restrict__ dxyo, size_t dframeBytes,
                                                    const int* __restrict__ tapsArr) {
    // register sliding window (see k_blur7): 5x5 blur rows -> 3-row Sobel window.  Blurring the reflect-extended source with
    // symmetric taps equals reflect-extending the blurred image, which is what Sobel's BORDER_REFLECT_101 reads.
    constexpr int R = 3;                                   // blur radius 2 + Sobel radius 1
    const int ngroups = (w + 3) >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int strip = t / ngroups, x4 = (t - strip * ngroups) * 4, y0 = strip * STRIP;
    if (y0 >= h) return;
    const int b = blockIdx.y;
    const uint8_t* s = src + (size_t)b * sframe;
    unsigned taps[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) taps[k] = (unsigned)tapsArr[k];
    const unsigned T0 = taps[0] | (taps[1] << 8) | (taps[2] << 16) | (taps[3] << 24), T1 = taps[4];       // q8 taps < 256
    const bool fast = x4 >= 4 && x4 + 8 <= w && ((spitch | (size_t)(uintptr_t)s) & 3) == 0;
    const bool vec = ((w & 3) == 0) && ((dframeBytes & 15) == 0);
    unsigned hb[5][6];                                     // horizontally blurred rows, columns x4-1 .. x4+4
    int bl[3][6];                                          // blurred rows
    constexpr int NR = STRIP + 2 * R;
    const bool uni = row12_uniform(s, spitch, w);      // kernel-uniform: branch-free row loads, ROW_AHEAD_SOBEL rows in flight (lsd_front.h)
    const int x4u = uni ? x4 : 0;                       // (otherwise the ring reads the first bytes of the rows, unused, and load_row12 does the work)
    Row12 ring[ROW_AHEAD_SOBEL];
#pragma unroll
    for (int k = 0; k < ROW_AHEAD_SOBEL; ++k) ring[k] = row12_issue(s, spitch, reflect_row(y0 - R + k, h, R), x4u, w);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (r >= 2 * R && y0 + r - 2 * R >= h) break;
        Row12 cur = ring[r % ROW_AHEAD_SOBEL];
        if (r + ROW_AHEAD_SOBEL < NR) ring[r % ROW_AHEAD_SOBEL] = row12_issue(s, spitch, reflect_row(y0 - R + r + ROW_AHEAD_SOBEL, h, R), x4u, w);
        row12_fix(cur, x4u, w);
        if (!uni) load_row12(s, spitch, reflect_row(y0 - R + r, h, R), x4, w, fast, cur.d0, cur.d1, cur.d2);
        const unsigned d0 = cur.d0, d1 = cur.d1, d2 = cur.d2;
#pragma unroll
        for (int c = 0; c < 6; ++c) hb[r % 5][c] = hdot(d0, d1, d2, c + 1, T0, T1);      // columns x4+c-3 .. x4+c+1 (sums fit 16 bits: taps sum to 256 or 257)
        if (r >= 4) {
            const int q = r - 4;                           // blurred row y0 - 1 + q
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                unsigned acc = 0;
#pragma unroll
                for (int k = 0; k < 5; ++k) acc = __umul24(hb[(q + k) % 5][c], taps[k]) + acc;      // 16-bit sums x 8-bit taps: v_mad_u32_u24 (a 32-bit multiply is quarter rate)
                bl[q % 3][c] = (int)min((acc + 32768u) >> 16, 255u);      // (saturation only bites with taps that sum to 257: blur variant 1)
            }
            if (q >= 2) {
                const int y = y0 + q - 2;
                const int* A = bl[(q - 2) % 3];
                const int* M = bl[(q - 1) % 3];
                const int* C = bl[q % 3];
                short gx[4], gy[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    gx[j] = (short)((A[j + 2] - A[j]) + 2 * (M[j + 2] - M[j]) + (C[j + 2] - C[j]));
                    gy[j] = (short)((C[j] - A[j]) + 2 * (C[j + 1] - A[j + 1]) + (C[j + 2] - A[j + 2]));
                }
                // interleaved {dx, dy} int16 pairs: the LBD walk fetches both with one dword gather
                unsigned* op = (unsigned*)((uint8_t*)dxyo + (size_t)b * dframeBytes) + (size_t)y * w + x4;
                unsigned pk[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[j] = ((unsigned)(unsigned short)gx[j]) | ((unsigned)(unsigned short)gy[j] << 16);
                if (x4 + 3 < w && vec) *(uint4*)op = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (x4 + j < w) op[j] = pk[j];
                }
            }
        }
    }
}

// ------------------------------------------------------------------ LBD (BinaryDescriptor::computeLBD)
// One wave per line: lane = row of the 63-row line-support region, walking its row in
// the reference's order so every fp32 accumulation matches bit for bit; then 9 lanes
// fold rows into bands (again in row order), and the 72-float vector is normalised,
// clipped and binarised by lane 0..31.
__constant__ float kGaussL[21];
__constant__ float kGaussL2[21];      // kGaussL[j] * kGaussL[j], rounded once (the reference's c * c * x is (c * c) * x)
__constant__ float kGaussG[63];
__constant__ signed char kComb[64];

// Round 6, the walk's instruction count (25 vector instructions per step and row before; no gain while the vector L1 was the limit, see k_lbd):
//   * (int)roundf(x) clamped to [0, W] is ONE conversion + ONE median: v_cvt_rpi_i32_f32 is floor(x + 0.5) evaluated exactly -- for x >= 0 that IS round-half-away, for x < 0
//     both are <= 0 and the clamp makes them 0 -- (sslam_selftest_lbd_round compares it with the previous form on every float of the coordinate range).  The reference's
//     (short) cast is the identity while |x| < 32 768, i.e. for images of up to 16 384 pixels a side (a walk stays within half a line length + 32 of the image); larger
//     images take the previous form (RPI = false).
//   * (gDL, gDO) = (dx dL0 + dy dL1, dy dL0 - dx dL1) as two packed multiplies whose operand halves are picked by op_sel and ONE packed add with a negated upper half;
//     min(g, 0) = g - max(g, 0) (exact: g - g = +0, g - 0 = g; a zero of either sign leaves the sums, which never leave +0 .. +inf, unchanged): 10 instructions per step
//     instead of 12, 7 instead of 13 for the coordinates.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int cvt_rpi(float x) { int r; asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
// (a.x * b.x, a.y * b.x) and (a.y * b.y, a.x * b.y)
__device__ __forceinline__ f32x2 pk_mul_lo(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_mul_hi_swapped(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
// (a.x + b.x, a.y - b.y)
__device__ __forceinline__ f32x2 pk_add_sub(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
// max(x, 0) and the clamp to [0, hi] as the one instruction each is (behind an asm result the compiler canonicalises before fmaxf -- a v_max x, x per operand --, and it
// cannot prove 0 <= hi for a run-time hi, so min(max()) stays two instructions)
__device__ __forceinline__ float max0(float x) { float r; asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ int med3_0(int x, int hi) { int r; asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi)); return r; }
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

#ifdef SSLAM_TESTING
// sslam_selftest_lbd_round: every float bit pattern with |x| < 24 608 (= 16 384 + 8 192 + 32: what a walk can reach on the largest image the RPI form is used for), the
// conversion form against the previous one under the clamps [0, W] for three W, and against roundf itself for x >= 0.  Both counters must come back 0.
__global__ __launch_bounds__(256) void k_selftest_lbd_round(unsigned long long* __restrict__ bad) {
    unsigned long long nb0 = 0, nb1 = 0;
    const unsigned base = (blockIdx.x * 256u + threadIdx.x) << 8;
    for (unsigned i = 0; i < 256u; ++i) {
        const float x = __uint_as_float(base + i);
        if (!(fabsf(x) < 24608.f)) continue;
        const int t = (int)__fadd_rn(x, x);
        const int o = (int)(short)((t + 1) >> 1), n = cvt_rpi(x);
        const int Ws[3] = {0, 639, 16383};
#pragma unroll
        for (int k = 0; k < 3; ++k) nb0 += (min(max(o, 0), Ws[k]) != min(max(n, 0), Ws[k])) ? 1 : 0;
        if (x >= 0.f) nb1 += (n != (int)roundf(x)) ? 1 : 0;
    }
    if (nb0) atomicAdd(bad, nb0);
    if (nb1) atomicAdd(bad + 1, nb1);
}
#endif
// The walk runs in blocks of LBD_TB steps with the gathers TRANSPOSED through LDS (round 6).  With a lane per row, one gather instruction read one step of all 63 rows: for a
// horizontal line that is one column of 63 image rows = 63 different 64-byte lines, 38 on average over the bench's lines -- and the kernel ran at the vector L1's one line
// per cycle (~500 k line accesses per frame, 37 % of the L1's cycles in tag-conflict stalls: profiles/r05l_tcp_counters.txt), not at the vector pipes' rate (a third fewer
// instructions per step changed nothing: 10.2 -> 10.1 ms, GPU call W).  Here every lane still walks ITS row (the coordinates are sequential float additions, as the reference
// has them) but only writes the byte offsets of a block into LDS; the gathers then run over the block in a lane mapping chosen per line -- 2^lgS consecutive steps of
// 64 >> lgS rows per instruction, lgS minimising the image rows an instruction touches (8 steps x 8 rows for a horizontal line: 8 .. 16 lines instead of 63; one step x 64 rows
// for a vertical one, as before) --, park their dwords in LDS IN THE PLACE of the offsets, and every lane reads its row's values back in step order for the sums, whose order is
// unchanged.  One wave: its LDS operations execute in program order, so the phases need no barrier; nothing is in flight across blocks (eight waves per SIMD cover a block's round trip:
// a form with two arrays and the next block's gathers issued ahead measured the same, 8.5 - 8.7 ms).
// Measured per 12 288 frames (GPU calls X, Y, AF): a gather per step 9.9 - 10.2 ms; blocks of 4 / 8 / 16 steps 10.1 / 8.5 / 8.7 - 9.9 ms.  The vector L1 still sets the pace:
// 328 k line accesses per frame (503 k before: ~23 per gather instruction at the bench's line lengths and angles), 7.5 of the 8.5 ms at a line per cycle and compute unit.
// Measured earlier and not kept: a plane pitch with an odd number of lines per row (10.3 -> 10.0 ms, but k_blur_sobel 6.2 -> 6.9); the per-line set-up (fp64 cos / sin, the
// region's corner: 260 vector instructions per line) moved into k_keylines and read back with scalar loads (10.3 -> 11.1 ms in round 5; the cos / sin alone moved there in round 6: no loss, kept); a clamp-free walk for support regions inside
// the image (no gain, and its first form faulted on the steps past a line's end).
constexpr int LBD_TB = 8;
template <bool RPI>
__global__ __launch_bounds__(64) void k_lbd(const uint8_t* __restrict__ ws, LsdPlan P, const sslam_keyline* __restrict__ kls,
                                            const int* __restrict__ counts, uint8_t* __restrict__ descOut, int cap) {
    constexpr int TB = LBD_TB, TN = TB, TP = TB + 1;            // TP: LDS pitch of a row's block (odd: lanes a row apart fall into different banks)
    __shared__ unsigned tbuf[64 * TP];                          // a block's byte offsets (row, step), then its gathered dwords in their place
    float (*rows)[64] = (float (*)[64])&tbuf[0];                // after the walk: pgdL, ngdL, pgdL2, ngdL2, pgdO, ngdO, pgdO2, ngdO2 per row
    static_assert(8 * 64 <= 64 * TP, "rows[][] lies over the walk's block array");
    __shared__ float band[8][NUM_BANDS];
    __shared__ float des[72];
    const int li = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    if (li >= counts[b]) return;
    const uint8_t* base = ws + (size_t)b * P.frameBytes;
    const unsigned* dxyImg = (const unsigned*)(base + P.offDxy);      // {dx, dy} int16 pairs
    const sslam_keyline kl = kls[(size_t)b * cap + li];
    const int lengthOfLSP = (short)kl.numOfPixels;
    const int halfWidth = (lengthOfLSP - 1) / 2, halfHeight = (LSP_H - 1) / 2;
    const float midX = (float)(0.5 * (double)__fadd_rn(kl.sPointInOctaveX, kl.ePointInOctaveX));
    const float midY = (float)(0.5 * (double)__fadd_rn(kl.sPointInOctaveY, kl.ePointInOctaveY));
    const float2 dLs = ((const float2*)(base + P.offLbdDir))[li];      // (float)cos((double)kl.angle), (float)sin(..): D5, evaluated by k_keylines; a scalar load
    const float dL0 = dLs.x, dL1 = dLs.y;
    const float dO0 = -dL1, dO1 = dL0;
    const int realWidth = P.w, imageWidth = P.w - 1, imageHeight = P.h - 1;
    {
        const int rl = min(lane, LSP_H - 1);                // (lane 63 walks row 62 once more: the transposed gathers want 64 rows of valid offsets; its sums are never read)
        float sx0 = __fadd_rn(__fadd_rn(__fmul_rn(-dL0, (float)halfWidth), __fmul_rn(dL1, (float)halfHeight)), midX);
        float sy0 = __fadd_rn(__fsub_rn(__fmul_rn(-dL1, (float)halfWidth), __fmul_rn(dL0, (float)halfHeight)), midY);
        // row start: sCor0 after `lane` steps of (sCorX0 -= dL1, sCorY0 += dL0), sequential float operations; x - y == x + (-y) bit for bit
        float ndL1 = -dL1;
        asm volatile("" : "+v"(ndL1));                      // (opaque, or the compiler turns the addition back into a subtraction)
        {
            f32x2 s0 = {sx0, sy0}; const f32x2 dRow = {ndL1, dL0};
            for (int r = 0; r < rl; ++r) s0 = pk_add(s0, dRow);      // (both coordinates step by an addition: one packed instruction per row)
            sx0 = s0.x; sy0 = s0.y;
        }
        float sx = sx0, sy = sy0;
        float pL = 0, nL = 0, pO = 0, nO = 0;
        f32x2 accP = {0.f, 0.f}, accN = {0.f, 0.f};
        const f32x2 dLv = {dL0, dL1};
        // steps per gather instruction: 2^lgS (<= TB), rows: 64 >> lgS; an instruction touches about 2^lgS |dL1| + (64 >> lgS) |dL0| image rows
        int lgS = 0;
        {
            float best = 3.0e38f;
#pragma unroll
            for (int c = 0; (1 << c) <= TB; ++c) {
                const float cost = __fadd_rn(__fmul_rn((float)(1 << c), fabsf(dL1)), __fmul_rn((float)(64 >> c), fabsf(dL0)));
                if (cost < best) { best = cost; lgS = c; }
            }
        }
        lgS = __builtin_amdgcn_readfirstlane(lgS);
        const int S = 1 << lgS, R = 64 >> lgS;
        const unsigned laneIdx = (unsigned)(((lane >> lgS) * TP + (lane & (S - 1))) * 4);       // byte address of this lane's element of instruction 0 inside a block array
        // instruction k of a block (TB of them): row group k & (S - 1) (rows from (k & (S - 1)) * R), step group k >> lgS (steps from (k >> lgS) * S)
        unsigned ko[TN];                                    // (wave-uniform: scalar registers, computed once per line)
#pragma unroll
        for (int k = 0; k < TN; ++k) ko[k] = (unsigned)__builtin_amdgcn_readfirstlane((((k & (S - 1)) * R) * TP + ((k >> lgS) << lgS)) * 4);
        auto kofs = [&](int k) -> unsigned { return ko[k]; };
        // one wave: its LDS operations execute in program order, so the phases need no barrier (and no wait for the gathers in flight, which __syncthreads would bring) --
        // only that the compiler keeps the order
        auto lds_order = []() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
        char* offB = (char*)&tbuf[0]; char* valB = offB;
        unsigned g[TN];
        auto offsets = [&]() {                              // the next TB steps of my row -> tbuf[0][lane][0 .. TB)
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                // RPI = false, (int)roundf(x) without the seven-instruction round-half-away sequence: 2x is exact, t = trunc(2x), and round-half-away(x) = (t + 1) >> 1 for
                // t >= 0; for t < 0 both that and the exact form are <= 0 and the clamp below makes them 0.  The reference's (short) cast and its clamp
                // "tc < 0 ? 0 : tc > W ? W : tc" are a sign extension and a median of three.
                auto rnd = [](float x) -> int { const int t = (int)__fadd_rn(x, x); return (t + 1) >> 1; };
                const int xCor = RPI ? med3_0(cvt_rpi(sx), imageWidth) : min(max((int)(short)rnd(sx), 0), imageWidth);
                const int yCor = RPI ? med3_0(cvt_rpi(sy), imageHeight) : min(max((int)(short)rnd(sy), 0), imageHeight);
                tbuf[lane * TP + u] = (__umul24((unsigned)yCor, (unsigned)realWidth) + (unsigned)xCor) << 2;
                sx = __fadd_rn(sx, dL0); sy = __fadd_rn(sy, dL1);
            }
        };
        auto issue = [&]() {                                // the block's TB gathers in the transposed mapping (steps past the line's end: loaded, never consumed; the clamp keeps them addressable)
            unsigned o[TN];
#pragma unroll
            for (int k = 0; k < TB; ++k) o[k] = *(const unsigned*)(offB + laneIdx + kofs(k));
#pragma unroll
            for (int k = 0; k < TB; ++k) g[k] = *(const unsigned*)((const char*)dxyImg + o[k]);
        };
        auto park = [&]() {
#pragma unroll
            for (int k = 0; k < TB; ++k) *(unsigned*)(valB + laneIdx + kofs(k)) = g[k];
        };
        auto consume = [&](int w0) {
            unsigned v[TN];
#pragma unroll
            for (int u = 0; u < TB; ++u) v[u] = tbuf[lane * TP + u];
            const bool whole = w0 + TB <= lengthOfLSP;
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                if (whole || w0 + u < lengthOfLSP) {
                    const float dx = (float)(short)(v[u] & 0xFFFFu), dy = (float)(short)(v[u] >> 16);
                    // "if (g > 0) p += g; else n -= g" without the branch: the side that is not taken adds / subtracts a zero, which leaves a sum that started at +0 and only
                    // ever took non-negative addends unchanged bit for bit (x + (+-0) == x; +0 + (+-0) == +0 under round-to-nearest)
                    if (RPI) {
                        const f32x2 d = {dx, dy};
                        const f32x2 G = pk_add_sub(pk_mul_lo(d, dLv), pk_mul_hi_swapped(d, dLv));      // (gDL, gDO)
                        const f32x2 M = {max0(G.x), max0(G.y)};
                        accP = pk_add(accP, M); accN = pk_sub(accN, pk_sub(G, M));
                    } else {
                        const float gDL = __fadd_rn(__fmul_rn(dx, dL0), __fmul_rn(dy, dL1));
                        const float gDO = __fadd_rn(__fmul_rn(dx, dO0), __fmul_rn(dy, dO1));
                        pL = __fadd_rn(pL, fmaxf(gDL, 0.f)); nL = __fsub_rn(nL, fminf(gDL, 0.f));
                        pO = __fadd_rn(pO, fmaxf(gDO, 0.f)); nO = __fsub_rn(nO, fminf(gDO, 0.f));
                    }
                }
            }
        };
        for (int w0 = 0; w0 < lengthOfLSP; w0 += TB) {
            lds_order();                                    // (the previous block's values are read: before they are overwritten)
            offsets();
            lds_order();
            issue();                                        // (all offsets are in registers before ...)
            lds_order();
            park();                                         // (... the values take their place)
            lds_order();
            consume(w0);
        }
        __syncthreads();                                    // rows[][] lies over tbuf
        if (RPI) { pL = accP.x; pO = accP.y; nL = accN.x; nO = accN.y; }
        const float cg = kGaussG[rl];
        pL = __fmul_rn(cg, pL); nL = __fmul_rn(cg, nL); pO = __fmul_rn(cg, pO); nO = __fmul_rn(cg, nO);
        rows[0][lane] = pL; rows[1][lane] = nL; rows[2][lane] = __fmul_rn(pL, pL); rows[3][lane] = __fmul_rn(nL, nL);
        rows[4][lane] = pO; rows[5][lane] = nO; rows[6][lane] = __fmul_rn(pO, pO); rows[7][lane] = __fmul_rn(nO, nO);
    }
    __syncthreads();
    // band sums: lane -> (pair p = lane / 9 of {pgdL, ngdL, pgdO, ngdO}, band = lane % 9), each lane folding the mean AND the squared quantity; rows visited in increasing
    // hID so the accumulation order equals the reference's (a row adds to its own band and to the bands above and below, one accumulator per band).  Row hID of the three
    // bands around `bd` carries gaussCoefL[hID - 7 * (bd - 1)] (entries 0..6: the band above contributes "downward", 7..13: own band, 14..20: the band below "upward"), so
    // the 21 steps are the same for every lane: scalar coefficients, LDS reads at constant offsets, no division by the band width and no table gather (round 5: the
    // previous loop spent ~1 000 of a line's ~3 200 vector instructions here: 634 k -> 415 k per frame of 200 lines, profiles/r05_pmc_sq_table.txt).
    if (lane < 4 * NUM_BANDS) {
        const int p = lane / NUM_BANDS, bd = lane - p * NUM_BANDS;
        const int q = (p & 1) + 4 * (p >> 1);               // rows[] of the mean quantity (0, 1, 4, 5); its square is rows[q + 2]
        const float* rf = &rows[0][0];
        const int r0 = q * 64 + (bd - 1) * BAND_W;          // row hID = (bd - 1) * 7 + j of quantity q
        float acc = 0, acc2 = 0;
        auto fold = [&](int j) { const float v = rf[r0 + j], v2 = rf[r0 + j + 2 * 64]; acc = __fadd_rn(acc, __fmul_rn(kGaussL[j], v)); acc2 = __fadd_rn(acc2, __fmul_rn(kGaussL2[j], v2)); };
        if (bd > 0) {
#pragma unroll
            for (int j = 0; j < BAND_W; ++j) fold(j);
        }
#pragma unroll
        for (int j = BAND_W; j < 2 * BAND_W; ++j) fold(j);
        if (bd < NUM_BANDS - 1) {
#pragma unroll
            for (int j = 2 * BAND_W; j < 3 * BAND_W; ++j) fold(j);
        }
        band[q][bd] = acc; band[q + 2][bd] = acc2;
    }
    __syncthreads();
    // sqrtf, not __fsqrt_rn: HIP maps the latter to the native (1-ulp) v_sqrt_f32, the former is correctly rounded
    if (lane < NUM_BANDS) {
        const int bd = lane;
        const float invN = (bd == 0 || bd == NUM_BANDS - 1) ? (float)(1.0 / (BAND_W * 2.0)) : (float)(1.0 / (BAND_W * 3.0));
        float t;
        t = __fmul_rn(band[0][bd], invN); des[bd * 8 + 0] = t; des[bd * 8 + 4] = sqrtf(__fsub_rn(__fmul_rn(band[2][bd], invN), __fmul_rn(t, t)));
        t = __fmul_rn(band[1][bd], invN); des[bd * 8 + 1] = t; des[bd * 8 + 5] = sqrtf(__fsub_rn(__fmul_rn(band[3][bd], invN), __fmul_rn(t, t)));
        t = __fmul_rn(band[4][bd], invN); des[bd * 8 + 2] = t; des[bd * 8 + 6] = sqrtf(__fsub_rn(__fmul_rn(band[6][bd], invN), __fmul_rn(t, t)));
        t = __fmul_rn(band[5][bd], invN); des[bd * 8 + 3] = t; des[bd * 8 + 7] = sqrtf(__fsub_rn(__fmul_rn(band[7][bd], invN), __fmul_rn(t, t)));
    }
    __syncthreads();
    // normalise means / stds separately, clip at 0.4, renormalise: sequential sums (every lane redundantly)
    float tempM = 0, tempS = 0;
    {   // the two sums advance in lockstep (means d[0..3], deviations d[4..7] of a band): one packed multiply and one packed add per pair (d[i], d[i + 4]), each sum in its own order
        f32x2 acc = {0.f, 0.f};
#pragma unroll
        for (int bd = 0; bd < NUM_BANDS; ++bd) {
            const float* d = des + bd * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const f32x2 v = {d[i], d[i + 4]}; acc = pk_add(acc, pk_mul(v, v)); }
        }
        tempM = acc.x; tempS = acc.y;
    }
    tempM = __fdiv_rn(1.f, sqrtf(tempM)); tempS = __fdiv_rn(1.f, sqrtf(tempS));
    __syncthreads();
    for (int i = lane; i < 72; i += 64) {
        float v = des[i];
        v = ((i & 7) < 4) ? __fmul_rn(v, tempM) : __fmul_rn(v, tempS);
        if (v > 0.4f) v = 0.4f;
        des[i] = v;
    }
    __syncthreads();
    float temp = 0;
#pragma unroll
    for (int i = 0; i < 72; i += 2) {       // squares two at a time (packed), the sum in index order
        const f32x2 v = {des[i], des[i + 1]}; const f32x2 q = pk_mul(v, v);
        temp = __fadd_rn(temp, q.x); temp = __fadd_rn(temp, q.y);
    }
    temp = __fdiv_rn(1.f, sqrtf(temp));
    __syncthreads();
    for (int i = lane; i < 72; i += 64) des[i] = __fmul_rn(des[i], temp);
    __syncthreads();
#ifdef SSLAM_LBD_DEBUG
    for (int i = lane; i < 72; i += 64) ((float*)(const_cast<uint8_t*>(base) + P.offCand))[li * 72 + i] = des[i];      // normalised 72-float vector (candidate buffer is free here)
#endif
    if (lane < 32) {
        const float* f1 = des + 8 * kComb[lane * 2];
        const float* f2 = des + 8 * kComb[lane * 2 + 1];
        unsigned r = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) if (f1[i] > f2[i]) r += 1u << i;
        if (P.lbdBitOrder) r = __brev(r) >> 24;          // decision D12's alternative: comparison i -> 0x80 >> i
        descOut[((size_t)b * cap + li) * 32 + lane] = (uint8_t)r;
    }
}
