// LSD front: 7x7 pre-blur (register sliding window), exact 0.8x resample + gradient / level-line angle, stable counting sort of the seeds.
// Part of lines.hip (included there, inside its anonymous namespace: one translation unit, so device helpers are shared
// without relocatable device code).  Not a standalone header.
#pragma once

// ------------------------------------------------------------------ separable blur (q8 taps, D6)
// Stencil kernels here are register sliding windows: one thread owns four adjacent columns of a strip of STRIP rows, reads
// each source row once as three aligned dwords (columns x-4 .. x+7), keeps the horizontally filtered rows it still needs
// in registers and emits one packed store per output row.  No LDS, no barriers, and the unrolled row loop keeps many
// loads in flight (these kernels are latency-bound, not byte-bound).
constexpr int STRIP = 32;

// source columns x4-4 .. x4+7 of row yy as three dwords (BORDER_REFLECT_101 in x for the threads that touch the border)
__device__ __forceinline__ void load_row12(const uint8_t* __restrict__ s, size_t spitch, int yy, int x4, int w, bool fast, unsigned& d0, unsigned& d1, unsigned& d2) {
    const uint8_t* row = s + (size_t)yy * spitch;
    if (fast) {
        const unsigned* q = (const unsigned*)(row + x4 - 4);
        d0 = q[0]; d1 = q[1]; d2 = q[2];
    } else {
        unsigned px[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            int xx = reflect101(min(x4 - 4 + i, w + 2), w);
            xx = min(max(xx, 0), w - 1);
            px[i] = row[xx];
        }
        d0 = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
        d1 = px[4] | (px[5] << 8) | (px[6] << 16) | (px[7] << 24);
        d2 = px[8] | (px[9] << 8) | (px[10] << 16) | (px[11] << 24);
    }
}
// Branch-free form of load_row12 for images whose width is a multiple of four (>= 12) with dword-aligned base and pitch: every lane loads three
// aligned dwords -- the lanes at the left / right image border re-read an in-range dword in place of the one outside and build the reflected
// bytes (BORDER_REFLECT_101: columns -1, -2, -3 = 1, 2, 3; w, w + 1, w + 2 = w - 2, w - 3, w - 4) with one v_perm after the data arrived.  Without
// a branch around the loads the rows of a strip can be requested ahead of the arithmetic (ROW_AHEAD_* rows requested ahead per lane, and the scheduler is free to hoist more): with the branch
// every row waited for its own round trip, 38 of them per strip (round 4, docs/history/DESIGN_rounds_1-4.md 5g).
struct Row12 { unsigned d0, d1, d2; };
// rows requested ahead per lane, per kernel (measured, profiles/r04_kernel_variants.txt: more rows in flight cost registers, i.e. resident waves)
constexpr int ROW_AHEAD_BLUR7 = 1, ROW_AHEAD_SOBEL = 2;
__device__ __forceinline__ bool row12_uniform(const uint8_t* s, size_t spitch, int w) { return (w & 3) == 0 && w >= 12 && ((spitch | (size_t)(uintptr_t)s) & 3) == 0; }
__device__ __forceinline__ Row12 row12_issue(const uint8_t* __restrict__ s, size_t spitch, int yy, int x4, int w) {
    const unsigned* q = (const unsigned*)(s + (size_t)yy * spitch + x4);
    Row12 r; r.d1 = q[0]; r.d0 = q[x4 < 4 ? 0 : -1]; r.d2 = q[x4 + 8 > w ? 0 : 1];
    return r;
}
__device__ __forceinline__ void row12_fix(Row12& r, int x4, int w) {
    const unsigned l = __builtin_amdgcn_perm(r.d2, r.d1, 0x01020304u);       // columns -4 .. -1 <- 4, 3, 2, 1
    const unsigned rr = __builtin_amdgcn_perm(r.d1, r.d0, 0x03040506u);      // columns w .. w + 3 <- w - 2, w - 3, w - 4, w - 5
    r.d0 = x4 < 4 ? l : r.d0; r.d2 = x4 + 8 > w ? rr : r.d2;
}
// sum over k of tap[k] * byte[s + k] of the 12 bytes d0:d1:d2, taps packed four to a dword (v_dot4_u32_u8); s = 1 .. 6,
// the taps beyond the kernel length are 0 so the bytes they meet do not matter
__device__ __forceinline__ unsigned hdot(unsigned d0, unsigned d1, unsigned d2, int s, unsigned T0, unsigned T1) {
    const unsigned lo = s < 4 ? __builtin_amdgcn_alignbyte(d1, d0, (unsigned)s) : __builtin_amdgcn_alignbyte(d2, d1, (unsigned)(s - 4));
    const unsigned hi = s < 4 ? __builtin_amdgcn_alignbyte(d2, d1, (unsigned)s) : (d2 >> (8 * (s - 4)));
    return __builtin_amdgcn_udot4(lo, T0, __builtin_amdgcn_udot4(hi, T1, 0u, false), false);
}
__device__ __forceinline__ int reflect_row(int y, int h, int R) {
    const int yy = reflect101(min(y, h + R - 1), h);
    return min(max(yy, 0), h - 1);
}

// 7x7 Gaussian (LSD's sigma = 0.6/0.8 pre-blur), 16.16 accumulation as in D6
__global__ __launch_bounds__(256) void k_blur7(const uint8_t* __restrict__ src, size_t spitch, size_t sframe,
                                               uint8_t* __restrict__ dst, size_t dpitch, size_t dframe, int w, int h,
                                               const int* __restrict__ tapsArr) {
    constexpr int R = 3;
    const int ngroups = (w + 3) >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int strip = t / ngroups, x4 = (t - strip * ngroups) * 4, y0 = strip * STRIP;
    if (y0 >= h) return;
    const int b = blockIdx.y;
    const uint8_t* s = src + (size_t)b * sframe;
    uint8_t* d = dst + (size_t)b * dframe;
    unsigned taps[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) taps[k] = (unsigned)tapsArr[k];
    const unsigned T0 = taps[0] | (taps[1] << 8) | (taps[2] << 16) | (taps[3] << 24), T1 = taps[4] | (taps[5] << 8) | (taps[6] << 16);   // q8 taps < 256
    const bool fast = x4 >= 4 && x4 + 8 <= w && ((spitch | (size_t)(uintptr_t)s) & 3) == 0;
    unsigned win[7][4];
    constexpr int NR = STRIP + 2 * R;
    const bool uni = row12_uniform(s, spitch, w);      // kernel-uniform: branch-free row loads, ROW_AHEAD_BLUR7 rows in flight (lsd_front.h)
    const int x4u = uni ? x4 : 0;                       // (otherwise the ring reads the first bytes of the rows, unused, and load_row12 does the work)
    Row12 ring[ROW_AHEAD_BLUR7];
#pragma unroll
    for (int k = 0; k < ROW_AHEAD_BLUR7; ++k) ring[k] = row12_issue(s, spitch, reflect_row(y0 - R + k, h, R), x4u, w);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (r >= 2 * R && y0 + r - 2 * R >= h) break;
        Row12 cur = ring[r % ROW_AHEAD_BLUR7];
        if (r + ROW_AHEAD_BLUR7 < NR) ring[r % ROW_AHEAD_BLUR7] = row12_issue(s, spitch, reflect_row(y0 - R + r + ROW_AHEAD_BLUR7, h, R), x4u, w);
        row12_fix(cur, x4u, w);
        if (!uni) load_row12(s, spitch, reflect_row(y0 - R + r, h, R), x4, w, fast, cur.d0, cur.d1, cur.d2);
        const unsigned d0 = cur.d0, d1 = cur.d1, d2 = cur.d2;
#pragma unroll
        for (int j = 0; j < 4; ++j) win[r % 7][j] = hdot(d0, d1, d2, j + 1, T0, T1);      // columns x4+j-3 .. x4+j+3
        if (r >= 2 * R) {
            const int y = y0 + r - 2 * R;
            unsigned o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned acc = 0;
#pragma unroll
                for (int k = 0; k < 7; ++k) acc = __umul24(win[(r - 2 * R + k) % 7][j], taps[k]) + acc;      // 16-bit sums x 8-bit taps: v_mad_u32_u24 (a 32-bit multiply is quarter rate)
                o[j] = min((acc + 32768u) >> 16, 255u);
            }
            uint8_t* dp = d + (size_t)y * dpitch + x4;
            if (x4 + 3 < w) *(unsigned*)dp = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);      // dpitch % 64 == 0, x4 % 4 == 0
            else { dp[0] = (uint8_t)o[0]; if (x4 + 1 < w) dp[1] = (uint8_t)o[1]; if (x4 + 2 < w) dp[2] = (uint8_t)o[2]; }
        }
    }
}

// ------------------------------------------------------------------ INTER_LINEAR_EXACT 0.8x (D7), consumed inside k_lsd_grad
// tx/ty entries: {source offset, coefficient of the second tap (q8)}; source rows are read as three aligned dwords.
__device__ __forceinline__ unsigned pick2(unsigned d0, unsigned d1, unsigned d2, int o) {      // bytes o, o+1 of d0:d1:d2 (o <= 10)
    const unsigned long long w01 = (unsigned long long)d0 | ((unsigned long long)d1 << 32);
    const unsigned long long w12 = (unsigned long long)d1 | ((unsigned long long)d2 << 32);
    return (unsigned)((o < 4 ? w01 : w12) >> (8 * (o < 4 ? o : o - 4)));
}
// ------------------------------------------------------------------ gradient / level-line angle (ll_angle)
// The 2x2 gradient of an 8-bit image takes only 1021 x 1021 values, so angle (exact fastAtan2), the defined test
// (|g|/2 > rho) and the D5 cos/sin of the angle are tabulated once per process; the per-frame kernel is then a gather.
constexpr int GT = 1021;          // gx, gy in [-510, 510]
__global__ void k_grad_table(float4* __restrict__ tab, double rho) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= GT * GT) return;
    const int gy = i / GT - 510, gx = i - (i / GT) * GT - 510;
    const int s = gx * gx + gy * gy;
    float a = NOTDEF_F, cs = 0.f, sn = 0.f;
    if (sqrt((double)s / 4.0) > rho) {
        a = fast_atan2_deg((float)gx, (float)(-gy));
        const float af = (float)((double)a * DEG2RAD);
        cs = (float)cos((double)af); sn = (float)sin((double)af);
    }
    tab[i] = make_float4(a, cs, sn, __int_as_float(s));
}

// smallest |g|^2 of a DEFINED pixel: "sqrt(s / 4.0) > rho" is monotone in the integer s, so the table's test is "s >= sMin"
__global__ void k_grad_smin(int* __restrict__ out, double rho) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s <= 2 * 510 * 510 && sqrt((double)s / 4.0) > rho) atomicMin(out, s);
}

// One thread = four horizontally adjacent pixels, one wave = 256 pixels of one row (a SEGMENT): the 2x5 source bytes come from two dword +
// two byte loads.  Round 4 (second half): only 10-20 % of the pixels of a frame are DEFINED (|g|/2 > rho), and everything but the angle plane
// is read for defined pixels only, so
//   * T (angle, NOTDEF) is the one plane written densely (4 B per pixel);
//   * the table gather, the Cs pair and the S value exist only for lanes that hold a defined pixel (undefined pixels keep whatever the
//     workspace held: region growing tests T before it looks at Cs, region2rect reads S at region pixels);
//   * the counting sort no longer streams a dense S plane twice: each wave appends its defined pixels, in raster order, to the segment's
//     list (ballot-free: a DPP prefix sum of the lanes' counts) as |g|^2 << 8 | column-in-segment, and stores the segment's count.
// 16 B written per pixel in round 3 (24 in rounds 1-2); now 4 B + ~24 B per defined pixel.
// The 0.8x INTER_LINEAR_EXACT image (D7) is never stored: the 2 x 5 scaled pixels a thread needs are recomputed here from the blurred
// source (four source rows as three aligned dwords each).
// LIN (decision D7's alternative, LsdPlan::lsdResize = 1): cv::resize(..., INTER_LINEAR) for 8u -- cx = a0 | a1 << 16, cy = b0 | b1 << 16 (the 11-bit coefficient PAIRS of the
// reference's ialpha / ibeta: each is rounded on its own, they need not sum to 2048) and the 8u two-stage rounding of its vertical pass (oracle/cvleaf.h resize_linear_8u).
template <bool LIN>
__device__ __forceinline__ int scaled_px(unsigned e0, unsigned e1, unsigned cx, unsigned cy) {      // e = {p0, p1} bytes of the two source rows
    if (LIN) {
        const unsigned a0 = cx & 0xFFFFu, a1 = cx >> 16, b0 = cy & 0xFFFFu, b1 = cy >> 16;
        const unsigned r0 = __umul24(e0 & 255u, a0) + __umul24((e0 >> 8) & 255u, a1), r1 = __umul24(e1 & 255u, a0) + __umul24((e1 >> 8) & 255u, a1);
        return (int)(((__umul24(b0, r0 >> 4) >> 16) + (__umul24(b1, r1 >> 4) >> 16) + 2u) >> 2);
    }
    // (every factor is below 2^17: v_mul_u32_u24 / v_mad_u32_u24, full rate; a 32-bit multiply is quarter rate)
    const unsigned r0 = __umul24(e0 & 255u, 256u - cx) + __umul24((e0 >> 8) & 255u, cx), r1 = __umul24(e1 & 255u, 256u - cx) + __umul24((e1 >> 8) & 255u, cx);
    return (int)((__umul24(r0, 256u - cy) + __umul24(r1, cy) + 32768u) >> 16);
}
#ifndef SSLAM_GRAD_ROWS
#define SSLAM_GRAD_ROWS 8
#endif
#ifndef SSLAM_GRAD_WAVES
#define SSLAM_GRAD_WAVES 0
#endif
constexpr int GRAD_ROWS = SSLAM_GRAD_ROWS;            // output rows a wave walks down: scaled row y + 1 of one step is scaled row y of the next
#if SSLAM_GRAD_WAVES
__attribute__((amdgpu_waves_per_eu(SSLAM_GRAD_WAVES, SSLAM_GRAD_WAVES)))
#endif
// MODE bit 0 = LIN (D7's alternative, see scaled_px); bit 1 = DENSE (D2's alternative, sslam_lines_set_seed_order(1)): S holds |g|^2 of EVERY pixel, because upstream's
// std::sort permutes the undefined pixels along with the defined ones and the host needs all the keys.
template <int MODE>
__global__ __launch_bounds__(256) void k_lsd_grad(const uint8_t* __restrict__ ws, LsdPlan P, const float4* __restrict__ gtab, size_t bpitch,
                                                  const int* __restrict__ tx, const int* __restrict__ ty) {
    constexpr bool LIN = (MODE & 1) != 0, DENSE = (MODE & 2) != 0;
    const int b = blockIdx.z;
    const uint8_t* base = ws + (size_t)b * P.frameBytes;
    const uint8_t* src = base + P.offBlur;
    float* T = (float*)(base + P.offT);
    float2* Cs = (float2*)(base + P.offCs);
    int* S = (int*)(base + P.offS); (void)S;
    Misc* misc = (Misc*)(base + P.offMisc);
    unsigned* comp = (unsigned*)(base + P.offComp);
    int* segCnt = (int*)(base + P.offSegCnt);
    const int lane = threadIdx.x;
    const int yBeg = __builtin_amdgcn_readfirstlane((blockIdx.y * 4 + threadIdx.y) * GRAD_ROWS), yEnd = min(P.sh, yBeg + GRAD_ROWS);
    if (yBeg >= P.sh) return;                               // the whole wave
    const int x4 = (blockIdx.x * 64 + lane) * 4;
    const bool inx = x4 < P.sw;                             // lanes beyond the row compute on clamped columns and store nothing
    int ofs[5]; unsigned cx[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) { const int2 t = ((const int2*)tx)[min(x4 + j, P.sw - 1)]; ofs[j] = t.x; cx[j] = (unsigned)t.y; }
    const int a = ofs[0] & ~3;
    // bpitch % 64 == 0 and another buffer follows the last row: whole dwords are readable.  A 0.8x row needs at most 10 bytes from `a`.
    const bool fast = __builtin_amdgcn_ballot_w64(ofs[4] - a <= 10) == ~0ull;
    const int2* tyv = (const int2*)ty;
    auto load_raw = [&](int yy, unsigned (&d)[2][3]) {
        const int r = tyv[yy].x;
        const unsigned* q0 = (const unsigned*)(src + (size_t)r * bpitch + a);
        const unsigned* q1 = (const unsigned*)(src + (size_t)min(r + 1, P.h - 1) * bpitch + a);
#pragma unroll
        for (int k = 0; k < 3; ++k) { d[0][k] = q0[k]; d[1][k] = q1[k]; }
    };
    auto scale_fast = [&](int yy, const unsigned (&d)[2][3], int (&p)[5]) {
        const unsigned cy = (unsigned)tyv[yy].y;
#pragma unroll
        for (int j = 0; j < 5; ++j) { const int o = ofs[j] - a; p[j] = scaled_px<LIN>(pick2(d[0][0], d[0][1], d[0][2], o), pick2(d[1][0], d[1][1], d[1][2], o), cx[j], cy); }      // second tap has weight 0 at the last column
    };
    auto scale_slow = [&](int yy, int (&p)[5]) {
        const int2 t = tyv[yy];
        const uint8_t* r0 = src + (size_t)t.x * bpitch; const uint8_t* r1 = src + (size_t)min(t.x + 1, P.h - 1) * bpitch;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int sx = ofs[j], sx1 = min(sx + 1, P.w - 1);
            p[j] = scaled_px<LIN>(r0[sx] | ((unsigned)r0[sx1] << 8), r1[sx] | ((unsigned)r1[sx1] << 8), cx[j], (unsigned)t.y);
        }
    };
    int smax = 0;
    auto process = [&](int y, const int (&p0)[5], const int (&p1)[5]) {      // p0 / p1: the scaled rows y and y + 1
        float ang[4]; float2 cs[4]; int sv[4], gidx[4];
        unsigned flags = 0;                                 // bit j: pixel x4 + j is defined
        const bool lastRow = y >= P.sh - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ang[j] = NOTDEF_F; cs[j] = make_float2(0.f, 0.f); sv[j] = -1;
            const int DA = p1[j + 1] - p0[j], BC = p0[j + 1] - p1[j];
            const int gx = DA + BC, gy = DA - BC, s = gx * gx + gy * gy;
            gidx[j] = (gy + 510) * GT + (gx + 510);
            if (inx && !lastRow && x4 + j < P.sw - 1 && s >= P.sMin) { flags |= 1u << j; sv[j] = s; }
            else if (DENSE) sv[j] = s;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (flags & (1u << j)) { const float4 rec = gtab[gidx[j]]; ang[j] = rec.x; cs[j] = make_float2(rec.y, rec.z); }      // the four gathers are in flight together
        const size_t i = (size_t)y * P.sw + x4;
        if ((P.sw & 3) == 0) {
            if (inx) *(float4*)(T + i) = make_float4(ang[0], ang[1], ang[2], ang[3]);
            if (DENSE && inx) *(int4*)(S + i) = make_int4(sv[0], sv[1], sv[2], sv[3]);
            if (flags) {
                if (!DENSE) *(int4*)(S + i) = make_int4(sv[0], sv[1], sv[2], sv[3]);
                float4* c = (float4*)(Cs + i);
                c[0] = make_float4(cs[0].x, cs[0].y, cs[1].x, cs[1].y);
                c[1] = make_float4(cs[2].x, cs[2].y, cs[3].x, cs[3].y);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (x4 + j < P.sw) {
                T[i + j] = ang[j];
                if (DENSE) S[i + j] = sv[j];
                if (flags & (1u << j)) { S[i + j] = sv[j]; Cs[i + j] = cs[j]; }
            }
        }
        // the segment's list of defined pixels, raster order = lane order, then column inside the lane
        const int seg = y * P.nXB + blockIdx.x;
        const int cnt = __popc(flags), incl = wave_incl_scan(cnt);
        unsigned* dst = comp + ((size_t)seg << 8);
        int pos = incl - cnt;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (flags & (1u << j)) dst[pos++] = ((unsigned)sv[j] << 8) | (unsigned)(4 * lane + j);
        if (lane == 63) segCnt[seg] = incl;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (!DENSE || (flags & (1u << j))) smax = max(smax, sv[j]);
    };
    int pc[5], pn[5];
    if (fast) {
        unsigned d0[2][3], d1[2][3];
        load_raw(yBeg, d0); load_raw(min(yBeg + 1, P.sh - 1), d1);
        scale_fast(yBeg, d0, pc);
        for (int y = yBeg; y < yEnd; ++y) {
            scale_fast(min(y + 1, P.sh - 1), d1, pn);
            load_raw(min(y + 2, P.sh - 1), d1);            // in flight while this row's gathers and stores are
            process(y, pc, pn);
#pragma unroll
            for (int j = 0; j < 5; ++j) pc[j] = pn[j];
        }
    } else {
        scale_slow(yBeg, pc);
        for (int y = yBeg; y < yEnd; ++y) {
            scale_slow(min(y + 1, P.sh - 1), pn);
            process(y, pc, pn);
#pragma unroll
            for (int j = 0; j < 5; ++j) pc[j] = pn[j];
        }
    }
    smax = wave_max(smax);
    if (lane == 0 && smax > 0) atomicMax(&misc->maxS, smax);
}


// ------------------------------------------------------------------ round 6: the pre-blur INSIDE the gradient kernel
// k_blur7 + k_lsd_grad as one kernel for the geometry the bench and the reference's camera have: w = 5m, sw = 4m, h = 5n, sh = 4n (640x480, 1280x960, 320x240 ...), the
// blur's outer taps zero (0 4 56 136 56 4 0 under both D6 variants) and resize tables of the plain 4-to-5 pattern (lines_build_plan checks all of it on the host; anything
// else keeps the two kernels).  The 8-bit Gaussian is exact integer arithmetic with ONE rounding at the end -- blurred = min((sum_k sum_l t[k] t[l] raw[r+k][c+l] + 32768)
// >> 16, 255) -- so its 5 x 5 sum can be evaluated wherever it is needed.  A lane (four output pixels of one row) needs the blurred columns c0 .. c0 + 6, c0 = 5 x4 / 4, of
// the blurred rows 10 q .. 10 q + 11 for its strip of eight output rows (q = strip index): sixteen raw rows 10 q - 2 .. 10 q + 13, each read as one 16-byte window, their
// horizontal 5-tap sums H (<= 65 280: two per dword) kept in a ring of five rows, a blurred row finished whenever its fifth H row arrives, a scaled row whenever its two
// blurred rows exist.  What the fusion removes: k_blur7's launch, the blurred image's write and its 2.5-fold re-read, and -- in the two-stream step -- the kernel the pyramid's
// k_resize launches stretched most (3.5 -> 9.0 ms).
// What bounds it at 11 ms (timing builds, call S): its stores -- 8.2 ms without the dense T plane, 8.2 ms without the sparse Cs / S planes; the table gathers cost 0.7 ms, more raw rows in flight nothing.
// BORDER_REFLECT_101: rows through reflect_row; columns only touch the row's first lane (c0 = 0: raw columns -2, -1 = 2, 1) and its last one (c0 = w - 5: columns w .. w + 3
// = w - 2 .. w - 5), each fixed with one v_perm on in-range dwords.
struct Raw16 { unsigned v[4]; };
template <int MODE>
__global__ __launch_bounds__(256) void k_lsd_grad_fused(const uint8_t* __restrict__ img, size_t ipitch, size_t iframe, const uint8_t* __restrict__ ws, LsdPlan P,
                                                        const float4* __restrict__ gtab, const int* __restrict__ tx, const int* __restrict__ ty, const int* __restrict__ tapsArr, int nframes) {
    constexpr bool LIN = (MODE & 1) != 0, DENSE = (MODE & 2) != 0;
    // gridDim.z may be smaller than the batch: the workgroups then walk the frames with that stride (lines.hip: a grid the chip holds at once, so that another stream's
    // kernels are dispatched beside this one and not behind its last workgroup)
    for (int b = blockIdx.z; b < nframes; b += gridDim.z) {
    const uint8_t* base = ws + (size_t)b * P.frameBytes;
    const uint8_t* src = img + (size_t)b * iframe;
    float* T = (float*)(base + P.offT);
    float2* Cs = (float2*)(base + P.offCs);
    int* S = (int*)(base + P.offS); (void)S;
    Misc* misc = (Misc*)(base + P.offMisc);
    unsigned* comp = (unsigned*)(base + P.offComp);
    int* segCnt = (int*)(base + P.offSegCnt);
    const int lane = threadIdx.x;
    const int strip = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + threadIdx.y);      // eight output rows, ten blurred rows
    const int yBeg = strip * 8, yEnd = min(P.sh, yBeg + 8);
    if (yBeg >= P.sh) return;                               // the whole wave
    const int x4 = (blockIdx.x * 64 + lane) * 4;
    const bool inx = x4 < P.sw;                             // lanes beyond the row compute on the row's last group and store nothing
    const int x4c = min(x4, P.sw - 4);
    const int c0 = (x4c >> 2) * 5;                          // first source column of the group (the tables were checked against this pattern)
    unsigned cx[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) cx[j] = (unsigned)((const int2*)tx)[min(x4c + j, P.sw - 1)].y;
    const int2* tyv = (const int2*)ty;
    // the 16-byte window of a raw row: columns a .. a + 15, a = (c0 - 2) & ~3, needed bytes o0 .. o0 + 10 (o0 = (c0 - 2) & 3).  First lane of a row: a = -4, the window is
    // read from column 0 and shifted by one dword, its first dword built from the reflection; last lane (a = w - 8): the third dword is the reflection, the fourth is not needed.
    const int a = (c0 - 2) & ~3;
    const unsigned o0 = (unsigned)(c0 - 2) & 3u;
    const bool isLeft = a < 0, isRight = a + 8 >= P.w;
    const int aLoad = isLeft ? 0 : a;
    const unsigned t1 = (unsigned)tapsArr[1], t2 = (unsigned)tapsArr[2], t3 = (unsigned)tapsArr[3], t4 = (unsigned)tapsArr[4], t5 = (unsigned)tapsArr[5];
    const unsigned TA = t1 | (t2 << 8) | (t3 << 16) | (t4 << 24);      // taps of bytes i .. i + 3; byte i + 4 takes t5
    unsigned Hring[5][4];                                   // packed pairs: H[2j] | H[2j + 1] << 16, columns c0 .. c0 + 7 (the eighth is never used)
    // (columns up to w - 1 only: what lies behind a row is the next row, or -- for the last row of the last frame -- nothing.  The third and fourth dword of the lanes at
    // the row's end are re-read from the row's last dword: the last lane rebuilds its third from the reflection, nobody needs the others)
    const int off2 = min(aLoad + 8, P.w - 4), off3 = min(aLoad + 12, P.w - 4);
    auto raw_row = [&](int n) -> Raw16 {                    // raw row 10 strip - 2 + n of this frame
        const int rr = reflect_row(10 * strip - 2 + n, P.h, 2);
        const uint8_t* row = src + (size_t)rr * ipitch;
        Raw16 r;
        const uint2 lo = *(const uint2*)(row + aLoad);      // (4-byte aligned: the hardware takes it)
        r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = *(const unsigned*)(row + off2); r.v[3] = *(const unsigned*)(row + off3);
        return r;
    };
    auto hrow = [&](const Raw16& r, unsigned (&Hp)[4]) {
        unsigned L0 = r.v[0], L1 = r.v[1], L2 = r.v[2], L3 = r.v[3];
        if (isLeft) {       // loaded columns 0 .. 15 are L1 .. L3 (+ one more); L0 = columns -4 .. -1 = 4, 3, 2, 1
            L3 = L2; L2 = L1; L1 = L0;
            L0 = __builtin_amdgcn_perm(L2, L1, 0x01020304u);
        }
        if (isRight) L2 = __builtin_amdgcn_perm(L1, L0, 0x03040506u);      // columns w .. w + 3 = w - 2, w - 3, w - 4, w - 5 (L0:L1 = columns w - 8 .. w - 1)
        // e = bytes o0 .. o0 + 11 of the window = raw columns c0 - 2 .. c0 + 9
        const unsigned e0 = __builtin_amdgcn_alignbyte(L1, L0, o0), e1 = __builtin_amdgcn_alignbyte(L2, L1, o0), e2 = __builtin_amdgcn_alignbyte(L3, L2, o0);
        unsigned H[8];
#pragma unroll
        for (int i = 0; i < 7; ++i) {                       // H[i] = sum over l of t[l + 1] * e.byte[i + l], l = 0 .. 4
            const int q = i >> 2, sft = i & 3;
            const unsigned lo = q == 0 ? (sft ? __builtin_amdgcn_alignbyte(e1, e0, (unsigned)sft) : e0) : (sft ? __builtin_amdgcn_alignbyte(e2, e1, (unsigned)sft) : e1);
            const int i4 = i + 4, q4 = i4 >> 2, s4 = i4 & 3;
            const unsigned b4 = ((q4 == 1 ? e1 : e2) >> (8 * s4)) & 255u;
            H[i] = __builtin_amdgcn_udot4(lo, TA, __umul24(b4, t5), false);
        }
        H[7] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) Hp[j] = H[2 * j] | (H[2 * j + 1] << 16);
    };
    auto blur_row = [&](int top, int (&B)[7]) {             // the blurred row whose five H rows start at ring slot `top`
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            unsigned acc = 32768u;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const unsigned hp = Hring[(top + k) % 5][i >> 1];
                const unsigned tk = k == 0 ? t1 : k == 1 ? t2 : k == 2 ? t3 : k == 3 ? t4 : t5;
                acc += __umul24((i & 1) ? (hp >> 16) : (hp & 0xFFFFu), tk);
            }
            B[i] = (int)min(acc >> 16, 255u);
        }
    };
    auto scale_row = [&](int yy, const int (&B0)[7], const int (&B1)[7], int (&p)[5]) {      // scaled row yy from its two blurred rows
        const unsigned cy = (unsigned)tyv[yy].y;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int i = j < 4 ? j : 5;                    // source columns c0 + {0, 1, 2, 3, 5}
            p[j] = scaled_px<LIN>((unsigned)B0[i] | ((unsigned)B0[i + 1] << 8), (unsigned)B1[i] | ((unsigned)B1[i + 1] << 8), cx[j], cy);
        }
        if (x4c + 4 > P.sw - 1) p[4] = p[3];                // the row's last group: column sw is clamped to sw - 1 (the table lookup above used its entry for both)
    };
    int smax = 0;
    auto process = [&](int y, const int (&p0)[5], const int (&p1)[5]) {      // p0 / p1: the scaled rows y and y + 1 (k_lsd_grad's, unchanged)
        float ang[4]; float2 cs[4]; int sv[4], gidx[4];
        unsigned flags = 0;
        const bool lastRow = y >= P.sh - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ang[j] = NOTDEF_F; cs[j] = make_float2(0.f, 0.f); sv[j] = -1;
            const int DA = p1[j + 1] - p0[j], BC = p0[j + 1] - p1[j];
            const int gx = DA + BC, gy = DA - BC, s = gx * gx + gy * gy;
            gidx[j] = (gy + 510) * GT + (gx + 510);
            if (inx && !lastRow && x4 + j < P.sw - 1 && s >= P.sMin) { flags |= 1u << j; sv[j] = s; }
            else if (DENSE) sv[j] = s;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (flags & (1u << j)) { const float4 rec = gtab[gidx[j]]; ang[j] = rec.x; cs[j] = make_float2(rec.y, rec.z); }      // (the four gathers are in flight together; all of them cost 0.7 of the kernel's 11 ms: call P)
        const size_t i = (size_t)y * P.sw + x4;
        if (inx) *(float4*)(T + i) = make_float4(ang[0], ang[1], ang[2], ang[3]);      // (sw % 4 == 0 on this path)
        if (DENSE && inx) *(int4*)(S + i) = make_int4(sv[0], sv[1], sv[2], sv[3]);
        if (flags) {
            if (!DENSE) *(int4*)(S + i) = make_int4(sv[0], sv[1], sv[2], sv[3]);
            float4* c = (float4*)(Cs + i);
            c[0] = make_float4(cs[0].x, cs[0].y, cs[1].x, cs[1].y);
            c[1] = make_float4(cs[2].x, cs[2].y, cs[3].x, cs[3].y);
        }
        const int seg = y * P.nXB + blockIdx.x;
        const int cnt = __popc(flags), incl = wave_incl_scan(cnt);
        unsigned* dst = comp + ((size_t)seg << 8);
        int pos = incl - cnt;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (flags & (1u << j)) dst[pos++] = ((unsigned)sv[j] << 8) | (unsigned)(4 * lane + j);
        if (lane == 63) segCnt[seg] = incl;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (!DENSE || (flags & (1u << j))) smax = max(smax, sv[j]);
    };
    // Straight-line schedule over the sixteen raw rows n = 0 .. 15 (AH loads ahead): blurred row bi = n - 4 is complete with raw row n; scaled row yBeg + i needs the
    // blurred rows {0,1}, {1,2}, {2,3}, {3,4}, {5,6}, {6,7}, {7,8}, {8,9}, {10,11} for i = 0 .. 8; output row yBeg + i needs the scaled rows i and i + 1.
    int Bprev[7], Bcur[7], pc[5], pn[5];
    constexpr int AH = 2;                                   // raw rows requested ahead of the one being consumed (3, 4 and 6 measured: 10.9 against 11.1 ms, call Q -- the loads are not what it waits for)
    Raw16 ring[AH];
#pragma unroll
    for (int k = 0; k < AH; ++k) ring[k] = raw_row(k);
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        const Raw16 cur = ring[n % AH];
        if (n + AH < 16) ring[n % AH] = raw_row(n + AH);
        hrow(cur, Hring[n % 5]);
        if (n < 4) continue;
        const int bi = n - 4;                               // blurred row 10 strip + bi; its H rows are ring slots (n - 4) % 5 .. n % 5
        blur_row((n + 1) % 5, Bcur);
        const int sIdx = bi == 1 ? 0 : bi == 2 ? 1 : bi == 3 ? 2 : bi == 4 ? 3 : bi == 6 ? 4 : bi == 7 ? 5 : bi == 8 ? 6 : bi == 9 ? 7 : bi == 11 ? 8 : -1;      // the scaled row this blurred row completes
        if (sIdx >= 0) {
            const int yy = min(yBeg + sIdx, P.sh - 1);
            if (sIdx == 0) scale_row(yy, Bprev, Bcur, pc);
            else {
                if (yBeg + sIdx > P.sh - 1) {               // below the image: the gradient's "next row" is the last row itself (k_lsd_grad: min(y + 1, sh - 1))
#pragma unroll
                    for (int j = 0; j < 5; ++j) pn[j] = pc[j];
                } else scale_row(yy, Bprev, Bcur, pn);
                const int y = yBeg + sIdx - 1;
                if (y < yEnd) process(y, pc, pn);
#pragma unroll
                for (int j = 0; j < 5; ++j) pc[j] = pn[j];
            }
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) Bprev[i] = Bcur[i];
    }
    smax = wave_max(smax);
    if (lane == 0 && smax > 0) atomicMax(&misc->maxS, smax);
    }
}

__device__ __forceinline__ int lsd_bin(int s, double binCoef) {
    int i = (int)(sqrt((double)s / 4.0) * binCoef);
    return min(max(i, 0), N_BINS - 1);
}
__device__ __forceinline__ double lsd_bin_coef(int maxS) {
    return maxS > 0 ? (double)(N_BINS - 1) / sqrt((double)maxS / 4.0) : 0.0;
}
// lsd_bin for a whole wave without the fp64 square root where fp32 decides: y = sqrt32(s) * (float)(binCoef / 2) is within 2.5e-4 of sqrt(s / 4) * binCoef (s < 2^24 is exact
// in fp32; v_sqrt_f32 1 ulp, the coefficient's conversion and the product half an ulp each: 2.4e-7 relative of at most 1 023), the fp64 expression within 1e-9 of it -- so
// floor(y) IS the bin unless y lies within 1e-3 of an integer (or at the clamp), and only then (about one slot in 500, one group of 64 in eight) the wave takes the exact
// expression for the lanes concerned.  `have`: the lane holds an entry.  ~10 instead of ~30 vector instructions per entry and pass (k_lsd_hist_sort bins every entry twice).
__device__ __forceinline__ int lsd_bin_wave(int s, bool have, double binCoef, float coef32) {
    const float y = __builtin_amdgcn_sqrtf((float)s) * coef32;
    const float fl = floorf(y), fr = y - fl;
    int i = (int)fl;
    const bool amb = have && !(fr >= 1e-3f && fr <= 0.999f && y < (float)(N_BINS - 2));
    if (__builtin_amdgcn_ballot_w64(amb)) { if (amb) i = lsd_bin(s, binCoef); }
    return i;
}

#ifdef SSLAM_TESTING
// sslam_selftest_lsd_bin: lsd_bin_wave against lsd_bin for EVERY s in [0, maxS] (what a frame whose largest |g|^2 is maxS can hold)
__global__ __launch_bounds__(256) void k_selftest_lsd_bin(int maxS, unsigned long long* __restrict__ bad) {
    const double bc = lsd_bin_coef(maxS); const float bc32 = (float)(bc * 0.5);
    unsigned long long nb = 0;
    for (long long s0 = (long long)blockIdx.x * 256; s0 <= maxS; s0 += (long long)gridDim.x * 256) {
        const long long s = s0 + threadIdx.x;
        const bool have = s <= maxS;
        const int a = lsd_bin_wave(have ? (int)s : 0, have, bc, bc32);
        if (have && a != lsd_bin((int)s, bc)) ++nb;
    }
    if (nb) atomicAdd(bad, nb);
}
#endif

// stable counting sort of the DEFINED pixels by descending bin, raster order inside a bin (D2): per-tile histograms -> scan -> stable
// scatter.  A tile = P.tileRows whole rows = the consecutive segments [seg0, seg0 + nseg) of k_lsd_grad's lists; both kernels walk the
// tile's entries as one flat sequence (exclusive prefix sums of the segment counts in LDS; a lane's segment cursor only moves forward).
constexpr int MAX_TSEG = 256;           // segments per tile (lines_build_plan)
__device__ __forceinline__ int tile_segments(const int* __restrict__ segCnt, int seg0, int nseg, int* __restrict__ pref, int lane) {
    int carry = 0;
    for (int s0 = 0; s0 < nseg; s0 += 64) {
        const int s = s0 + lane;
        const int c = s < nseg ? segCnt[seg0 + s] : 0;
        const int incl = wave_incl_scan(c);
        if (s < nseg) pref[s] = carry + incl - c;
        carry += __builtin_amdgcn_readlane(incl, 63);
    }
    if (lane == 0) pref[nseg] = carry;
    return carry;
}

// Which (tile, frame) a workgroup of the two tile kernels takes: workgroups are dealt round-robin over the eight XCDs, and every tile of a
// frame goes to the SAME one -- the scatter's 4-byte stores of the tiles of a frame interleave in the frame's `order` array, and lines
// that several L2s hold partially leave as partial writes.  Grid = 8 * ceil(nframes / 8) * nTiles workgroups (sort_grid).
__device__ __forceinline__ bool sort_tile(int nTiles, int nframes, int& tile, int& b) {
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3, q = k / nTiles;
    tile = k - q * nTiles; b = q * 8 + xcd;
    return b < nframes;
}
static inline unsigned sort_grid(int nTiles, int nframes) { return 8u * (unsigned)((nframes + 7) / 8) * (unsigned)nTiles; }

// histogram pass: bins the entries once (the fp64 square root) and leaves bin << 8 | column in place of |g|^2 << 8 | column for the scatter pass
__global__ __launch_bounds__(64) void k_lsd_hist(uint8_t* __restrict__ ws, LsdPlan P, int nframes) {
    __shared__ int hist[N_BINS];
    __shared__ int pref[MAX_TSEG + 1];
    int tile, b; const int lane = threadIdx.x;
    if (!sort_tile(P.nTiles, nframes, tile, b)) return;
    uint8_t* base = ws + (size_t)b * P.frameBytes;
    const Misc* misc = (const Misc*)(base + P.offMisc);
    unsigned* comp = (unsigned*)(base + P.offComp);
    int* th = (int*)(base + P.offTileHist) + (size_t)tile * N_BINS;
    for (int i = lane; i < N_BINS; i += 64) hist[i] = 0;
    const int r0 = tile * P.tileRows, r1 = min(P.sh, r0 + P.tileRows), seg0 = r0 * P.nXB, nseg = (r1 - r0) * P.nXB;
    const int total = tile_segments((const int*)(base + P.offSegCnt), seg0, nseg, pref, lane);
    __syncthreads();
    const double bc = lsd_bin_coef(misc->maxS);
    int sgi = 0;
    for (int e0 = 0; e0 < total; e0 += 256) {            // four loads in flight per lane
        unsigned* p[4]; unsigned ent[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = e0 + k * 64 + lane;
            p[k] = nullptr; ent[k] = 0;
            if (e < total) {
                while (e >= pref[sgi + 1]) ++sgi;
                p[k] = comp + ((size_t)(seg0 + sgi) << 8) + (e - pref[sgi]);
                ent[k] = *p[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) if (p[k]) {
            const int bin = lsd_bin((int)(ent[k] >> 8), bc);
            atomicAdd(&hist[bin], 1);
            *p[k] = ((unsigned)bin << 8) | (ent[k] & 255u);
        }
    }
    __syncthreads();
    for (int i = lane; i < N_BINS; i += 64) th[i] = hist[i];
}

__global__ __launch_bounds__(1024) void k_lsd_scan(uint8_t* __restrict__ ws, LsdPlan P) {
    __shared__ int part[1024];
    const int b = blockIdx.x, t = threadIdx.x;
    uint8_t* base = ws + (size_t)b * P.frameBytes;
    int* th = (int*)(base + P.offTileHist);
    Misc* misc = (Misc*)(base + P.offMisc);
    const int bin = N_BINS - 1 - t;           // thread t owns the t-th bin in descending order
    int tot = 0;
    for (int k = 0; k < P.nTiles; ++k) tot += th[(size_t)k * N_BINS + bin];
    part[t] = tot;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {      // inclusive Hillis-Steele scan
        int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int basePos = part[t] - tot;
    for (int k = 0; k < P.nTiles; ++k) {
        int c = th[(size_t)k * N_BINS + bin];
        th[(size_t)k * N_BINS + bin] = basePos;
        basePos += c;
    }
    if (t == 1023) misc->nDefined = part[1023];
}

// which lanes hold the same 10-bit key: one ballot per key bit, every lane keeps the lanes that agree with it on that bit (`own` is 0 or
// -1: m ^ own is m or ~m) -- 60 lane-parallel instructions per 64 entries, where a leader loop runs once per DISTINCT key (round 3: ~10
// dependent instructions for each of up to 64 bins)
__device__ __forceinline__ unsigned long long same_key10(int key, unsigned long long valid) {
    unsigned lo = (unsigned)valid, hi = (unsigned)(valid >> 32);
#pragma unroll
    for (int bit = 0; bit < 10; ++bit) {
        const int own = -((key >> bit) & 1);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(own != 0);
        lo &= ~((unsigned)m ^ (unsigned)own); hi &= ~((unsigned)(m >> 32) ^ (unsigned)own);
    }
    return ((unsigned long long)hi << 32) | lo;
}

// One wave per tile walks its entries in raster order, 64 at a time (four such groups are loaded ahead).  Inside a group the rank of an
// entry among the lanes of the same bin comes from same_key10; the wave's LDS accesses execute in program order, so the cursor read /
// write-back needs no barrier.
__global__ __launch_bounds__(64) void k_lsd_scatter(uint8_t* __restrict__ ws, LsdPlan P, int nframes) {
    __shared__ int cursor[N_BINS];
    __shared__ int pref[MAX_TSEG + 1];
    __shared__ unsigned segXY[MAX_TSEG];      // first column | row << 16 of each segment of the tile
    int tile, b; const int lane = threadIdx.x;
    if (!sort_tile(P.nTiles, nframes, tile, b)) return;
    uint8_t* base = ws + (size_t)b * P.frameBytes;
    const unsigned* comp = (const unsigned*)(base + P.offComp);
    const int* th = (const int*)(base + P.offTileHist) + (size_t)tile * N_BINS;
    unsigned* order = (unsigned*)(base + P.offOrder);
    for (int i = lane; i < N_BINS; i += 64) cursor[i] = th[i];
    const int r0 = tile * P.tileRows, r1 = min(P.sh, r0 + P.tileRows), seg0 = r0 * P.nXB, nseg = (r1 - r0) * P.nXB;
    for (int s = lane; s < nseg; s += 64) { const int r = s / P.nXB; segXY[s] = (unsigned)((s - r * P.nXB) << 8) | ((unsigned)(r0 + r) << 16); }
    const int total = tile_segments((const int*)(base + P.offSegCnt), seg0, nseg, pref, lane);
    __syncthreads();
    int sgi = 0;
    for (int e0 = 0; e0 < total; e0 += 256) {
        unsigned ent[4], xy[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = e0 + k * 64 + lane;
            ent[k] = 0; xy[k] = 0;
            if (e < total) {
                while (e >= pref[sgi + 1]) ++sgi;
                ent[k] = comp[((size_t)(seg0 + sgi) << 8) + (e - pref[sgi])];
                xy[k] = segXY[sgi];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (e0 + k * 64 >= total) break;             // wave-uniform
            const bool def = e0 + k * 64 + lane < total;
            const int bin = (int)(ent[k] >> 8);
            const unsigned long long peers = same_key10(bin, __ballot(def));
            if (def) {
                const int rank = mbcnt(peers), pos = cursor[bin] + rank;
                order[pos] = xy[k] + (ent[k] & 255u);      // x | y << 16 (the packing of the region lists)
                if (rank == __popcll(peers) - 1) cursor[bin] = pos + 1;
            }
        }
    }
}

// ------------------------------------------------------------------ round 6: the counting sort with tile-sorted runs
// k_lsd_scatter issues one 4-byte store per defined pixel to wherever that pixel's bin lives in the frame's seed list: 369 M scattered stores per step of 12 288 frames, and
// the kernel runs at the rate the chip answers random sectors (tools/gather_probe), 6.6 x write amplification included.  But a tile's ~1 200 entries fall into only ~300 of the
// 1 024 bins, most of them in the few populous low-gradient bins: sorted by bin INSIDE the tile (stable), same-bin entries are neighbours, their positions in the frame's list are
// consecutive, and neighbouring lanes store to neighbouring dwords -- 3.5 x fewer requests on the bench's frames.  The tile-local sort is a second pass of the histogram kernel
// (it has the tile's histogram in LDS anyway): its scattered stores stay inside the tile's own 5 KB window, which the L2 absorbs.  A sorted entry is bin << 22 | y << 11 | x
// (scaled images up to 2048 x 2048; larger ones keep k_lsd_hist + k_lsd_scatter).
constexpr int SORT_XY_BITS = 11;
constexpr int HSD = 12;      // groups of 64 entries loaded ahead per lane in both passes (2 / 4 / 8 / 12 / 16: 3.9 / 3.5 / 3.2 / 3.1 / 3.1 ms per 12 288 frames, GPU call AL: the wave waits for its loads)
__global__ __launch_bounds__(64) void k_lsd_hist_sort(uint8_t* __restrict__ ws, LsdPlan P, int nframes) {
    __shared__ int hist[N_BINS];              // the tile's histogram, then its exclusive positions (descending bins) as the local cursors
    __shared__ int pref[MAX_TSEG + 1];
    __shared__ unsigned segXY[MAX_TSEG];      // first column | row << SORT_XY_BITS of each segment of the tile
    int tile, b; const int lane = threadIdx.x;
    if (!sort_tile(P.nTiles, nframes, tile, b)) return;
    uint8_t* base = ws + (size_t)b * P.frameBytes;
    const Misc* misc = (const Misc*)(base + P.offMisc);
    const unsigned* comp = (const unsigned*)(base + P.offComp);
    unsigned* sorted = (unsigned*)(base + P.offSorted);
    int* th = (int*)(base + P.offTileHist) + (size_t)tile * N_BINS;
    for (int i = lane; i < N_BINS; i += 64) hist[i] = 0;
    const int r0 = tile * P.tileRows, r1 = min(P.sh, r0 + P.tileRows), seg0 = r0 * P.nXB, nseg = (r1 - r0) * P.nXB;
    for (int sgm = lane; sgm < nseg; sgm += 64) { const int r = sgm / P.nXB; segXY[sgm] = (unsigned)((sgm - r * P.nXB) << 8) | ((unsigned)(r0 + r) << SORT_XY_BITS); }
    const int total = tile_segments((const int*)(base + P.offSegCnt), seg0, nseg, pref, lane);
    __syncthreads();
    const double bc = lsd_bin_coef(misc->maxS);
    const float bc32 = (float)(bc * 0.5);
    int sgi = 0;
    for (int e0 = 0; e0 < total; e0 += 64 * HSD) {            // pass 1: bin every entry and count (the entries stay as they are: pass 2 bins them again instead of reading back what this pass would have stored)
        unsigned ent[HSD]; bool have[HSD];
#pragma unroll
        for (int k = 0; k < HSD; ++k) {
            const int e = e0 + k * 64 + lane;
            have[k] = e < total; ent[k] = 0;
            if (have[k]) {
                while (e >= pref[sgi + 1]) ++sgi;
                ent[k] = comp[((size_t)(seg0 + sgi) << 8) + (e - pref[sgi])];
            }
        }
#pragma unroll
        for (int k = 0; k < HSD; ++k) {
            if (e0 + k * 64 >= total) break;             // wave-uniform
            const int bin = lsd_bin_wave((int)(ent[k] >> 8), have[k], bc, bc32);
            if (have[k]) atomicAdd(&hist[bin], 1);
        }
    }
    __syncthreads();
    // the tile's histogram goes to the scan kernel; locally it becomes exclusive positions in DESCENDING bin order (the frame list's order): lane l owns bins 1023 - 16 l .. 1008 - 16 l
    {
        int c[16], sum = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) { c[j] = hist[N_BINS - 1 - (16 * lane + j)]; sum += c[j]; }
        for (int i = lane; i < N_BINS; i += 64) th[i] = hist[i];
        int run = wave_incl_scan(sum) - sum;
#pragma unroll
        for (int j = 0; j < 16; ++j) { hist[N_BINS - 1 - (16 * lane + j)] = run; run += c[j]; }
    }
    __syncthreads();
    unsigned* out = sorted + ((size_t)seg0 << 8);        // the tile's block: as many slots as its segments have (>= total)
    sgi = 0;
    for (int e0 = 0; e0 < total; e0 += 64 * HSD) {            // pass 2: the tile's entries in raster order to their places in the tile-sorted block
        unsigned ent[HSD], xy[HSD];
#pragma unroll
        for (int k = 0; k < HSD; ++k) {
            const int e = e0 + k * 64 + lane;
            ent[k] = 0; xy[k] = 0;
            if (e < total) {
                while (e >= pref[sgi + 1]) ++sgi;
                ent[k] = comp[((size_t)(seg0 + sgi) << 8) + (e - pref[sgi])];
                xy[k] = segXY[sgi];
            }
        }
#pragma unroll
        for (int k = 0; k < HSD; ++k) {
            if (e0 + k * 64 >= total) break;             // wave-uniform
            const bool def = e0 + k * 64 + lane < total;
            const int bin = lsd_bin_wave((int)(ent[k] >> 8), def, bc, bc32);
            const unsigned long long peers = same_key10(bin, __ballot(def));
            if (def) {
                const int rank = mbcnt(peers), pos = hist[bin] + rank;
                out[pos] = ((unsigned)bin << (2 * SORT_XY_BITS)) | (xy[k] + (ent[k] & 255u));
                if (rank == __popcll(peers) - 1) hist[bin] = pos + 1;
            }
        }
    }
}

// The scatter over tile-sorted runs: lanes of one run hold one bin, their rank is their distance from the run's first lane, and they store to consecutive dwords.
constexpr int SRD = 4;       // (8 / 16 groups ahead: no change, 1.9 ms)
__global__ __launch_bounds__(64) void k_lsd_scatter_runs(uint8_t* __restrict__ ws, LsdPlan P, int nframes) {
    __shared__ int cursor[N_BINS];
    int tile, b; const int lane = threadIdx.x;
    if (!sort_tile(P.nTiles, nframes, tile, b)) return;
    uint8_t* base = ws + (size_t)b * P.frameBytes;
    const int* th = (const int*)(base + P.offTileHist) + (size_t)tile * N_BINS;
    unsigned* order = (unsigned*)(base + P.offOrder);
    for (int i = lane; i < N_BINS; i += 64) cursor[i] = th[i];
    const int r0 = tile * P.tileRows, r1 = min(P.sh, r0 + P.tileRows), seg0 = r0 * P.nXB, nseg = (r1 - r0) * P.nXB;
    const int* segCnt = (const int*)(base + P.offSegCnt) + seg0;
    int total = 0;
    for (int sgm = lane; sgm < nseg; sgm += 64) total += segCnt[sgm];
    total = wave_sum(total);
    const unsigned* in = (const unsigned*)(base + P.offSorted) + ((size_t)seg0 << 8);
    __syncthreads();
    constexpr unsigned XYM = (1u << SORT_XY_BITS) - 1;
    for (int e0 = 0; e0 < total; e0 += 64 * SRD) {
        unsigned v[SRD];
#pragma unroll
        for (int k = 0; k < SRD; ++k) { const int e = e0 + k * 64 + lane; v[k] = e < total ? in[e] : 0xFFFFFFFFu; }
#pragma unroll
        for (int k = 0; k < SRD; ++k) {
            if (e0 + k * 64 >= total) break;             // wave-uniform
            const bool def = e0 + k * 64 + lane < total;
            const int bin = (int)(v[k] >> (2 * SORT_XY_BITS));
            const int prevBin = __builtin_amdgcn_update_dpp(-1, bin, 0x138, 0xF, 0xF, false);      // wave_shr:1 -- lane 0 keeps -1: it always starts a run
            const int nextBin = __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, bin);
            const unsigned long long heads = __ballot(def && (lane == 0 || bin != prevBin));
            const unsigned long long below = heads & (~0ull >> (63 - lane));                      // run starts at or below this lane
            const int start = 63 - __clzll((long long)below);
            if (def) {
                const int rank = lane - start, pos = cursor[bin] + rank;
                order[pos] = (v[k] & XYM) | (((v[k] >> SORT_XY_BITS) & XYM) << 16);                 // x | y << 16 (the packing of the region lists)
                const bool last = lane == 63 || nextBin != bin || e0 + k * 64 + lane + 1 >= total;
                if (last) cursor[bin] = pos + 1;
            }
        }
    }
}
