// Line front-end for MI355X (gfx950): LSD detector (REFINE_ADV) + KeyLine fill +
// top-N by response + LBD band descriptor + line equations — the device side of
// LineSegment::ExtractLineSegment (reference src/ExtractLineSegment.cpp:18-69), whose
// arithmetic lives in un-vendored OpenCV 3.4 (imgproc/lsd.cpp, contrib
// line_descriptor LSDDetector.cpp / binary_descriptor.cpp).
//
// Design:
//   * data-parallel stages (blur, 0.8x rescale, gradient/level-line angle, 1024-bin
//     stable counting sort of the seeds, Sobel, LBD) are ordinary batched launches;
//   * region growing is sequential by definition (seed order + a global `used` map
//     + a region angle that changes with every accepted pixel).  It runs as one
//     persistent 256-thread workgroup per frame: wave 0 replays the exact sequential
//     order, the `used` bitmap lives in LDS, and the whole workgroup shares the
//     data-parallel parts (seed scan, un-marking, NFA rectangle counting).  Frames of
//     a batch run concurrently, one workgroup each;
//   * all order-dependent fp64 sums are accumulated in the reference's order so the
//     segments match the CPU oracle bit-for-bit.
#include "common.h"
#include <cmath>
#include <cstdlib>
#include <cfloat>
#include <algorithm>

using namespace sslam;

namespace {

constexpr double kPI = 3.14159265358979323846;
constexpr double DEG2RAD = kPI / 180;
constexpr double M_3_2_PI_ = (3 * kPI) / 2, M_2PI_ = 2 * kPI;
constexpr float NOTDEF_F = -1024.0f;
constexpr float USED_F = -2048.0f;       // written over pix[].x while a pixel belongs to a region (the `used` map)
constexpr int N_BINS = 1024;
constexpr int TILE_PX = 8192;           // raster tile of the counting sort
constexpr int MAX_SEG = 8192;           // segments per frame (LSD output capacity)
constexpr int NUM_BANDS = 9, BAND_W = 7, LSP_H = 63;

struct LsdPlan {
    int w, h;                 // source image
    int sw, sh, spitch;       // scaled image (0.8x)
    int npx;                  // sw*sh
    int nTiles;
    size_t frameBytes;        // per-frame workspace
    size_t offBlur, offAng, offS, offPix, offCand, offFlag, offNfa, offOrder, offTileHist, offReg, offSeg, offMisc, offDxy, offKl, offSortIdx;
    int blurTaps[7];          // sigma 0.75, 7 taps (q8)
    int blur5Taps[5];         // sigma 1, 5 taps (q8)
    int tabX, tabY;           // offsets into the resize table (int: ofs, c1)
    double rho, prec, p, logNT;
    int minRegSize;
};

struct Misc {                 // per-frame scalars
    int maxS;                 // max gx^2+gy^2 over defined pixels
    int nDefined;
    int nSeg;
    int nCand;                // rectangles handed from the sequential core to the NFA stage
    int nKl;
    int overflow;
    long long cyc[8];         // master-wave cycle breakdown (debug): grow, rect, refine, nfa count, nfa math, seed scan
};

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
#pragma unroll
    for (int r = 0; r < STRIP + 2 * R; ++r) {
        if (r >= 2 * R && y0 + r - 2 * R >= h) break;
        unsigned d0, d1, d2;
        load_row12(s, spitch, reflect_row(y0 - R + r, h, R), x4, w, fast, d0, d1, d2);
#pragma unroll
        for (int j = 0; j < 4; ++j) win[r % 7][j] = hdot(d0, d1, d2, j + 1, T0, T1);      // columns x4+j-3 .. x4+j+3
        if (r >= 2 * R) {
            const int y = y0 + r - 2 * R;
            unsigned o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned acc = 0;
#pragma unroll
                for (int k = 0; k < 7; ++k) acc += win[(r - 2 * R + k) % 7][j] * taps[k];
                o[j] = (acc + 32768u) >> 16;
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

// One thread = four horizontally adjacent pixels: the 2x5 source bytes come from two dword + two byte loads, the four
// table gathers are in flight together, and angle / key / record leave as 16-byte stores when the row length allows.
// S[i] = |g|^2 for DEFINED pixels and -1 otherwise, so the counting sort reads one array.
// The 0.8x INTER_LINEAR_EXACT image (D7) is never stored: this kernel is bound by its 24 B/pixel of writes, so the 2 x 5
// scaled pixels a thread needs are recomputed here from the blurred source (four source rows as three aligned dwords each).
__device__ __forceinline__ int scaled_px(unsigned e0, unsigned e1, unsigned cx, unsigned cy) {      // e = {p0, p1} bytes of the two source rows
    const unsigned r0 = (e0 & 255u) * (256u - cx) + ((e0 >> 8) & 255u) * cx, r1 = (e1 & 255u) * (256u - cx) + ((e1 >> 8) & 255u) * cx;
    return (int)((r0 * (256u - cy) + r1 * cy + 32768u) >> 16);
}
__global__ __launch_bounds__(256) void k_lsd_grad(const uint8_t* __restrict__ ws, LsdPlan P, const float4* __restrict__ gtab, size_t bpitch,
                                                  const int* __restrict__ tx, const int* __restrict__ ty) {
    const int b = blockIdx.z;
    const uint8_t* base = ws + (size_t)b * P.frameBytes;
    const uint8_t* src = base + P.offBlur;
    float* ang = (float*)(base + P.offAng);
    int* S = (int*)(base + P.offS);
    float4* pix = (float4*)(base + P.offPix);
    Misc* misc = (Misc*)(base + P.offMisc);
    const int y = blockIdx.y * 4 + threadIdx.y, x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    int smax = 0;
    if (y < P.sh && x4 < P.sw) {
        const bool lastRow = y >= P.sh - 1;
        const int2 ty0 = ((const int2*)ty)[y], ty1 = ((const int2*)ty)[min(y + 1, P.sh - 1)];
        int2 txv[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) txv[j] = ((const int2*)tx)[min(x4 + j, P.sw - 1)];
        const uint8_t* rw[4] = {src + (size_t)ty0.x * bpitch, src + (size_t)min(ty0.x + 1, P.h - 1) * bpitch,
                                src + (size_t)ty1.x * bpitch, src + (size_t)min(ty1.x + 1, P.h - 1) * bpitch};
        int p0[5], p1[5];
        const int a = txv[0].x & ~3;
        if (txv[4].x - a <= 10) {                       // bpitch % 64 == 0 and another buffer follows the last row: whole dwords are readable
            unsigned d[4][3];
#pragma unroll
            for (int r = 0; r < 4; ++r) { const unsigned* q = (const unsigned*)(rw[r] + a); d[r][0] = q[0]; d[r][1] = q[1]; d[r][2] = q[2]; }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int o = txv[j].x - a;                                    // second tap has weight 0 at the last column
                p0[j] = scaled_px(pick2(d[0][0], d[0][1], d[0][2], o), pick2(d[1][0], d[1][1], d[1][2], o), (unsigned)txv[j].y, (unsigned)ty0.y);
                p1[j] = scaled_px(pick2(d[2][0], d[2][1], d[2][2], o), pick2(d[3][0], d[3][1], d[3][2], o), (unsigned)txv[j].y, (unsigned)ty1.y);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int sx = txv[j].x, sx1 = min(sx + 1, P.w - 1);
                p0[j] = scaled_px(rw[0][sx] | ((unsigned)rw[0][sx1] << 8), rw[1][sx] | ((unsigned)rw[1][sx1] << 8), (unsigned)txv[j].y, (unsigned)ty0.y);
                p1[j] = scaled_px(rw[2][sx] | ((unsigned)rw[2][sx1] << 8), rw[3][sx] | ((unsigned)rw[3][sx1] << 8), (unsigned)txv[j].y, (unsigned)ty1.y);
            }
        }
        float4 rec[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int DA = p1[j + 1] - p0[j], BC = p0[j + 1] - p1[j];
            const int gx = DA + BC, gy = DA - BC;
            const bool in = !lastRow && x4 + j < P.sw - 1;
            rec[j] = in ? gtab[(gy + 510) * GT + (gx + 510)] : make_float4(NOTDEF_F, 0.f, 0.f, 0.f);
        }
        int sv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool def = rec[j].x != NOTDEF_F;
            sv[j] = def ? __float_as_int(rec[j].w) : -1;
            smax = max(smax, sv[j]);
        }
        const size_t i = (size_t)y * P.sw + x4;
        if ((P.sw & 3) == 0) {
            *(float4*)(ang + i) = make_float4(rec[0].x, rec[1].x, rec[2].x, rec[3].x);
            *(int4*)(S + i) = make_int4(sv[0], sv[1], sv[2], sv[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (x4 + j < P.sw) { ang[i + j] = rec[j].x; S[i + j] = sv[j]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (x4 + j < P.sw) pix[i + j] = rec[j];      // per-pixel record for region growing: angle, cosf/sinf (D5), |g|^2
    }
    smax = wave_max(smax);
    if (threadIdx.x == 0 && smax > 0) atomicMax(&misc->maxS, smax);
}

__device__ __forceinline__ int lsd_bin(int s, double binCoef) {
    int i = (int)(sqrt((double)s / 4.0) * binCoef);
    return min(max(i, 0), N_BINS - 1);
}
__device__ __forceinline__ double lsd_bin_coef(int maxS) {
    return maxS > 0 ? (double)(N_BINS - 1) / sqrt((double)maxS / 4.0) : 0.0;
}

// stable counting sort of the DEFINED pixels by descending bin, raster order inside a bin (D2):
// per-tile histograms -> scan -> stable scatter.
__global__ __launch_bounds__(64) void k_lsd_hist(uint8_t* __restrict__ ws, LsdPlan P) {
    __shared__ int hist[N_BINS];
    const int tile = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    uint8_t* base = ws + (size_t)b * P.frameBytes;
    const int* S = (const int*)(base + P.offS);
    const Misc* misc = (const Misc*)(base + P.offMisc);
    int* th = (int*)(base + P.offTileHist) + (size_t)tile * N_BINS;
    for (int i = lane; i < N_BINS; i += 64) hist[i] = 0;
    __syncthreads();
    const double bc = lsd_bin_coef(misc->maxS);
    const int beg = tile * TILE_PX, end = min(beg + TILE_PX, P.npx);
    for (int i0 = beg; i0 < end; i0 += 512) {            // eight coalesced loads in flight per lane (order is irrelevant here)
        int v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int i = i0 + k * 64 + lane; v[k] = i < end ? S[i] : -1; }
#pragma unroll
        for (int k = 0; k < 8; ++k) if (v[k] >= 0) atomicAdd(&hist[lsd_bin(v[k], bc)], 1);
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

// One wave per tile walks its pixels in raster order, 64 at a time (four such groups are loaded ahead).  Inside a group
// the rank of a pixel among the lanes of the same bin comes from ballots; the wave's LDS accesses execute in program order,
// so the cursor read / write-back needs no barrier.
__global__ __launch_bounds__(64) void k_lsd_scatter(uint8_t* __restrict__ ws, LsdPlan P) {
    __shared__ int cursor[N_BINS];
    const int tile = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    uint8_t* base = ws + (size_t)b * P.frameBytes;
    const int* S = (const int*)(base + P.offS);
    const Misc* misc = (const Misc*)(base + P.offMisc);
    const int* th = (const int*)(base + P.offTileHist) + (size_t)tile * N_BINS;
    unsigned* order = (unsigned*)(base + P.offOrder);
    for (int i = lane; i < N_BINS; i += 64) cursor[i] = th[i];
    __syncthreads();
    const double bc = lsd_bin_coef(misc->maxS);
    const int beg = tile * TILE_PX, end = min(beg + TILE_PX, P.npx);
    for (int i0 = beg; i0 < end; i0 += 256) {
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = i0 + k * 64 + lane; v[k] = i < end ? S[i] : -1; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool def = v[k] >= 0;
            const int bin = def ? lsd_bin(v[k], bc) : -1;
            unsigned long long todo = __ballot(def);
            int rank = 0, total = 0;
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int bsel = __builtin_amdgcn_readlane(bin, leader);
                const unsigned long long m = __ballot(bin == bsel);
                if (bin == bsel) { rank = mbcnt(m); total = __popcll(m); }
                todo &= ~m;
            }
            if (def) {
                const int pos = cursor[bin] + rank;
                order[pos] = (unsigned)(i0 + k * 64 + lane);
                if (rank == total - 1) cursor[bin] = pos + 1;
            }
        }
    }
}

// ------------------------------------------------------------------ the sequential core
struct RectD { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; };

#ifndef SSLAM_LSD_QCAP
#define SSLAM_LSD_QCAP 1024
#endif
constexpr int QCAP = SSLAM_LSD_QCAP;  // region points kept in LDS; longer regions continue in global memory
constexpr int MAXC = 5;             // rectangle candidates evaluated per NFA job

__device__ __forceinline__ double angle_diff_signed(double a, double b) {
    double diff = a - b;
    while (diff <= -kPI) diff += M_2PI_;
    while (diff > kPI) diff -= M_2PI_;
    return diff;
}
__device__ __forceinline__ bool is_aligned_val(float aDeg, double theta, double prec) {
    // isAligned: |theta - a|, folded once around the circle (fabs == the reference's conditional negations; +-0 compare alike)
    double n_theta = fabs(theta - (double)aDeg * DEG2RAD);
    const double wrapped = fabs(n_theta - M_2PI_);
    n_theta = n_theta > M_3_2_PI_ ? wrapped : n_theta;
    return aDeg != NOTDEF_F && n_theta <= prec;
}
// Tables of nfa(): lgam[j] = log_gamma(j) for integer j >= 1 (every argument nfa() uses is an integer + 1), then
// plog[h] = {log(p), log(1-p), log10(p)} for p = 0.125 * 2^-h (every precision rect_improve can reach), then 1/j for
// exact_div().  They are evaluated on the HOST with the same libm calls, in the same order, as the reference's
// log_gamma_windschitl / log_gamma_lanczos (opencv lsd.cpp): when the binomial tail is ~1 the NFA is -logNT + O(1e-15),
// and rect_improve's strict `v > log_nfa` comparisons between such values depend on the last bit of every term.
static double host_log_gamma(double x) {
    if (x > 15) return 0.918938533204673 + (x - 0.5) * std::log(x) - x + 0.5 * x * std::log(x * std::sinh(1 / x) + 1 / (810.0 * std::pow(x, 6.0)));
    static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
    double a = (x + 0.5) * std::log(x + 5.5) - (x + 5.5), bq = 0;
    for (int n = 0; n < 7; ++n) { a -= std::log(x + (double)n); bq += q[n] * std::pow(x, (double)n); }
    return a + std::log(bq);
}
static int upload_nfa_tables(double* d_tab, int n, hipStream_t st) {
    std::vector<double> t(2 * (size_t)n + 48);
    for (int j = 0; j < n; ++j) t[j] = j >= 1 ? host_log_gamma((double)j) : 0.0;
    for (int j = 0; j < 16; ++j) { const double pp = std::ldexp(0.125, -j); t[n + 3 * j] = std::log(pp); t[n + 3 * j + 1] = std::log(1.0 - pp); t[n + 3 * j + 2] = std::log10(pp); }
    for (int j = 0; j < n; ++j) t[(size_t)n + 48 + j] = j >= 1 ? 1.0 / (double)j : 0.0;          // correctly rounded reciprocals for exact_div()
    SSLAM_HIP(hipMemcpyAsync(d_tab, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    return SSLAM_OK;
}

// a / b for small positive integers, bit-identical to the IEEE quotient: with y = RN(1/b) from the table, q0 = RN(a*y),
// the FMA residual r = a - b*q0 is exact and q0 + r*y rounds to RN(a/b) (Markstein's division theorem; checked against the
// hardware division by sslam_selftest_exact_div).
__device__ __forceinline__ double exact_div(double a, double b, double y) {
    const double q0 = a * y;
    const double r = fma(-b, q0, a);
    return fma(r, y, q0);
}
__global__ void k_selftest_div(const double* __restrict__ rcp, int n, unsigned long long seed, int iters, unsigned long long* __restrict__ bad) {
    unsigned long long x = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned long long nb = 0;
    for (int it = 0; it < iters; ++it) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const int b = 1 + (int)((x >> 8) % (unsigned long long)(n - 1));
        const int a = 1 + (int)((x >> 36) % (unsigned long long)(n - 1));
        const double q = exact_div((double)a, (double)b, rcp[b]), ref = (double)a / (double)b;
        nb += (__double_as_longlong(q) != __double_as_longlong(ref)) ? 1 : 0;
    }
    if (nb) atomicAdd(bad, nb);
}
// plog[h] = {log(p), log(1-p), log10(p)} for p = 0.125 * 2^-h: every precision rect_improve can reach
struct PLog { double lp, l1mp, l10p; };

// nfa()'s early-exit test `err < tolerance * |-log10(bin_tail) - logNT| * bin_tail` with
// err = term * ((1 - m^q) / (1 - m) - 1), decided from fp32 log2/exp2 estimates inside rigorous guard bands:
// returns 1 (test holds) / 0 (test fails) when the estimate cannot disagree with the fp64 expression, -1 when it might.
// Here 0 < m < 1/7 (bin_term < 1 and p <= 1/8), so B = m + m^2 + .. + m^(q-1) lies in [m, 1.17 m] for q >= 2 and is exactly 0
// for q == 1 (fl((1-m)/(1-m)) - 1).  v_log_f32 / v_exp_f32 are 1-ulp: the estimate of m^q is within 1e-5 relative for
// |q log2 m| <= 60 (and m^q < 1e-18 otherwise), the fp64 evaluation of B is within 6e-16 absolute, log10(bin_tail) from the
// split exponent + fp32 mantissa log is within 1e-7 absolute; the bands below are several times wider than that.
__device__ __forceinline__ int tail_test_cheap(double term, double m, int q, double bin_tail, double logNT) {
    if (!(m > 0.0 && m < 0.15)) return -1;
    double B = 0.0;
    if (q >= 2) {
        double mq = 0.0;
        if (m > 1e-30) {
            const float x = (float)q * __builtin_amdgcn_logf((float)m);
            if (x > -60.f) mq = (double)__builtin_amdgcn_exp2f(x);
        }
        B = (1.0 - mq) / (1.0 - m) - 1.0;
    }
    const double errHi = term * (B * (1.0 + 1e-5) + 4e-15), errLo = term * (B * (1.0 - 1e-5) - 4e-15);
    int e;
    const double f = frexp(bin_tail, &e);                                 // bin_tail = f * 2^e, f in [0.5, 1)
    const double l10 = ((double)e + (double)__builtin_amdgcn_logf((float)f)) * 0.30102999566398120;
    const double A = fabs(-l10 - logNT);
    const double rhsHi = 0.1 * (A + 2e-6) * bin_tail * (1.0 + 1e-14), rhsLo = 0.1 * fmax(A - 2e-6, 0.0) * bin_tail * (1.0 - 1e-14);
    if (errHi < rhsLo) return 1;
    if (errLo >= rhsHi) return 0;
    return -1;
}

// Self-test of tail_test_cheap (sslam_selftest_tail_test): random (term, m, q, bin_tail), half of them steered onto the
// decision boundary err ~ rhs, counted as disagreeing when the cheap verdict differs from the fp64 expression.
__global__ void k_selftest_tail(unsigned long long seed, int iters, double logNT, unsigned long long* __restrict__ out) {
    unsigned long long x = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned long long bad = 0, amb = 0;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (double)(x >> 11) * 0x1p-53; };
    for (int it = 0; it < iters; ++it) {
        const double m = exp2(-(2.81 + rnd() * rnd() * 38.0));
        const double u = rnd();
        const int q = 1 + (int)(u * u * u * 200000.0);
        const double term = exp2(-rnd() * 1000.0);
        double bin_tail = term * (1.0 + exp2(rnd() * 30.0 - 10.0));
        if (it & 1) {       // onto the boundary: err = 0.1 * A * bin_tail, +- up to 1e-3 relative
            const double err = term * ((1 - pow(m, (double)q)) / (1 - m) - 1);
            if (err > 0) {
                const double bt0 = err / (0.1 * 13.0), A0 = fabs(-log10(bt0) - logNT);
                if (A0 > 0) bin_tail = err / (0.1 * A0) * (1.0 + (rnd() - 0.5) * 2e-3 * rnd());
            }
        }
        const int dec = tail_test_cheap(term, m, q, bin_tail, logNT);
        const double err = term * ((1 - pow(m, (double)q)) / (1 - m) - 1);
        const bool ref = err < 0.1 * fabs(-log10(bin_tail) - logNT) * bin_tail;
        if (dec < 0) ++amb; else if ((dec > 0) != ref) ++bad;
    }
    if (bad) atomicAdd(out, bad);
    if (amb) atomicAdd(out + 1, amb);
}

// FETCH_SIZE calibration probes (tools/fetch_probe.py under rocprofv3 --pmc FETCH_SIZE): a known number of bytes read
// in the two access patterns this library uses most, 16 B/lane coalesced streams and scattered 16-B gathers.
__global__ void k_probe_stream16(const float4* __restrict__ buf, size_t nElem, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nElem; i += (size_t)gridDim.x * blockDim.x) acc += buf[i].x;
    if (acc == 12345.678f) *sink = acc;
}
__global__ void k_probe_gather16(const float4* __restrict__ buf, size_t nElem, int iters, float* __restrict__ sink) {
    unsigned long long x = 0x9E3779B97F4A7C15ull * (1 + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        acc += buf[(size_t)(x >> 8) % nElem].x;
    }
    if (acc == 12345.678f) *sink = acc;
}
// LineSegmentDetectorImpl::nfa() split for lane-dynamic scheduling: nfa_setup() covers everything before the binomial-tail
// loop, tail_block() advances the loop by up to eight terms, and the caller finishes with -log10(bin_tail) - logNT.
struct TailState { double term, bin_tail, p_term; int n, i; };           // i = next term index (k+1 .. n)

// returns true when the tail loop has to run; otherwise v is the function value
__device__ __forceinline__ bool nfa_setup(int n, int k, double p, double logNT, const double* __restrict__ lgam, const PLog* __restrict__ plog, TailState& S, double& v) {
    if (n == 0 || k == 0) { v = -logNT; return false; }
    const int h = 1020 - ((__double2hiint(p) >> 20) & 0x7FF);        // p is an exact power of two
    const bool tab = h >= 0 && h < 16 && p == ldexp(0.125, -h);
    if (n == k) { v = -logNT - (double)n * (tab ? plog[h].l10p : log10(p)); return false; }
    const double p_term = p / (1 - p);
    const double log1term = lgam[n + 1] - lgam[k + 1] - lgam[n - k + 1] + (double)k * (tab ? plog[h].lp : log(p)) + (double)(n - k) * (tab ? plog[h].l1mp : log(1.0 - p));
    const double term = exp(log1term);
    if (term == 0.0) {      // double_equal(term, 0) holds only for an exact zero
        v = ((double)k > (double)n * p) ? -log1term / 2.30258509299404568402 - logNT : -logNT;
        return false;
    }
    S.term = term; S.bin_tail = term; S.p_term = p_term; S.n = n; S.i = k + 1;
    return true;
}

// up to eight terms of the tail loop; true when the loop is over (early exit or i > n)
__device__ __forceinline__ bool tail_block(TailState& S, double logNT, const double* __restrict__ rcp) {
    const double tolerance = 0.1;
    const int n = S.n, i0 = S.i;
    double term = S.term, bin_tail = S.bin_tail;
    double mt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {           // independent divisions: issue back to back
        const int i = min(i0 + j, n);
        mt[j] = exact_div((double)(n - i + 1), (double)i, rcp[i]) * S.p_term;
    }
    bool done = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = i0 + j;
        if (i <= n && !done) {
            term *= mt[j];
            bin_tail += term;
            // exact shortcut: past the mode (ratio < 1, and the ratio only shrinks with i) every later term is smaller than
            // this one; once a term is below half an ulp of the sum, no later addition can change bin_tail, and bin_tail is
            // all the function returns from here on.
            if (mt[j] < 1.0 && term < bin_tail * 0x1p-54) done = true;
            if (!done && n - i + 1 < i) {             // bin_term < 1
                const int dec = tail_test_cheap(term, mt[j], n - i + 1, bin_tail, logNT);
                if (dec > 0) done = true;
                else if (dec < 0) {                    // the guard bands overlap (rare): evaluate the reference's expression itself
                    const double err = term * ((1 - pow(mt[j], (double)(n - i + 1))) / (1 - mt[j]) - 1);
                    if (err < tolerance * fabs(-log10(bin_tail) - logNT) * bin_tail) done = true;
                }
            }
        }
    }
    S.term = term; S.bin_tail = bin_tail; S.i = i0 + 8;
    return done || S.i > n;
}

// region point list: first QCAP entries in LDS, the rest in global memory.  entry = x | y<<16
struct RegQ {
    unsigned* lds; unsigned* glb;
    __device__ __forceinline__ unsigned get(int i) const { return i < QCAP ? lds[i] : glb[i]; }
    __device__ __forceinline__ void set(int i, unsigned v) const { if (i < QCAP) lds[i] = v; else glb[i] = v; }
};
__device__ __forceinline__ void rq_fence(int n) { if (n > QCAP) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

__device__ __forceinline__ bool used_get(const unsigned* ub, int idx) { return (ub[idx >> 5] >> (idx & 31)) & 1u; }

__device__ __forceinline__ double readlane_d(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// LineSegmentDetectorImpl::region_grow by one wave.  Eight queue entries are staged at a time (8
// lanes each: the 3x3 neighbourhood in row-major order without its centre), so one global-load round
// trip serves up to eight points.  Lane order == the reference's visiting order, and the region angle only
// changes when a pixel is accepted, so ONE ballot over all staged lanes finds the next accepted
// pixel exactly as the sequential scan would; lanes before it are consumed, lanes after it are
// re-tested against the updated angle.
// LAT selects the accept-chain flavour: v_readlane + pre-converted operands shorten the dependent chain of a lone wave
// (single-frame latency, -11 %), while with six waves per SIMD the LDS-permute form issues fewer wait states (throughput).
template <bool LAT>
__device__ int region_grow_m(int seedX, int seedY, int sw, int sh, float4* __restrict__ pix, const float* __restrict__ ang, const RegQ& rq,
                             double prec, double& regAngleOut) {
    const int lane = threadIdx.x & 63;
    const int seed = seedY * sw + seedX;
    int n = 1;
    double regAngle = (double)ang[seed] * DEG2RAD;
    float sumdx = (float)cos(regAngle), sumdy = (float)sin(regAngle);
    if (lane == 0) { rq.set(0, (unsigned)seedX | ((unsigned)seedY << 16)); pix[seed].x = USED_F; }
    const int g = lane >> 3, k8 = lane & 7;             // group (queue slot) and neighbour slot (centre skipped)
    const int k = k8 + (k8 >= 4 ? 1 : 0);                // row-major 3x3 position 0..8 without 4
    const int dy = k / 3 - 1, dx = k - (k / 3) * 3 - 1;
    int i = 0;
    while (i < n) {
        const int np = min(8, n - i);
        bool cand = false; int nidx = -1, xx = 0, yy = 0; float4 px4 = make_float4(NOTDEF_F, 0.f, 0.f, 0.f);
        if (g < np) {
            const unsigned e = rq.get(i + g);
            xx = (int)(e & 0xFFFF) + dx; yy = (int)(e >> 16) + dy;
            if (xx >= 0 && yy >= 0 && xx < sw && yy < sh) {
                nidx = yy * sw + xx;
                px4 = pix[nidx];                 // .x < 0: NOTDEF or already USED
                cand = px4.x >= 0.f;
            }
        }
        int lastSel = -1;
        const int nBefore = n;
        unsigned long long accMask = 0;                    // lanes accepted from this staging, in lane (= acceptance) order
        const double candRad = LAT ? (double)px4.x * DEG2RAD : 0.0;     // isAligned's operand, converted once per staging
        while (true) {
            bool al;
            if (LAT) {
                double nt = regAngle - candRad;              // is_aligned_val(px4.x, regAngle, prec), same operations
                if (nt < 0) nt = -nt;
                if (nt > M_3_2_PI_) { nt -= M_2PI_; if (nt < 0) nt = -nt; }
                al = cand && lane > lastSel && nt <= prec;
            } else { const bool ok = is_aligned_val(px4.x, regAngle, prec); al = cand && lane > lastSel && ok; }      // straight-line: no exec-masked region around the test
            const unsigned long long m = __ballot(al);
            if (!m) break;
            const int sel = __ffsll((long long)m) - 1;       // wave-uniform: the lane reads below are v_readlane, not LDS permutes
            const int selIdx = LAT ? __builtin_amdgcn_readlane(nidx, sel) : __shfl(nidx, sel, 64);
            if (LAT) accMask |= 1ull << sel;               // lone wave: the used-map / queue stores follow the loop, all lanes at once
            else if (lane == sel) { pix[nidx].x = USED_F; rq.set(n, (unsigned)xx | ((unsigned)yy << 16)); }
            ++n;
            sumdx = __fadd_rn(sumdx, LAT ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px4.y), sel)) : __shfl(px4.y, sel, 64));
            sumdy = __fadd_rn(sumdy, LAT ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px4.z), sel)) : __shfl(px4.z, sel, 64));
            regAngle = (double)fast_atan2_deg<LAT>(sumdy, sumdx) * DEG2RAD;
            if (nidx == selIdx) cand = false;              // the accepted pixel is now USED for every later visitor
            lastSel = sel;
        }
        if (LAT && accMask != 0 && ((accMask >> lane) & 1ull)) { pix[nidx].x = USED_F; rq.set(nBefore + mbcnt(accMask), (unsigned)xx | ((unsigned)yy << 16)); }
        i += np;
    }
    regAngleOut = regAngle;
    return n;
}

// Three fp64 running sums that must be folded strictly in region order (the reference adds point after point).  The wave
// computes the 64 addends of each sum lane-parallel and parks them in LDS; then lanes 0..2 each walk ONE of the three
// arrays, so a step of all three chains is one ds_read_b64 + one v_add_f64 (a readlane walk costs nine instructions).
// acc is live in lanes 0..2 only; ordered_sums_get() broadcasts the results.
struct OrdSum { double acc; };
__device__ __forceinline__ void ordered_sums_add(OrdSum& S, double* __restrict__ red, double v0, double v1, double v2, int cnt, int lane) {
    red[lane] = v0; red[64 + lane] = v1; red[128 + lane] = v2;
    if (lane < 3) {
        const double* src = red + lane * 64;
        double acc = S.acc;
        int j = 0;
        for (; j + 4 <= cnt; j += 4) {
            const double a0 = src[j], a1 = src[j + 1], a2 = src[j + 2], a3 = src[j + 3];
            acc = acc + a0; acc = acc + a1; acc = acc + a2; acc = acc + a3;
        }
        for (; j < cnt; ++j) acc = acc + src[j];
        S.acc = acc;
    }
}
__device__ __forceinline__ double ordered_sums_get(const OrdSum& S, int which) { return readlane_d(S.acc, which); }

// one wave: region2rect + get_theta.  Loads and per-point products run lane-parallel; the
// fp64 sums are then folded strictly in region order (ordered_sums_add), extents by min/max.
__device__ void region2rect_m(const RegQ& rq, int n, int sw, const float4* __restrict__ pix, double regAngle, double prec, double p, RectD& rec, double* __restrict__ red) {
    const int lane = threadIdx.x & 63;
    OrdSum S1; S1.acc = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        double fx = 0, fy = 0, wgt = 0;
        if (i < n) {
            const unsigned e = rq.get(i);
            const int px = e & 0xFFFF, py = e >> 16;
            wgt = sqrt((double)__float_as_int(pix[py * sw + px].w) / 4.0);
            fx = (double)px * wgt; fy = (double)py * wgt;
        }
        ordered_sums_add(S1, red, fx, fy, wgt, min(64, n - base), lane);
    }
    double x = ordered_sums_get(S1, 0), y = ordered_sums_get(S1, 1);
    const double sum = ordered_sums_get(S1, 2);
    x /= sum; y /= sum;
    OrdSum S2; S2.acc = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        double a = 0, b = 0, c = 0;
        if (i < n) {
            const unsigned e = rq.get(i);
            const int px = e & 0xFFFF, py = e >> 16;
            const double wgt = sqrt((double)__float_as_int(pix[py * sw + px].w) / 4.0);
            const double ddx = (double)px - x, ddy = (double)py - y;
            a = ddy * ddy * wgt; b = ddx * ddx * wgt; c = ddx * ddy * wgt;
        }
        ordered_sums_add(S2, red, a, b, -c, min(64, n - base), lane);      // Ixy -= c  ==  Ixy += (-c), exactly
    }
    const double Ixx = ordered_sums_get(S2, 0), Iyy = ordered_sums_get(S2, 1), Ixy = ordered_sums_get(S2, 2);
    const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (fabs(Ixx) > fabs(Iyy)) ? (double)fast_atan2_deg((float)(lambda - Ixx), (float)Ixy)
                                           : (double)fast_atan2_deg((float)Ixy, (float)(lambda - Iyy));
    theta *= DEG2RAD;
    if (fabs(angle_diff_signed(theta, regAngle)) > prec) theta += kPI;
    const double dx = cos(theta), dy = sin(theta);
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;       // running min/max from 0: order independent
    for (int i = lane; i < n; i += 64) {
        const unsigned e = rq.get(i);
        const double rdx = (double)(e & 0xFFFF) - x, rdy = (double)(e >> 16) - y;
        const double l = rdx * dx + rdy * dy, ww = -rdx * dy + rdy * dx;
        l_max = fmax(l_max, l); l_min = fmin(l_min, l); w_max = fmax(w_max, ww); w_min = fmin(w_min, ww);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        l_max = fmax(l_max, __shfl_xor(l_max, o, 64)); l_min = fmin(l_min, __shfl_xor(l_min, o, 64));
        w_max = fmax(w_max, __shfl_xor(w_max, o, 64)); w_min = fmin(w_min, __shfl_xor(w_min, o, 64));
    }
    rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy; rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
    rec.width = w_max - w_min; rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy; rec.prec = prec; rec.p = p;
    if (rec.width < 1.0) rec.width = 1.0;
}

__device__ __forceinline__ double dist_d(double x1, double y1, double x2, double y2) {
    return sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1));
}

enum { NFA_MAXROWS = 64 };
struct NfaGeom { int mx, y0, y1, ly, ry, fl, sl, fr, sr; };

__device__ __forceinline__ int sel4(int i, int a, int b, int c, int d) { return i == 0 ? a : i == 1 ? b : i == 2 ? c : d; }

// rect_nfa's corner bookkeeping (upstream's integer edge stepping and p.y-vs-p.x comparisons
// included), in registers only.
__device__ NfaGeom nfa_geom(const RectD& rec, int sh) {
    const double half_width = rec.width / 2.0;
    const double dyhw = rec.dy * half_width, dxhw = rec.dx * half_width;
    long long k0 = ((long long)((int)(rec.x1 - dyhw) + (1 << 30)) << 32) | (unsigned)((int)(rec.y1 + dxhw) + (1 << 30));
    long long k1 = ((long long)((int)(rec.x2 - dyhw) + (1 << 30)) << 32) | (unsigned)((int)(rec.y2 + dxhw) + (1 << 30));
    long long k2 = ((long long)((int)(rec.x2 + dyhw) + (1 << 30)) << 32) | (unsigned)((int)(rec.y2 - dxhw) + (1 << 30));
    long long k3 = ((long long)((int)(rec.x1 + dyhw) + (1 << 30)) << 32) | (unsigned)((int)(rec.y1 - dxhw) + (1 << 30));
#define CSWAP(a, b) { long long lo = a < b ? a : b, hi = a < b ? b : a; a = lo; b = hi; }
    CSWAP(k0, k1) CSWAP(k2, k3) CSWAP(k0, k2) CSWAP(k1, k3) CSWAP(k1, k2)      // ascending by (x, y)
#undef CSWAP
    const int x0 = (int)(k0 >> 32) - (1 << 30), x1 = (int)(k1 >> 32) - (1 << 30), x2 = (int)(k2 >> 32) - (1 << 30), x3 = (int)(k3 >> 32) - (1 << 30);
    const int y0 = (int)(unsigned)k0 - (1 << 30), y1 = (int)(unsigned)k1 - (1 << 30), y2 = (int)(unsigned)k2 - (1 << 30), y3 = (int)(unsigned)k3 - (1 << 30);
    int imin = 0, imax = 0;
    if (sel4(imin, y0, y1, y2, y3) > y1) imin = 1;
    if (sel4(imax, y0, y1, y2, y3) < y1) imax = 1;
    if (sel4(imin, y0, y1, y2, y3) > y2) imin = 2;
    if (sel4(imax, y0, y1, y2, y3) < y2) imax = 2;
    if (sel4(imin, y0, y1, y2, y3) > y3) imin = 3;
    if (sel4(imax, y0, y1, y2, y3) < y3) imax = 3;
    // leftmost = first untaken with the smallest x (strict compare keeps the earliest)
    int il = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i != imin) { if (il < 0) il = i; else if (sel4(il, x0, x1, x2, x3) > sel4(i, x0, x1, x2, x3)) il = i; }
    int ir = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i != imin && i != il) { if (ir < 0) ir = i; else if (sel4(ir, x0, x1, x2, x3) < sel4(i, x0, x1, x2, x3)) ir = i; }
    int it = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i != imin && i != il && i != ir) it = i;
    NfaGeom g;
    const int mx = sel4(imin, x0, x1, x2, x3), my = sel4(imin, y0, y1, y2, y3);
    const int lx = sel4(il, x0, x1, x2, x3), ly = sel4(il, y0, y1, y2, y3);
    const int rx = sel4(ir, x0, x1, x2, x3), ry = sel4(ir, y0, y1, y2, y3);
    const int tx = sel4(it, x0, x1, x2, x3);
    g.mx = mx; g.ly = ly; g.ry = ry;
    g.fl = (my != ly) ? (mx - lx) / (my - ly) : 0;
    g.sl = (ly != tx) ? (lx - tx) / (ly - tx) : 0;
    g.fr = (my != ry) ? (mx - rx) / (my - ry) : 0;
    g.sr = (ry != tx) ? (rx - tx) / (ry - tx) : 0;
    // rows outside the image are skipped WITHOUT stepping the edges (upstream `continue`)
    g.y0 = max(my, 0); g.y1 = min(sel4(imax, y0, y1, y2, y3), sh - 1);
    return g;
}

// x-range of row y (clipped to the image); a row has seen (y - y0) edge steps, the step taken after
// row t uses the second slope iff t >= ly (resp. ry).
template <bool SMALL>
__device__ __forceinline__ void nfa_row_edges(const NfaGeom& g, int y, long long& lft, long long& rgt) {
    const int steps = y - g.y0;
    int nl2 = 0, nr2 = 0;
    if (steps > 0) {
        nl2 = max(0, y - max(g.ly, g.y0));
        nr2 = max(0, y - max(g.ry, g.y0));
    }
    if (SMALL) {      // images below 32768 x 32768: |slope| < 2^15 and steps < 2^15, every product and sum fits 32 bits
        lft = g.mx + (steps - nl2) * g.fl + nl2 * g.sl;
        rgt = g.mx + (steps - nr2) * g.fr + nr2 * g.sr;
    } else {
        lft = (long long)g.mx + (long long)(steps - nl2) * g.fl + (long long)nl2 * g.sl;
        rgt = (long long)g.mx + (long long)(steps - nr2) * g.fr + (long long)nr2 * g.sr;
    }
}
template <bool SMALL>
__device__ __forceinline__ void nfa_row_range(const NfaGeom& g, int y, int sw, int& xa, int& xb) {
    long long lft, rgt;
    nfa_row_edges<SMALL>(g, y, lft, rgt);
    xa = (int)max(lft, 0LL); xb = (int)min(rgt, (long long)sw - 1);
}
// upper bound of the (unclipped) row width of a rectangle: the edges are linear in y between the corner rows, so the
// maximum sits at one of them.  Only used to pick how many lanes share a row.
__device__ int nfa_max_width(const NfaGeom& g) {
    int best = 1;
    const int ys[8] = {g.y0, g.y1, g.ly - 1, g.ly, g.ly + 1, g.ry - 1, g.ry, g.ry + 1};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int y = min(max(ys[i], g.y0), g.y1);
        long long lft, rgt;
        nfa_row_edges<false>(g, y, lft, rgt);
        best = max(best, (int)min(rgt - lft + 1, 1LL << 20));
    }
    return best;
}

// ------------------------------------------------------------------ rect_improve as staged, fully parallel kernels
// rect_improve (LSD_REFINE_ADV) evaluates the rectangle, then five refinement stages of up to five candidate rectangles
// each; inside a stage the candidates do not depend on which of them is accepted.  Per stage two launches cover every
// candidate of every frame: k_nfa_count (one wave per rectangle: aligned-point counts of the stage's candidates) and
// k_nfa_eval (the binomial-tail NFAs of all candidates, lanes scheduled dynamically) + k_nfa_accept (the reference's sequential acceptance).
struct NfaState { double logNfa; int done, nc; int cnt[6][2]; double val[6]; };      // per rectangle; cnt[k] = {total, aligned}, val[j] = NFA of candidate j

// candidate j of stage `stage` (0..4) grown from the stage's starting rectangle exactly like rect_improve's loops;
// false when iteration j is skipped (width floor) — then every later iteration is skipped too.
__device__ bool stage_cand(const RectD& rec, int stage, int j, RectD& r) {
    const double delta = 0.5, delta_2 = delta / 2.0;
    r = rec;
    for (int n = 0; n <= j; ++n) {
        if (stage == 0 || stage == 4) {
            if (stage == 4 && !((r.width - delta) >= 0.5)) return false;
            r.p /= 2; r.prec = r.p * kPI;
        } else {
            if (!((r.width - delta) >= 0.5)) return false;
            if (stage == 2) { r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2; r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2; }
            if (stage == 3) { r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2; r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2; }
            r.width -= delta;
        }
    }
    return true;
}

__device__ __forceinline__ void load_rect(const double* o, RectD& rec) {
    rec.x1 = o[0]; rec.y1 = o[1]; rec.x2 = o[2]; rec.y2 = o[3]; rec.width = o[4]; rec.x = o[5]; rec.y = o[6];
    rec.theta = o[7]; rec.dx = o[8]; rec.dy = o[9]; rec.prec = o[10]; rec.p = o[11];
}
__device__ __forceinline__ void store_rect(double* o, const RectD& rec) {
    o[0] = rec.x1; o[1] = rec.y1; o[2] = rec.x2; o[3] = rec.y2; o[4] = rec.width; o[5] = rec.x; o[6] = rec.y;
    o[7] = rec.theta; o[8] = rec.dx; o[9] = rec.dy; o[10] = rec.prec; o[11] = rec.p;
}

// angular distance used by isAligned (NOTDEF -> +inf)
__device__ __forceinline__ double align_dist(float aDeg, double theta) {
    const double n_theta = fabs(theta - (double)aDeg * DEG2RAD);          // fabs == the reference's conditional negations
    const double wrapped = fabs(n_theta - M_2PI_);
    return aDeg == NOTDEF_F ? 1e300 : (n_theta > M_3_2_PI_ ? wrapped : n_theta);
}

// k_nfa_count: one wave walks a frame's rectangles.  The corner bookkeeping of rect_nfa (nfa_geom: sorting, slopes, integer
// divisions) is the same few hundred instructions whether one lane or sixty-four execute it, so it runs lane-parallel for a
// batch of up to 64 (rectangle, candidate) items whose results are parked in LDS; the wave then counts the items one after
// the other with all lanes on the pixels.  Counters are wave-uniform (ballot + popcount), so nothing is reduced at the end.
constexpr int EVAL_CH = 1024;          // rectangles per item-list chunk (k_nfa_count, k_nfa_eval)
constexpr int EVAL_REFILL = 16;
struct CntItem { NfaGeom g; int c, j, lg; double theta, prec, p; };      // lg: log2 of the lanes sharing a row

// Pixel walk shared by the two counters below.  A row is shared by 2^lg lanes (lg picked per rectangle from its widest
// row: tall thin rectangles put 32 rows in flight, flat ones spread one row over the whole wave); each lane owns a
// contiguous run of the row and the wave steps through the runs twelve pixels at a time.  Every step starts with
// ballot(pixel exists), which both ends the loop early and counts the rectangle's pixels.

// aligned-point counts of one rectangle for K nested precisions; total = pixels visited
template <int K, bool SMALL>
__device__ __forceinline__ void count_item(const NfaGeom& g, int lg, double theta, const double (&prec)[6], const float* __restrict__ ang, int sw,
                                           int lane, int& totalOut, int (&alg)[6]) {
    const int nrows = g.y1 - g.y0 + 1;
    const int rowsPer = 64 >> lg, r = lane >> lg, sub = lane & ((1 << lg) - 1);
    int total = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) alg[k] = 0;
    for (int t0 = 0; t0 < nrows; t0 += rowsPer) {
        const int t = t0 + r;
        int xa = 0, xb = -1; const int y = g.y0 + t;
        if (t < nrows) nfa_row_range<SMALL>(g, y, sw, xa, xb);
        const int width = max(xb - xa + 1, 0);
        const int share = (width + (1 << lg) - 1) >> lg;
        const int xs = xa + sub * share;
        const int mine = max(min(share, xb - xs + 1), 0);
        const float* row = ang + (size_t)y * sw + xs;
        for (int c0 = 0; __ballot(c0 < mine) != 0; c0 += 12) {
            float a[12];
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = c0 + q < mine ? row[c0 + q] : NOTDEF_F;
            if (__ballot(c0 + 4 < mine)) {
#pragma unroll
                for (int q = 4; q < 8; ++q) a[q] = c0 + q < mine ? row[c0 + q] : NOTDEF_F;
            }
            if (__ballot(c0 + 8 < mine)) {
#pragma unroll
                for (int q = 8; q < 12; ++q) a[q] = c0 + q < mine ? row[c0 + q] : NOTDEF_F;
            }
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const unsigned long long have = __ballot(c0 + q < mine);
                if (!have) break;
                total += __popcll(have);
                const double d = c0 + q < mine ? align_dist(a[q], theta) : 1e300;
#pragma unroll
                for (int k = 0; k < K; ++k) alg[k] += __popcll(__ballot(d <= prec[k]));
            }
        }
    }
    totalOut = total;
}

// Stages 1-3: the (up to five) candidates of a rectangle differ by half-pixel width / offset steps and share theta and the
// tolerance, so they are counted in ONE pass over the union of their rows: the angle test runs once per pixel, membership in
// candidate j is two integer compares against that candidate's own row range (rect_nfa's edge stepping, per candidate).
template <bool SMALL>
__device__ __forceinline__ void count_rect5(const CntItem* __restrict__ it5, int nc, int lg, const float* __restrict__ ang, int sw, int lane,
                                            int (&total)[MAXC], int (&alg)[MAXC]) {
    const double theta = it5[0].theta, prec = it5[0].prec;
    NfaGeom g[MAXC];
#pragma unroll
    for (int j = 0; j < MAXC; ++j) g[j] = it5[j < nc ? j : 0].g;
    int y0u = g[0].y0, y1u = g[0].y1;
#pragma unroll
    for (int j = 1; j < MAXC; ++j) if (j < nc) { y0u = min(y0u, g[j].y0); y1u = max(y1u, g[j].y1); }
    const int nrows = y1u - y0u + 1;
    const int rowsPer = 64 >> lg, r = lane >> lg, sub = lane & ((1 << lg) - 1);
#pragma unroll
    for (int j = 0; j < MAXC; ++j) { total[j] = 0; alg[j] = 0; }
    for (int t0 = 0; t0 < nrows; t0 += rowsPer) {
        const int t = t0 + r;
        const int y = y0u + t;
        int xaj[MAXC], xbj[MAXC];
        int xa = 0x7fffffff, xb = -1;
#pragma unroll
        for (int j = 0; j < MAXC; ++j) {
            xaj[j] = 1; xbj[j] = 0;
            if (j < nc && t < nrows && y >= g[j].y0 && y <= g[j].y1) {
                nfa_row_range<SMALL>(g[j], y, sw, xaj[j], xbj[j]);
                if (xbj[j] >= xaj[j]) { xa = min(xa, xaj[j]); xb = max(xb, xbj[j]); }
                else { xaj[j] = 1; xbj[j] = 0; }
            }
        }
        const int width = xb >= xa ? xb - xa + 1 : 0;
        const int share = (width + (1 << lg) - 1) >> lg;
        const int xs = xa + sub * share;
        const int mine = width > 0 ? max(min(share, xb - xs + 1), 0) : 0;
        const float* row = ang + (size_t)y * sw + xs;
        for (int c0 = 0; __ballot(c0 < mine) != 0; c0 += 12) {
            float a[12];
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = c0 + q < mine ? row[c0 + q] : NOTDEF_F;
            if (__ballot(c0 + 4 < mine)) {
#pragma unroll
                for (int q = 4; q < 8; ++q) a[q] = c0 + q < mine ? row[c0 + q] : NOTDEF_F;
            }
            if (__ballot(c0 + 8 < mine)) {
#pragma unroll
                for (int q = 8; q < 12; ++q) a[q] = c0 + q < mine ? row[c0 + q] : NOTDEF_F;
            }
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const bool have = c0 + q < mine;
                if (!__ballot(have)) break;
                const bool al = have && align_dist(a[q], theta) <= prec;
                const int x = xs + c0 + q;
#pragma unroll
                for (int j = 0; j < MAXC; ++j) {
                    if (j < nc) {
                        const bool in = have && x >= xaj[j] && x <= xbj[j];
                        total[j] += __popcll(__ballot(in));
                        alg[j] += __popcll(__ballot(in && al));
                    }
                }
            }
        }
    }
}

// stage 0 is merged with the initial evaluation: same rectangle, six precisions (p, p/2 .. p/32); stage 4 likewise has one
// geometry and five precisions.  Stages 1-3 change the rectangle itself: up to five candidates per rectangle.
#ifndef SSLAM_COUNT_MINWAVES
#define SSLAM_COUNT_MINWAVES 4
#endif
__global__ __launch_bounds__(64, SSLAM_COUNT_MINWAVES) void k_nfa_count(uint8_t* __restrict__ ws, LsdPlan P, int stage) {
    __shared__ CntItem its[64];
    __shared__ unsigned short act[EVAL_CH];
    const int b = gridDim.x == 1 ? xcd_mix_frame(blockIdx.y, gridDim.y) : blockIdx.y, lane = threadIdx.x;
    uint8_t* base = ws + (size_t)b * P.frameBytes;
    const Misc* misc = (const Misc*)(base + P.offMisc);
    const int nCand = misc->nCand;
    const float* ang = (const float*)(base + P.offAng);
    const double* rects = (const double*)(base + P.offCand);
    NfaState* st = (NfaState*)(base + P.offNfa);
    const int sw = P.sw, sh = P.sh;
    const int per = (nCand + gridDim.x - 1) / gridDim.x;
    const int c0 = blockIdx.x * per, c1 = min(c0 + per, nCand);
    const bool nested = stage == 0 || stage == 4;
    const bool small = sw < 32768 && sh < 32768;
    const int rpb = nested ? 64 : 12;                              // rectangles per batch (stages 1-3: five lanes each)
    for (int chunk = c0; chunk < c1; chunk += EVAL_CH) {
        const int cend = min(chunk + EVAL_CH, c1);
        int nAct = 0;
        for (int cb = chunk; cb < cend; cb += 64) {               // rectangles still being refined
            const int c = cb + lane;
            const bool on = c < cend && (stage == 0 || !st[c].done);
            const unsigned long long m = __ballot(on);
            if (on) act[nAct + mbcnt(m)] = (unsigned short)(c - chunk);
            nAct += __popcll(m);
        }
        __syncthreads();
        for (int a0 = 0; a0 < nAct; a0 += rpb) {
            const int nr = min(rpb, nAct - a0);
            const int nIt = nested ? nr : nr * MAXC;
            {
                const int ri = nested ? lane : lane / MAXC, j = nested ? 0 : lane - ri * MAXC;
                bool valid = false;
                int c = 0;
                if (lane < nIt) {
                    c = chunk + act[a0 + ri];
                    RectD rec, r; load_rect(rects + (size_t)c * 12, rec);
                    if (nested) { r = rec; valid = stage == 0 || (rec.width - 0.5) >= 0.5; }
                    else valid = stage_cand(rec, stage, j, r);
                    CntItem& I = its[lane];
                    I.c = c; I.j = valid ? j : -1;
                    if (valid) {
                        I.g = nfa_geom(r, sh); I.theta = r.theta; I.prec = r.prec; I.p = r.p;
                        const int need = (nfa_max_width(I.g) + (nested ? 0 : 3) + 11) / 12;       // lanes per row so that a run is <= 12 pixels
                        int lg = 1; while ((1 << lg) < need && lg < 6) ++lg;
                        I.lg = lg;
                    }
                }
                const unsigned long long vm = __ballot(valid);
                if (lane < nIt) {
                    if (nested) { if (!valid) st[c].nc = 0; }
                    else if (j == 0) st[c].nc = __popcll((vm >> lane) & 31ull);
                }
            }
            __syncthreads();
            if (!nested) {
                for (int ri = 0; ri < nr; ++ri) {
                    const CntItem* it5 = its + ri * MAXC;
                    int nc = 0;
#pragma unroll
                    for (int j = 0; j < MAXC; ++j) nc += it5[j].j >= 0 ? 1 : 0;           // valid candidates form a prefix
                    if (nc == 0) continue;
                    int total[MAXC], alg[MAXC];
                    if (small) count_rect5<true>(it5, nc, it5[0].lg, ang, sw, lane, total, alg);
                    else count_rect5<false>(it5, nc, it5[0].lg, ang, sw, lane, total, alg);
                    const int c = it5[0].c;
                    if (lane < nc) {
                        const int tj = lane == 0 ? total[0] : lane == 1 ? total[1] : lane == 2 ? total[2] : lane == 3 ? total[3] : total[4];
                        const int aj = lane == 0 ? alg[0] : lane == 1 ? alg[1] : lane == 2 ? alg[2] : lane == 3 ? alg[3] : alg[4];
                        st[c].cnt[lane][0] = tj; st[c].cnt[lane][1] = aj;
                    }
                }
            } else
            for (int it = 0; it < nIt; ++it) {
                const int j = its[it].j;
                if (j < 0) continue;
                const NfaGeom g = its[it].g;
                const double theta = its[it].theta, p = its[it].p;
                const int c = its[it].c;
                double prec[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) prec[k] = stage == 0 ? (k == 0 ? its[it].prec : ldexp(p, -k) * kPI) : ldexp(p, -(k + 1)) * kPI;
                int total, alg[6];
                const int lg = its[it].lg;
                if (stage == 0) { if (small) count_item<6, true>(g, lg, theta, prec, ang, sw, lane, total, alg); else count_item<6, false>(g, lg, theta, prec, ang, sw, lane, total, alg); }
                else { if (small) count_item<5, true>(g, lg, theta, prec, ang, sw, lane, total, alg); else count_item<5, false>(g, lg, theta, prec, ang, sw, lane, total, alg); }
                if (lane == 0) {
                    const int K = stage == 0 ? 6 : 5;
#pragma unroll
                    for (int k = 0; k < 6; ++k) if (k < K) { st[c].cnt[k][0] = total; st[c].cnt[k][1] = alg[k]; }
                    st[c].nc = K;
                }
            }
            __syncthreads();
        }
    }
}

// stage -1: the initial evaluation (cnt[0]); stage 0: cnt[1..5]; stages 1-4: cnt[0..nc).
// One wave walks a frame's (rectangle, candidate) evaluations with lane-level dynamic scheduling: the tail loop's trip count
// varies from 1 to thousands, so lanes that finish pick up the next evaluation instead of idling until the slowest lane of a
// fixed assignment is done.  Setup (log-gamma terms, exp) and the final log10 run only when at least EVAL_REFILL lanes need
// them.  Results land in NfaState::val; k_nfa_accept applies the reference's in-order acceptance.
__device__ __forceinline__ int stage_ncand(const NfaState& s, int stage) {
    if (stage < 0) return 1;
    if (s.done) return 0;
    return stage == 0 ? 5 : stage == 4 ? (s.nc > 0 ? 5 : 0) : s.nc;
}
__global__ __launch_bounds__(64) void k_nfa_eval(uint8_t* __restrict__ ws, LsdPlan P, int stage, const double* __restrict__ lgam) {
    __shared__ unsigned short items[EVAL_CH * 5];            // (rect - chunk) << 3 | candidate
    const int b = gridDim.x == 1 ? xcd_mix_frame(blockIdx.y, gridDim.y) : blockIdx.y, lane = threadIdx.x;
    uint8_t* base = ws + (size_t)b * P.frameBytes;
    Misc* misc = (Misc*)(base + P.offMisc);
    const int nCand = misc->nCand;
    const double* rects = (const double*)(base + P.offCand);
    NfaState* st = (NfaState*)(base + P.offNfa);
    const PLog* plog = (const PLog*)(lgam + P.npx + 4);
    const double* rcp = lgam + P.npx + 4 + 48;
    const int per = (nCand + gridDim.x - 1) / gridDim.x;
    const int c0 = blockIdx.x * per, c1 = min(c0 + per, nCand);
#ifdef SSLAM_LSD_STATS
    long long useful = 0, executed = 0, evals = 0;
#endif
    for (int chunk = c0; chunk < c1; chunk += EVAL_CH) {
        const int cend = min(chunk + EVAL_CH, c1);
        int nItems = 0;
        for (int cb = chunk; cb < cend; cb += 64) {
            const int c = cb + lane;
            const int cnt = c < cend ? stage_ncand(st[c], stage) : 0;
            const int incl = wave_incl_scan(cnt);
            const int ex = nItems + incl - cnt;
            for (int j = 0; j < cnt; ++j) items[ex + j] = (unsigned short)(((c - chunk) << 3) | j);
            nItems += __builtin_amdgcn_readlane(incl, 63);
        }
        __syncthreads();
        int pos = 0, myc = 0, myj = 0;
        bool active = false, pending = false, needLog = false;
        TailState S; S.term = 0; S.bin_tail = 1; S.p_term = 0; S.n = 0; S.i = 1;
        double v = 0;
        while (true) {
            const unsigned long long am = __ballot(active);
            const int nIdle = 64 - __popcll(am);
            const bool more = pos < nItems;
            if ((more && nIdle >= EVAL_REFILL) || am == 0) {
                if (!active && pending) {                      // finish and publish what the idle lanes hold
                    if (needLog) v = -log10(S.bin_tail) - P.logNT;
                    st[myc].val[myj] = v;
                    pending = false;
                }
                if (!more) { if (am == 0) break; }
                else {
                    if (!active) {
                        const int my = pos + mbcnt(~am);
                        if (my < nItems) {
                            const unsigned it = items[my];
                            myc = chunk + (int)(it >> 3); myj = (int)(it & 7);
                            const int kofs = stage == 0 ? 1 : 0;
                            const int n = st[myc].cnt[myj + kofs][0], k = st[myc].cnt[myj + kofs][1];
                            double p = rects[(size_t)myc * 12 + 11];
                            if (stage == 0 || stage == 4) p = ldexp(p, -(myj + 1));       // stage_cand halves p once per step
                            needLog = nfa_setup(n, k, p, P.logNT, lgam, plog, S, v);
                            active = needLog; pending = true;
#ifdef SSLAM_LSD_STATS
                            ++evals;
#endif
                        }
                    }
                    pos += nIdle;
                    continue;
                }
            }
            if (active) {
#ifdef SSLAM_LSD_STATS
                useful += min(8, S.n - S.i + 1);
#endif
                if (tail_block(S, P.logNT, rcp)) active = false;
            }
#ifdef SSLAM_LSD_STATS
            executed += 8;
#endif
        }
        __syncthreads();
    }
#ifdef SSLAM_LSD_STATS
    // cyc[5] = useful tail iterations (upper bound: whole blocks), cyc[6] = lane-iterations the wave executed, cyc[7] = evaluations
    atomicAdd((unsigned long long*)&misc->cyc[5], (unsigned long long)useful); atomicAdd((unsigned long long*)&misc->cyc[6], (unsigned long long)executed);
    atomicAdd((unsigned long long*)&misc->cyc[7], (unsigned long long)evals);
#endif
}

// rect_improve's acceptance, in candidate order, one lane per rectangle (the candidates of a stage do not depend on which of
// them is accepted, so they were all evaluated up front).
__global__ __launch_bounds__(256) void k_nfa_accept(uint8_t* __restrict__ ws, LsdPlan P, int stage) {
    const int b = blockIdx.y;
    uint8_t* base = ws + (size_t)b * P.frameBytes;
    const Misc* misc = (const Misc*)(base + P.offMisc);
    const int nCand = misc->nCand;
    double* rects = (double*)(base + P.offCand);
    NfaState* st = (NfaState*)(base + P.offNfa);
    for (int c = blockIdx.x * 256 + threadIdx.x; c < nCand; c += gridDim.x * 256) {
        if (stage < 0) { const double v0 = st[c].val[0]; st[c].logNfa = v0; st[c].done = v0 > 0.0 ? 1 : 0; continue; }
        const int nc = stage_ncand(st[c], stage);
        if (st[c].done) continue;
        double log_nfa = st[c].logNfa;
        int best = -1;
        for (int q = 0; q < nc; ++q) { const double vq = st[c].val[q]; if (vq > log_nfa) { log_nfa = vq; best = q; } }
        if (best >= 0) {
            RectD rec, r; load_rect(rects + (size_t)c * 12, rec);
            stage_cand(rec, stage, best, r);
            store_rect(rects + (size_t)c * 12, r); st[c].logNfa = log_nfa;
        }
        if (stage < 4 && log_nfa > 0.0) st[c].done = 1;
    }
}

__global__ __launch_bounds__(256) void k_nfa_finish(uint8_t* __restrict__ ws, LsdPlan P) {
    const int b = blockIdx.y;
    uint8_t* base = ws + (size_t)b * P.frameBytes;
    const Misc* misc = (const Misc*)(base + P.offMisc);
    const int nCand = misc->nCand;
    const double* rects = (const double*)(base + P.offCand);
    const NfaState* st = (const NfaState*)(base + P.offNfa);
    float4* seg = (float4*)(base + P.offSeg);
    int* flag = (int*)(base + P.offFlag);
    for (int c = blockIdx.x * 256 + threadIdx.x; c < nCand; c += gridDim.x * 256) {
        const bool ok = st[c].logNfa > 0.0;
        flag[c] = ok ? 1 : 0;
        if (ok) {
            const double* o = rects + (size_t)c * 12;
            const double SCALE = 0.8;
            seg[c] = make_float4((float)((o[0] + 0.5) / SCALE), (float)((o[1] + 0.5) / SCALE), (float)((o[2] + 0.5) / SCALE), (float)((o[3] + 0.5) / SCALE));
        }
    }
}

// One persistent single-wave workgroup per frame: the flsd() main loop replayed in order.
#ifndef SSLAM_LSD_MINWAVES
#define SSLAM_LSD_MINWAVES 6          // waves/SIMD the register allocator must leave room for (6 x 4 SIMDs = 24 frames per CU, LDS allows 32)
#endif
template <bool LAT>
__global__ __launch_bounds__(64, SSLAM_LSD_MINWAVES) void k_lsd_regions(uint8_t* __restrict__ ws, LsdPlan P, const double* __restrict__ lgam) {
    extern __shared__ __align__(16) unsigned dynLds[];           // region queue (first QCAP points)
    const int b = xcd_mix_frame(blockIdx.x, gridDim.x), lane = threadIdx.x;
    uint8_t* base = ws + (size_t)b * P.frameBytes;
    const float* ang = (const float*)(base + P.offAng);
    float4* pix = (float4*)(base + P.offPix);
    const unsigned* order = (const unsigned*)(base + P.offOrder);
    double* candOut = (double*)(base + P.offCand);
    Misc* misc = (Misc*)(base + P.offMisc);
    const int sw = P.sw, sh = P.sh;
    RegQ rq; rq.lds = dynLds; rq.glb = (unsigned*)(base + P.offReg);
    __shared__ double red[3 * 64];                                 // addends of the ordered fp64 sums
    __syncthreads();
    const int nOrd = misc->nDefined;
    const double prec = P.prec, p = P.p, DENSITY_TH = 0.7;
    int nSeg = 0;
    long long cyc0 = 0, cyc1 = 0, cyc2 = 0, cyc3 = 0;
    const long long tStart = __builtin_readcyclecounter();
    for (int pos0 = 0; pos0 < nOrd; pos0 += 64) {
        const int q = pos0 + lane;
        const int idx = q < nOrd ? (int)order[q] : -1;
        int after = -1;                                    // lanes <= after are consumed
        while (true) {
            const bool un = idx >= 0 && lane > after && pix[idx].x >= 0.f;
            const unsigned long long m = __ballot(un);
            if (!m) break;
            const int first = __ffsll((long long)m) - 1;
            after = first;
            const int seed = LAT ? __builtin_amdgcn_readlane(idx, first) : __shfl(idx, first, 64);
            const int sy = seed / sw, sx = seed - sy * sw;
            double regAngle;
            long long t0 = __builtin_readcyclecounter();
            int n = region_grow_m<LAT>(sx, sy, sw, sh, pix, ang, rq, prec, regAngle);
            long long t1 = __builtin_readcyclecounter(); cyc0 += t1 - t0;
            if (n < P.minRegSize) continue;
            RectD rec;
            region2rect_m(rq, n, sw, pix, regAngle, prec, p, rec, red);
            long long t2 = __builtin_readcyclecounter(); cyc1 += t2 - t1;
            // ---- refine (LSD_REFINE_STD part)
            double density = (double)n / (dist_d(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
            if (density < DENSITY_TH) {
                const unsigned e0 = rq.get(0);
                const int x0 = e0 & 0xFFFF, y0 = e0 >> 16;
                const double xc = (double)x0, yc = (double)y0;
                const double ang_c = (double)ang[y0 * sw + x0] * DEG2RAD;
                OrdSum SR; SR.acc = 0; int cnt = 0;
                for (int bs = 0; bs < n; bs += 64) {
                    const int i = bs + lane;
                    double ad = 0; bool in = false;
                    if (i < n) {
                        const unsigned e = rq.get(i);
                        const int px = e & 0xFFFF, py = e >> 16, id = py * sw + px;
                        const float aOrig = ang[id];
                        pix[id].x = aOrig;                 // NOTUSED again
                        if (dist_d(xc, yc, (double)px, (double)py) < rec.width) { in = true; ad = angle_diff_signed((double)aOrig * DEG2RAD, ang_c); }
                    }
                    // points outside the radius contribute an exact +0.0 (the sums start at +0 and can never be -0)
                    const unsigned long long mi = __ballot(in);
                    ordered_sums_add(SR, red, in ? ad : 0.0, in ? ad * ad : 0.0, 0.0, min(64, n - bs), lane);
                    cnt += __popcll(mi);
                }
                const double sum = ordered_sums_get(SR, 0), s_sum = ordered_sums_get(SR, 1);
                const double mean_angle = sum / (double)cnt;
                const double tau = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / (double)cnt + mean_angle * mean_angle);
                n = region_grow_m<LAT>(x0, y0, sw, sh, pix, ang, rq, tau, regAngle);
                if (n < 2) continue;
                region2rect_m(rq, n, sw, pix, regAngle, prec, p, rec, red);
                density = (double)n / (dist_d(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
                if (density < DENSITY_TH) {
                    const long long tr0 = __builtin_readcyclecounter();
                    // reduce_region_radius: sequential swap-with-last removal (the order feeds later sums)
                    const double r1 = (rec.x1 - xc) * (rec.x1 - xc) + (rec.y1 - yc) * (rec.y1 - yc);
                    const double r2 = (rec.x2 - xc) * (rec.x2 - xc) + (rec.y2 - yc) * (rec.y2 - yc);
                    double radSq = r1 > r2 ? r1 : r2;
                    bool good = true;
                    while (density < DENSITY_TH) {
                        radSq *= 0.75 * 0.75;
                        for (int i = 0; i < n; ++i) {
                            const unsigned e = rq.get(i);
                            const int px = e & 0xFFFF, py = e >> 16;
                            const double d2 = ((double)px - xc) * ((double)px - xc) + ((double)py - yc) * ((double)py - yc);
                            if (d2 > radSq) {
                                const int id = py * sw + px;
                                const unsigned last = rq.get(n - 1);
                                if (lane == 0) { pix[id].x = ang[id]; rq.set(i, last); }
                                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                                --n; --i;
                            }
                        }
                        if (n < 2) { good = false; break; }
                        region2rect_m(rq, n, sw, pix, regAngle, prec, p, rec, red);
                        density = (double)n / (dist_d(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
                    }
                    cyc3 += __builtin_readcyclecounter() - tr0;
                    if (!good) continue;
                }
            }
            // ---- hand the rectangle to the NFA stage (rect_improve reads only the static angle map and never touches
            // `used`, so it is not part of the sequential dependency chain: k_lsd_nfa evaluates all candidates in parallel)
            long long t3 = __builtin_readcyclecounter(); cyc2 += t3 - t2;
            if (nSeg < MAX_SEG && lane == 0) {
                double* o = candOut + (size_t)nSeg * 12;
                o[0] = rec.x1; o[1] = rec.y1; o[2] = rec.x2; o[3] = rec.y2; o[4] = rec.width; o[5] = rec.x; o[6] = rec.y;
                o[7] = rec.theta; o[8] = rec.dx; o[9] = rec.dy; o[10] = rec.prec; o[11] = rec.p;
            }
            ++nSeg;
        }
    }
    if (lane == 0) {
        misc->nCand = min(nSeg, MAX_SEG); if (nSeg > MAX_SEG) misc->overflow = 1;
        misc->cyc[0] = cyc0; misc->cyc[1] = cyc1; misc->cyc[2] = cyc2; misc->cyc[3] = cyc3; misc->cyc[4] = __builtin_readcyclecounter() - tStart;
    }
}

// ------------------------------------------------------------------ KeyLine fill + top-N (LSDDetector::detectImpl, ExtractLineSegment :42-51)
__global__ __launch_bounds__(256) void k_keylines(uint8_t* __restrict__ ws, LsdPlan P, int maxLines,
                                                  sslam_keyline* __restrict__ klOut, double* __restrict__ fnOut,
                                                  int* __restrict__ counts, int cap) {
    __shared__ unsigned long long keys[MAX_SEG];
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
        keys[i] = ((unsigned long long)(~__float_as_uint(k.response)) << 32) | (unsigned)i;   // response >= 0: descending response, ascending index (D3 stable)
    }
    __syncthreads();
    int nOut = n;
    const bool doSort = n > maxLines;
    if (doSort) {
        int P2 = 1; while (P2 < n) P2 <<= 1;
        for (int i = n + tid; i < P2; i += 256) keys[i] = ~0ull;
        __syncthreads();
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
        nOut = maxLines;
    }
    nOut = min(nOut, cap);
    for (int i = tid; i < nOut; i += 256) {
        const int src = doSort ? (int)(unsigned)keys[i] : i;
        sslam_keyline k = klw[src];
        if (doSort) k.class_id = i;
        klOut[(size_t)b * cap + i] = k;
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
#pragma unroll
    for (int r = 0; r < STRIP + 2 * R; ++r) {
        if (r >= 2 * R && y0 + r - 2 * R >= h) break;
        unsigned d0, d1, d2;
        load_row12(s, spitch, reflect_row(y0 - R + r, h, R), x4, w, fast, d0, d1, d2);
#pragma unroll
        for (int c = 0; c < 6; ++c) hb[r % 5][c] = hdot(d0, d1, d2, c + 1, T0, T1);      // columns x4+c-3 .. x4+c+1 (sums fit 16 bits: taps sum to 256)
        if (r >= 4) {
            const int q = r - 4;                           // blurred row y0 - 1 + q
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                unsigned acc = 0;
#pragma unroll
                for (int k = 0; k < 5; ++k) acc += hb[(q + k) % 5][c] * taps[k];
                bl[q % 3][c] = (int)(((acc + 32768u) >> 16) & 255u);
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
__constant__ float kGaussG[63];
__constant__ signed char kComb[64];

__global__ __launch_bounds__(64) void k_lbd(const uint8_t* __restrict__ ws, LsdPlan P, const sslam_keyline* __restrict__ kls,
                                            const int* __restrict__ counts, uint8_t* __restrict__ descOut, int cap) {
    __shared__ float rows[8][64];       // pgdL, ngdL, pgdL2, ngdL2, pgdO, ngdO, pgdO2, ngdO2 per row
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
    const float dL0 = (float)cos((double)kl.angle), dL1 = (float)sin((double)kl.angle);      // D5
    const float dO0 = -dL1, dO1 = dL0;
    const int realWidth = P.w, imageWidth = P.w - 1, imageHeight = P.h - 1;
    if (lane < LSP_H) {
        // row start: sCor0 after `lane` steps of (sCorX0 -= dL1, sCorY0 += dL0), sequential float ops
        float sx0 = __fadd_rn(__fadd_rn(__fmul_rn(-dL0, (float)halfWidth), __fmul_rn(dL1, (float)halfHeight)), midX);
        float sy0 = __fadd_rn(__fsub_rn(__fmul_rn(-dL1, (float)halfWidth), __fmul_rn(dL0, (float)halfHeight)), midY);
        for (int r = 0; r < lane; ++r) { sx0 = __fsub_rn(sx0, dL1); sy0 = __fadd_rn(sy0, dL0); }
        float sx = sx0, sy = sy0;
        float pL = 0, nL = 0, pO = 0, nO = 0;
        for (int w0 = 0; w0 < lengthOfLSP; w0 += 8) {
            // coordinates of eight consecutive steps (the float walk itself stays sequential), then the eight gathers together
            int idx8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int tc = (int)(short)(int)roundf(sx);
                const int xCor = tc < 0 ? 0 : (tc > imageWidth ? imageWidth : tc);
                tc = (int)(short)(int)roundf(sy);
                const int yCor = tc < 0 ? 0 : (tc > imageHeight ? imageHeight : tc);
                idx8[u] = yCor * realWidth + xCor;
                sx = __fadd_rn(sx, dL0); sy = __fadd_rn(sy, dL1);
            }
            unsigned g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) g[u] = dxyImg[idx8[u]];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (w0 + u < lengthOfLSP) {
                    const float dx = (float)(short)(g[u] & 0xFFFFu), dy = (float)(short)(g[u] >> 16);
                    const float gDL = __fadd_rn(__fmul_rn(dx, dL0), __fmul_rn(dy, dL1));
                    const float gDO = __fadd_rn(__fmul_rn(dx, dO0), __fmul_rn(dy, dO1));
                    if (gDL > 0) pL = __fadd_rn(pL, gDL); else nL = __fsub_rn(nL, gDL);
                    if (gDO > 0) pO = __fadd_rn(pO, gDO); else nO = __fsub_rn(nO, gDO);
                }
            }
        }
        const float cg = kGaussG[lane];
        pL = __fmul_rn(cg, pL); nL = __fmul_rn(cg, nL); pO = __fmul_rn(cg, pO); nO = __fmul_rn(cg, nO);
        rows[0][lane] = pL; rows[1][lane] = nL; rows[2][lane] = __fmul_rn(pL, pL); rows[3][lane] = __fmul_rn(nL, nL);
        rows[4][lane] = pO; rows[5][lane] = nO; rows[6][lane] = __fmul_rn(pO, pO); rows[7][lane] = __fmul_rn(nO, nO);
    }
    __syncthreads();
    // band sums: lane -> (quantity q = lane/9, band = lane%9); rows visited in increasing hID so the
    // accumulation order equals the reference's (own band, band above, band below contributions interleave by row)
    for (int t = lane; t < 8 * NUM_BANDS; t += 64) {
        const int q = t / NUM_BANDS, bd = t - q * NUM_BANDS;
        const bool sq = (q == 2 || q == 3 || q == 6 || q == 7);
        float acc = 0;
        const int h0 = max(0, (bd - 1) * BAND_W), h1 = min(LSP_H, (bd + 2) * BAND_W);
        for (int hID = h0; hID < h1; ++hID) {
            const int own = hID / BAND_W, m = hID - own * BAND_W;
            float c;
            if (own == bd) c = kGaussL[m + BAND_W];
            else if (own == bd + 1) c = kGaussL[m + 2 * BAND_W];     // row of the band below contributes "upward"
            else c = kGaussL[m];                                       // row of the band above contributes "downward"
            const float v = rows[q][hID];
            acc = sq ? __fadd_rn(acc, __fmul_rn(__fmul_rn(c, c), v)) : __fadd_rn(acc, __fmul_rn(c, v));
        }
        band[q][bd] = acc;
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
    for (int bd = 0; bd < NUM_BANDS; ++bd) {
        const float* d = des + bd * 8;
        tempM = __fadd_rn(tempM, __fmul_rn(d[0], d[0])); tempM = __fadd_rn(tempM, __fmul_rn(d[1], d[1]));
        tempM = __fadd_rn(tempM, __fmul_rn(d[2], d[2])); tempM = __fadd_rn(tempM, __fmul_rn(d[3], d[3]));
        tempS = __fadd_rn(tempS, __fmul_rn(d[4], d[4])); tempS = __fadd_rn(tempS, __fmul_rn(d[5], d[5]));
        tempS = __fadd_rn(tempS, __fmul_rn(d[6], d[6])); tempS = __fadd_rn(tempS, __fmul_rn(d[7], d[7]));
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
    for (int i = 0; i < 72; ++i) temp = __fadd_rn(temp, __fmul_rn(des[i], des[i]));
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
        descOut[((size_t)b * cap + li) * 32 + lane] = (uint8_t)r;
    }
}

__global__ void k_zero_misc(uint8_t* ws, LsdPlan P) {
    Misc* m = (Misc*)(ws + (size_t)blockIdx.x * P.frameBytes + P.offMisc);
    if (threadIdx.x == 0) { m->maxS = 0; m->nDefined = 0; m->nSeg = 0; m->nCand = 0; m->nKl = 0; m->overflow = 0; }
}

}  // namespace

// =============================================================== host side
struct sslam_lines {
    sslam_ctx* ctx;
    int maxLines;
    int planW = 0, planH = 0;
    LsdPlan plan;
    DevBuf dWs, dTabs, dTaps, dLgam, dGtab;
    int wsFrames = 0, lastFrames = 0;
    int lastN = -1;                 // lines of the last sslam_lines_extract (still resident in dKl/dDesc)
    DevBuf dImg, dKl, dDesc, dFn, dCounts;
    HostPinned hOut;
    bool constsUploaded = false;
};

static std::vector<int> taps_q8(int n, double sigma) {
    std::vector<double> k(n);
    double sum = 0, s2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; ++i) { double x = i - (n - 1) * 0.5; k[i] = std::exp(s2 * x * x); sum += k[i]; }
    std::vector<int> t(n);
    double err = 0; long isum = 0;
    for (int i = 0; i < n / 2; ++i) {
        double adj = k[i] / sum * 256.0 + err;
        int v = (int)lrint(adj);
        err = adj - v; t[i] = t[n - 1 - i] = v; isum += v;
    }
    t[n / 2] = (int)(256 - 2 * isum);
    return t;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int lines_build_plan(sslam_lines* L, int w, int h) {
    LsdPlan& P = L->plan;
    memset(&P, 0, sizeof(P));
    P.w = w; P.h = h;
    const double SCALE = 0.8;
    P.sw = (int)lrint(w * SCALE); P.sh = (int)lrint(h * SCALE);
    if (P.sw < 8 || P.sh < 8) { set_error("image %dx%d too small for LSD", w, h); return SSLAM_ERR_UNSUPPORTED; }
    P.spitch = (P.sw + 63) & ~63;
    P.npx = P.sw * P.sh;
    if (P.sw > 65535 || P.sh > 65535) { set_error("image %dx%d too large", w, h); return SSLAM_ERR_UNSUPPORTED; }
    P.nTiles = (P.npx + TILE_PX - 1) / TILE_PX;
    const double ANG_TH = 22.5, QUANT = 2.0, SIGMA_SCALE = 0.6;
    P.prec = kPI * ANG_TH / 180; P.p = ANG_TH / 180;
    P.rho = QUANT / std::sin(P.prec);
    P.logNT = 5 * (std::log10((double)P.sw) + std::log10((double)P.sh)) / 2 + std::log10(11.0);
    P.minRegSize = (int)(size_t)(-P.logNT / std::log10(P.p));
    const double sigma = SIGMA_SCALE / SCALE;
    const unsigned hk = (unsigned)std::ceil(sigma * std::sqrt(2 * 3.0 * std::log(10.0)));
    if (hk != 3) { set_error("unexpected LSD kernel size"); return SSLAM_ERR_UNSUPPORTED; }
    std::vector<int> t7 = taps_q8(7, sigma), t5 = taps_q8(5, 1.0);
    for (int i = 0; i < 7; ++i) P.blurTaps[i] = t7[i];
    for (int i = 0; i < 5; ++i) P.blur5Taps[i] = t5[i];
    // INTER_LINEAR_EXACT tables (D7)
    std::vector<int> tabs;
    auto coeffs = [&](double inv_scale, int ssz, int dsz) {
        double scale = 1.0 / inv_scale;
        for (int v = 0; v < dsz; ++v) {
            double fv = scale * ((double)v + 0.5) - 0.5;
            int iv = (int)std::floor(fv);
            int ofs = 0, c1 = 0;
            if (iv >= 0 && ssz > 1) { if (iv < ssz - 1) { ofs = iv; c1 = (int)lrint((fv - iv) * 256.0); } else { ofs = ssz - 1; c1 = 0; } }
            tabs.push_back(ofs); tabs.push_back(c1);
        }
    };
    P.tabX = 0; coeffs(SCALE, w, P.sw);
    P.tabY = (int)tabs.size(); coeffs(SCALE, h, P.sh);
    int rc;
    if ((rc = L->dTabs.ensure(tabs.size() * sizeof(int)))) return rc;
    SSLAM_HIP(hipMemcpy(L->dTabs.p, tabs.data(), tabs.size() * sizeof(int), hipMemcpyHostToDevice));
    int taps[16] = {0};
    for (int i = 0; i < 7; ++i) taps[i] = t7[i];
    for (int i = 0; i < 5; ++i) taps[8 + i] = t5[i];
    if ((rc = L->dTaps.ensure(sizeof(taps)))) return rc;
    SSLAM_HIP(hipMemcpy(L->dTaps.p, taps, sizeof(taps), hipMemcpyHostToDevice));
    // per-frame workspace layout
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t bpitch = ((size_t)w + 63) & ~(size_t)63;
    P.offBlur = take(bpitch * h);                         // sigma-0.75 blur of the source (LSD)
    P.offAng = take(sizeof(float) * (size_t)P.npx);
    P.offS = take(sizeof(int) * (size_t)P.npx);
    P.offPix = take(sizeof(float4) * (size_t)P.npx);
    P.offOrder = take(sizeof(unsigned) * (size_t)P.npx);
    P.offTileHist = take(sizeof(int) * (size_t)P.nTiles * N_BINS);
    P.offReg = take(sizeof(unsigned) * (size_t)P.npx);
    P.offSeg = take(sizeof(float4) * MAX_SEG);
    P.offCand = take(sizeof(double) * 12 * MAX_SEG);      // candidate rectangles (RectD) awaiting the NFA stage, seed order
    P.offFlag = take(sizeof(int) * MAX_SEG);
    P.offNfa = take(sizeof(NfaState) * MAX_SEG);
    P.offMisc = take(sizeof(Misc));
    P.offDxy = take(sizeof(unsigned) * (size_t)w * h);       // Sobel {dx, dy} of the sigma-1 blur, int16 pairs
    P.offKl = take(sizeof(sslam_keyline) * MAX_SEG);
    P.frameBytes = align_up(off, 4096);
    if (!L->dGtab.p) {   // gradient -> {angle, cos, sin, |g|^2} table (rho depends only on LSD constants)
        if ((rc = L->dGtab.ensure(sizeof(float4) * (size_t)GT * GT))) return rc;
        hipLaunchKernelGGL(k_grad_table, dim3((GT * GT + 255) / 256), dim3(256), 0, L->ctx->stream, L->dGtab.as<float4>(), P.rho);
        SSLAM_HIP(hipStreamSynchronize(L->ctx->stream));
    }
    {   // log-gamma table for nfa(): arguments are integers in [1, npx+2]
        const int nl = P.npx + 4;
        if ((rc = L->dLgam.ensure(sizeof(double) * (2 * (size_t)nl + 48)))) return rc;
        if ((rc = upload_nfa_tables(L->dLgam.as<double>(), nl, L->ctx->stream))) return rc;
    }
    L->planW = w; L->planH = h; L->wsFrames = 0;
    return SSLAM_OK;
}

extern "C" int sslam_lines_create(sslam_ctx* ctx, int max_lines, sslam_lines** out) {
    if (!ctx || !out || max_lines <= 0 || max_lines > MAX_SEG) { set_error("sslam_lines_create: invalid arguments"); return SSLAM_ERR_INVALID; }
    sslam_lines* L = new sslam_lines();
    L->ctx = ctx; L->maxLines = max_lines;
    *out = L;
    return SSLAM_OK;
}

extern "C" int sslam_lines_destroy(sslam_lines* L) {
    if (!L) return SSLAM_OK;
    (void)hipSetDevice(L->ctx->device);
    (void)hipStreamSynchronize(L->ctx->stream);
    DevBuf* bufs[] = {&L->dWs, &L->dTabs, &L->dTaps, &L->dLgam, &L->dGtab, &L->dImg, &L->dKl, &L->dDesc, &L->dFn, &L->dCounts};
    for (DevBuf* b : bufs) b->release();
    L->hOut.release();
    delete L;
    return SSLAM_OK;
}

extern "C" int sslam_lines_extract_batch_dev(sslam_lines* L, const uint8_t* d_images, int w, int h, size_t pitch, size_t image_stride,
                                             int nframes, sslam_keyline* d_kl, uint8_t* d_ldesc, double* d_linefn, int32_t* d_counts,
                                             int cap, void* stream_) {
    if (!L || !d_images || !d_kl || !d_ldesc || !d_linefn || !d_counts || w <= 0 || h <= 0 || nframes <= 0 || cap <= 0 || pitch < (size_t)w) {
        set_error("sslam_lines_extract_batch_dev: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    SSLAM_HIP(hipSetDevice(L->ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : L->ctx->stream;
    int rc;
    if (w != L->planW || h != L->planH) { SSLAM_HIP(hipStreamSynchronize(st)); if ((rc = lines_build_plan(L, w, h))) return rc; }
    if (!L->constsUploaded) {
        float gl[21], gg[63];
        {   // BinaryDescriptor ctor weights (integer divisions are the library's)
            double u = (BAND_W * 3 - 1) / 2, sigma = (BAND_W * 2 + 1) / 2, inv = -1 / (2 * sigma * sigma);
            for (int i = 0; i < 21; ++i) { double d = i - u; gl[i] = (float)std::exp(d * d * inv); }
            u = (NUM_BANDS * BAND_W - 1) / 2; sigma = u; inv = -1 / (2 * sigma * sigma);
            for (int i = 0; i < 63; ++i) { double d = i - u; gg[i] = (float)std::exp(d * d * inv); }
        }
        static const signed char comb[64] = {0, 1, 0, 2, 0, 3, 0, 4, 0, 5, 0, 6, 1, 2, 1, 3, 1, 4, 1, 5, 1, 6, 2, 3, 2, 4, 2, 5, 2, 6, 2, 7,
                                             2, 8, 3, 4, 3, 5, 3, 6, 3, 7, 3, 8, 4, 5, 4, 6, 4, 7, 4, 8, 5, 6, 5, 7, 5, 8, 6, 7, 6, 8, 7, 8};
        SSLAM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(kGaussL), gl, sizeof(gl)));
        SSLAM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(kGaussG), gg, sizeof(gg)));
        SSLAM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(kComb), comb, sizeof(comb)));
        L->constsUploaded = true;
    }
    const LsdPlan& P = L->plan;
    if (nframes > L->wsFrames) {
        SSLAM_HIP(hipStreamSynchronize(st));
        if ((rc = L->dWs.ensure(P.frameBytes * (size_t)nframes))) return rc;
        L->wsFrames = nframes;
    }
    uint8_t* ws = L->dWs.as<uint8_t>();
    const size_t bpitch = ((size_t)w + 63) & ~(size_t)63;
    const int* taps = L->dTaps.as<int>();
    { sslam::ProfScope _ps(L->ctx, "k_zero_misc", st); hipLaunchKernelGGL(k_zero_misc, dim3(nframes), dim3(64), 0, st, ws, P); }
    // LSD: blur(7, 0.75) -> 0.8x -> gradient
    { sslam::ProfScope _ps(L->ctx, "k_blur7", st); hipLaunchKernelGGL(k_blur7, dim3((((w + 3) / 4) * ((h + STRIP - 1) / STRIP) + 255) / 256, nframes), dim3(256), 0, st, d_images, pitch, image_stride,
                       ws + P.offBlur, bpitch, P.frameBytes, w, h, taps); }
    { sslam::ProfScope _ps(L->ctx, "k_lsd_grad", st); hipLaunchKernelGGL(k_lsd_grad, dim3((P.sw + 255) / 256, (P.sh + 3) / 4, nframes), dim3(64, 4), 0, st, ws, P, L->dGtab.as<float4>(), bpitch,
                                                                        L->dTabs.as<int>() + P.tabX, L->dTabs.as<int>() + P.tabY); }
    { sslam::ProfScope _ps(L->ctx, "k_lsd_hist", st); hipLaunchKernelGGL(k_lsd_hist, dim3(P.nTiles, nframes), dim3(64), 0, st, ws, P); }
    { sslam::ProfScope _ps(L->ctx, "k_lsd_scan", st); hipLaunchKernelGGL(k_lsd_scan, dim3(nframes), dim3(1024), 0, st, ws, P); }
    { sslam::ProfScope _ps(L->ctx, "k_lsd_scatter", st); hipLaunchKernelGGL(k_lsd_scatter, dim3(P.nTiles, nframes), dim3(64), 0, st, ws, P); }
    {
        size_t lds = sizeof(unsigned) * QCAP;
        if (const char* e = getenv("SSLAM_LSD_LDS_PAD")) lds = std::max(lds, (size_t)atoi(e));      // experiment knob: cap resident region workgroups per CU
        if (lds > 48 * 1024) {
            SSLAM_HIP(hipFuncSetAttribute((const void*)k_lsd_regions<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            SSLAM_HIP(hipFuncSetAttribute((const void*)k_lsd_regions<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        sslam::ProfScope _ps(L->ctx, "k_lsd_regions", st);
        if (nframes < 1024) hipLaunchKernelGGL(k_lsd_regions<true>, dim3(nframes), dim3(64), lds, st, ws, P, L->dLgam.as<double>());      // lone waves: shortest chain
        else hipLaunchKernelGGL(k_lsd_regions<false>, dim3(nframes), dim3(64), lds, st, ws, P, L->dLgam.as<double>());
    }
    int evalWaves = nframes >= 1024 ? 1 : nframes >= 64 ? 4 : 16;      // waves per frame walking the NFA evaluations
    int countWaves = nframes >= 2048 ? 1 : nframes >= 128 ? 8 : 64;    // waves per frame walking the rectangle counts
    if (const char* e = getenv("SSLAM_COUNT_WAVES")) countWaves = std::max(1, atoi(e));
    if (const char* e = getenv("SSLAM_EVAL_WAVES")) evalWaves = std::max(1, atoi(e));
    for (int stage = 0; stage <= 4; ++stage) {
        { static const char* kCountNames[5] = {"k_nfa_count", "k_nfa_count/s1", "k_nfa_count/s2", "k_nfa_count/s3", "k_nfa_count/s4"};
          sslam::ProfScope _ps(L->ctx, getenv("SSLAM_PROF_STAGES") ? kCountNames[stage] : "k_nfa_count", st); hipLaunchKernelGGL(k_nfa_count, dim3(countWaves, nframes), dim3(64), 0, st, ws, P, stage); }
        if (stage == 0) {
            { sslam::ProfScope _ps(L->ctx, "k_nfa_eval", st); hipLaunchKernelGGL(k_nfa_eval, dim3(evalWaves, nframes), dim3(64), 0, st, ws, P, -1, L->dLgam.as<double>()); }
            { sslam::ProfScope _ps(L->ctx, "k_nfa_accept", st); hipLaunchKernelGGL(k_nfa_accept, dim3(4, nframes), dim3(256), 0, st, ws, P, -1); }
        }
        { sslam::ProfScope _ps(L->ctx, "k_nfa_eval", st); hipLaunchKernelGGL(k_nfa_eval, dim3(evalWaves, nframes), dim3(64), 0, st, ws, P, stage, L->dLgam.as<double>()); }
        { sslam::ProfScope _ps(L->ctx, "k_nfa_accept", st); hipLaunchKernelGGL(k_nfa_accept, dim3(4, nframes), dim3(256), 0, st, ws, P, stage); }
    }
    { sslam::ProfScope _ps(L->ctx, "k_nfa_finish", st); hipLaunchKernelGGL(k_nfa_finish, dim3(4, nframes), dim3(256), 0, st, ws, P); }
    { sslam::ProfScope _ps(L->ctx, "k_keylines", st); hipLaunchKernelGGL(k_keylines, dim3(nframes), dim3(256), 0, st, ws, P, L->maxLines, d_kl, d_linefn, d_counts, cap); }
    // LBD: blur(5, 1) + Sobel fused -> bands
    { sslam::ProfScope _ps(L->ctx, "k_blur_sobel", st); hipLaunchKernelGGL(k_blur_sobel, dim3((((w + 3) / 4) * ((h + STRIP - 1) / STRIP) + 255) / 256, nframes), dim3(256), 0, st, d_images, pitch, image_stride, w, h,
                       (unsigned*)(ws + P.offDxy), P.frameBytes, taps + 8); }
    { sslam::ProfScope _ps(L->ctx, "k_lbd", st); hipLaunchKernelGGL(k_lbd, dim3(std::min(L->maxLines, cap), nframes), dim3(64), 0, st, ws, P, d_kl, d_counts, d_ldesc, cap); }
    SSLAM_HIP(hipGetLastError());
    L->lastFrames = nframes;
    return SSLAM_OK;
}

extern "C" int sslam_lines_extract(sslam_lines* L, const uint8_t* gray, int w, int h, size_t stride, sslam_keyline* kl_out, uint8_t* ldesc_out,
                                   double* linefn_out, int cap, int* n_out) {
    if (!L || !n_out) { set_error("sslam_lines_extract: null handle"); return SSLAM_ERR_INVALID; }
    *n_out = 0;
    if (w == 0 || h == 0 || !gray) return SSLAM_OK;
    if (w < 0 || h < 0 || stride < (size_t)w || !kl_out || !ldesc_out || !linefn_out || cap <= 0) { set_error("sslam_lines_extract: invalid arguments"); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(L->ctx->mu);
    SSLAM_HIP(hipSetDevice(L->ctx->device));
    hipStream_t st = L->ctx->stream;
    const int icap = L->maxLines;
    int rc;
    const size_t dpitch = ((size_t)w + 63) & ~(size_t)63;
    if ((rc = L->dImg.ensure(dpitch * h))) return rc;
    if ((rc = L->dKl.ensure(sizeof(sslam_keyline) * (size_t)icap))) return rc;
    if ((rc = L->dDesc.ensure(32 * (size_t)icap))) return rc;
    if ((rc = L->dFn.ensure(24 * (size_t)icap))) return rc;
    if ((rc = L->dCounts.ensure(16))) return rc;
    const size_t total = 64 + (sizeof(sslam_keyline) + 32 + 24) * (size_t)icap;
    if ((rc = L->hOut.ensure(total))) return rc;
    SSLAM_HIP(hipMemcpy2DAsync(L->dImg.p, dpitch, gray, stride, w, h, hipMemcpyHostToDevice, st));
    if ((rc = sslam_lines_extract_batch_dev(L, L->dImg.as<uint8_t>(), w, h, dpitch, dpitch * h, 1, L->dKl.as<sslam_keyline>(), L->dDesc.as<uint8_t>(),
                                            L->dFn.as<double>(), L->dCounts.as<int>(), icap, st))) return rc;
    uint8_t* hp = L->hOut.as<uint8_t>();
    uint8_t* hk = hp + 64; uint8_t* hd = hk + sizeof(sslam_keyline) * (size_t)icap; uint8_t* hf = hd + 32 * (size_t)icap;
    SSLAM_HIP(hipMemcpyAsync(hp, L->dCounts.p, sizeof(int), hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(hk, L->dKl.p, sizeof(sslam_keyline) * (size_t)icap, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(hd, L->dDesc.p, 32 * (size_t)icap, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(hf, L->dFn.p, 24 * (size_t)icap, hipMemcpyDeviceToHost, st));
    Misc hm;
    SSLAM_HIP(hipMemcpyAsync(&hm, L->dWs.as<uint8_t>() + L->plan.offMisc, sizeof(hm), hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    if (hm.overflow) { set_error("sslam_lines_extract: more than %d candidate rectangles in one frame", MAX_SEG); return SSLAM_ERR_UNSUPPORTED; }
    const int n = *(int*)hp;
    *n_out = n;
    L->lastN = n;
    if (n > cap) { set_error("sslam_lines_extract: %d lines exceed caller capacity %d", n, cap); return SSLAM_ERR_CAPACITY; }
    memcpy(kl_out, hk, sizeof(sslam_keyline) * (size_t)n);
    memcpy(ldesc_out, hd, 32 * (size_t)n);
    memcpy(linefn_out, hf, 24 * (size_t)n);
    return SSLAM_OK;
}

extern "C" int sslam_lines_debug_segments(sslam_lines* L, int frame, float* seg_out, int cap, int* n_out) {
    if (!L || frame < 0 || frame >= L->lastFrames || !n_out) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(L->ctx->device));
    SSLAM_HIP(hipStreamSynchronize(L->ctx->stream));
    const LsdPlan& P = L->plan;
    const uint8_t* base = L->dWs.as<uint8_t>() + (size_t)frame * P.frameBytes;
    Misc m;
    SSLAM_HIP(hipMemcpy(&m, base + P.offMisc, sizeof(m), hipMemcpyDeviceToHost));
    *n_out = m.nSeg;
    int n = std::min(m.nSeg, cap);
    if (n > 0 && seg_out) SSLAM_HIP(hipMemcpy(seg_out, base + P.offSeg, sizeof(float) * 4 * (size_t)n, hipMemcpyDeviceToHost));
    return SSLAM_OK;
}

extern "C" int sslam_lines_debug_cycles(sslam_lines* L, int frame, long long* out8) {
    if (!L || frame < 0 || frame >= L->lastFrames || !out8) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(L->ctx->device));
    SSLAM_HIP(hipDeviceSynchronize());
    const LsdPlan& P = L->plan;
    Misc m;
    SSLAM_HIP(hipMemcpy(&m, L->dWs.as<uint8_t>() + (size_t)frame * P.frameBytes + P.offMisc, sizeof(m), hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i) out8[i] = m.cyc[i];
    return SSLAM_OK;
}

// Self-test of the table-based exact division used in the NFA tail (tests/test_lines_gpu.py): returns the number of
// random (a, b) pairs, 1 <= a, b < n, whose quotient differs from the hardware IEEE division (must be 0).
extern "C" int sslam_selftest_exact_div(sslam_ctx* ctx, int n, long long pairs, long long* mismatches_out) {
    if (!ctx || n < 3 || pairs <= 0 || !mismatches_out) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(ctx->device));
    double* tab = nullptr; unsigned long long* bad = nullptr;
    SSLAM_HIP(hipMalloc(&tab, sizeof(double) * (2 * (size_t)n + 48)));
    SSLAM_HIP(hipMalloc(&bad, sizeof(unsigned long long)));
    SSLAM_HIP(hipMemset(bad, 0, sizeof(unsigned long long)));
    { const int rc = upload_nfa_tables(tab, n, ctx->stream); if (rc) return rc; }
    const int threads = 256 * 1024, iters = (int)((pairs + threads - 1) / threads);
    hipLaunchKernelGGL(k_selftest_div, dim3(1024), dim3(256), 0, ctx->stream, tab + n + 48, n, 0x1234567ull, iters, bad);
    unsigned long long h = 0;
    SSLAM_HIP(hipMemcpyAsync(&h, bad, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SSLAM_HIP(hipStreamSynchronize(ctx->stream));
    (void)hipFree(tab); (void)hipFree(bad);
    *mismatches_out = (long long)h;
    return SSLAM_OK;
}

// Self-test of the guarded fp32 early-exit test of the NFA tail (tests/test_lines_gpu.py): `samples` random inputs, half
// of them on the decision boundary.  disagree_out = decided cases whose verdict differs from the fp64 expression (must be
// 0); ambiguous_out = cases handed to the fp64 expression.
extern "C" int sslam_selftest_tail_test(sslam_ctx* ctx, long long samples, long long* disagree_out, long long* ambiguous_out) {
    if (!ctx || samples <= 0 || !disagree_out || !ambiguous_out) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(ctx->device));
    unsigned long long* d = nullptr;
    SSLAM_HIP(hipMalloc(&d, 2 * sizeof(unsigned long long)));
    SSLAM_HIP(hipMemset(d, 0, 2 * sizeof(unsigned long long)));
    const int threads = 256 * 1024, iters = (int)((samples + threads - 1) / threads);
    const double logNT = 5 * (std::log10(512.0) + std::log10(384.0)) / 2 + std::log10(11.0);
    hipLaunchKernelGGL(k_selftest_tail, dim3(1024), dim3(256), 0, ctx->stream, 0xABCDEF12345ull, iters, logNT, d);
    unsigned long long h[2] = {0, 0};
    SSLAM_HIP(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SSLAM_HIP(hipStreamSynchronize(ctx->stream));
    (void)hipFree(d);
    *disagree_out = (long long)h[0]; *ambiguous_out = (long long)h[1];
    return SSLAM_OK;
}

// FETCH_SIZE calibration: reads `bytes` of a freshly allocated buffer once with 16 B/lane coalesced loads (mode 0) or
// issues bytes/16 scattered 16-B gathers over it (mode 1).  bytes_requested_out = 16 x loads issued.
extern "C" int sslam_selftest_fetch_probe(sslam_ctx* ctx, size_t bytes, int mode, long long* bytes_requested_out) {
    if (!ctx || bytes < (1u << 20) || (mode != 0 && mode != 1) || !bytes_requested_out) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(ctx->device));
    float4* buf = nullptr; float* sink = nullptr;
    SSLAM_HIP(hipMalloc(&buf, bytes));
    SSLAM_HIP(hipMalloc(&sink, sizeof(float)));
    SSLAM_HIP(hipMemsetAsync(buf, 0, bytes, ctx->stream));
    const size_t nElem = bytes / sizeof(float4);
    if (mode == 0) {
        hipLaunchKernelGGL(k_probe_stream16, dim3(8192), dim3(256), 0, ctx->stream, buf, nElem, sink);
        *bytes_requested_out = (long long)(nElem * 16);
    } else {
        const int threads = 8192 * 256, iters = (int)std::max<size_t>(1, nElem / threads);
        hipLaunchKernelGGL(k_probe_gather16, dim3(8192), dim3(256), 0, ctx->stream, buf, nElem, iters, sink);
        *bytes_requested_out = (long long)threads * iters * 16;
    }
    SSLAM_HIP(hipStreamSynchronize(ctx->stream));
    (void)hipFree(buf); (void)hipFree(sink);
    return SSLAM_OK;
}

// Device-resident frame handle of the last sslam_lines_extract call (keylines + LBD descriptors), see sslam_frame_from_orb.
int sslam_frame_from_device(sslam_ctx* ctx, int kind, const void* d_feats, const uint8_t* d_desc, int n, const float bounds[4], sslam_frame** out);
extern "C" int sslam_frame_from_lines(sslam_lines* L, const float bounds[4], sslam_frame** out) {
    if (!L || !bounds || !out) { set_error("sslam_frame_from_lines: invalid arguments"); return SSLAM_ERR_INVALID; }
    if (L->lastN < 0) { set_error("sslam_frame_from_lines: no sslam_lines_extract call to snapshot"); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(L->ctx->mu);
    return sslam_frame_from_device(L->ctx, 1, L->dKl.p, L->dDesc.as<uint8_t>(), L->lastN, bounds, out);
}

#if defined(SSLAM_LBD_DEBUG) || defined(SSLAM_NFA_DEBUG)
// development aid: candidate rectangles (12 doubles) and their NfaState after the last stage (SSLAM_NFA_DEBUG: the LBD dump of
// SSLAM_LBD_DEBUG reuses the candidate buffer)
extern "C" int sslam_lines_debug_nfa(sslam_lines* L, int frame, double* rects_out, void* state_out, int cap, int* n_out, int* state_size) {
    if (!L || frame < 0 || frame >= L->lastFrames) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(L->ctx->device));
    SSLAM_HIP(hipStreamSynchronize(L->ctx->stream));
    const LsdPlan& P = L->plan;
    const uint8_t* base = L->dWs.as<uint8_t>() + (size_t)frame * P.frameBytes;
    Misc m; SSLAM_HIP(hipMemcpy(&m, base + P.offMisc, sizeof(m), hipMemcpyDeviceToHost));
    const int n = std::min(m.nCand, cap);
    *n_out = n; *state_size = (int)sizeof(NfaState);
    SSLAM_HIP(hipMemcpy(rects_out, base + P.offCand, sizeof(double) * 12 * (size_t)n, hipMemcpyDeviceToHost));
    SSLAM_HIP(hipMemcpy(state_out, base + P.offNfa, sizeof(NfaState) * (size_t)n, hipMemcpyDeviceToHost));
    return SSLAM_OK;
}
#endif
#ifdef SSLAM_LBD_DEBUG
// development aid: the normalised 72-float LBD vectors of the last extraction (k_lbd parks them in the candidate buffer)
extern "C" int sslam_lines_debug_lbd_floats(sslam_lines* L, int frame, float* out, int nlines) {
    if (!L || frame < 0 || frame >= L->lastFrames || !out) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(L->ctx->device));
    SSLAM_HIP(hipStreamSynchronize(L->ctx->stream));
    const LsdPlan& P = L->plan;
    SSLAM_HIP(hipMemcpy(out, L->dWs.as<uint8_t>() + (size_t)frame * P.frameBytes + P.offCand, sizeof(float) * 72 * (size_t)nlines, hipMemcpyDeviceToHost));
    return SSLAM_OK;
}
#endif
