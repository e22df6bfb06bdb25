// Line front-end for MI355X (gfx950): LSD detector (REFINE_ADV) + KeyLine fill +
// top-N by response + LBD band descriptor + line equations — the device side of
// LineSegment::ExtractLineSegment (reference src/ExtractLineSegment.cpp:18-69), whose
// arithmetic lives in un-vendored OpenCV 3.4 (imgproc/lsd.cpp, contrib
// line_descriptor LSDDetector.cpp / binary_descriptor.cpp).
//
// Design:
//   * data-parallel stages (blur, 0.8x resample + gradient / level-line angle, 1024-bin stable counting sort of the seeds,
//     Sobel, LBD) are ordinary batched launches                                              -> lsd_front.h, lbd.h
//   * region growing is sequential by definition (seed order + a global `used` map + a region angle that changes with every
//     accepted pixel): one persistent single-wave workgroup per frame replays the exact order; frames of a batch run
//     concurrently, 24 per CU                                                                -> lsd_regions.h
//   * rect_improve reads only the static angle map, so the candidate rectangles of all frames are validated by fully
//     parallel count / evaluate / accept launches per refinement stage                      -> lsd_nfa.h
//   * all order-dependent fp64 sums are accumulated in the reference's order, so the segments match the CPU oracle bit for bit.
// This file: the per-frame workspace plan and the C-ABI entry points (sslam_lines_*).
#include "common.h"
#include <cmath>
#include <cstdlib>
#include <cfloat>
#include <algorithm>

namespace sslam { int launch_nfa_stream(sslam_ctx* ctx, hipStream_t st, uint8_t* ws, const void* plan, size_t planBytes, const double* lgam, uint8_t* clArea, size_t clFrameBytes,
                                        size_t stageOff, int nframes, int waves, long long spinTicks, size_t ldsPad, int takeMax, int sleepReps, const char* scope); }      // lines_nfa.hip
namespace sslam { int launch_nfa_stage(sslam_ctx* ctx, hipStream_t st, uint8_t* ws, const void* plan, size_t planBytes, const double* lgam, int nframes); }      // lines_nfa.hip

using namespace sslam;

namespace {

#include "lsd_plan.h"
#include "lsd_front.h"
#include "lsd_regions.h"
#include "lsd_cluster.h"
#include "lsd_nfa.h"
#include "lbd.h"

__global__ void k_zero_misc(uint8_t* ws, LsdPlan P) {
    Misc* m = (Misc*)(ws + (size_t)blockIdx.x * P.frameBytes + P.offMisc);
    if (threadIdx.x == 0) {
        m->maxS = 0; m->nDefined = 0; m->nSeg = 0; m->nCand = 0; m->nKl = 0; m->overflow = 0; m->claim = 0;
#if defined(SSLAM_LSD_DRIFT_VERIFY) || defined(SSLAM_LSD_CYCLES)
        m->cyc[5] = 0; m->cyc[6] = 0; m->cyc[7] = 0;      // shortcut decisions that disagree with the exact test; (shortcuts << 32) + decisions
#endif
    }
}

// batch status (one workgroup): status[0] = frames holding more lines than the caller's capacity (k_keylines drops the rows past
// it), status[1] = the first of them, status[2] = frames whose LSD produced more than MAX_SEG candidate rectangles (results
// incomplete), status[3] = the first of those; "first" is INT_MAX when there is none
__global__ __launch_bounds__(256) void k_lines_status(const uint8_t* __restrict__ ws, LsdPlan P, int nframes, int maxLines, int cap, int* __restrict__ status) {
    __shared__ int cnt[2], first[2];
    if (threadIdx.x < 2) { cnt[threadIdx.x] = 0; first[threadIdx.x] = 0x7FFFFFFF; }
    __syncthreads();
    for (int b = threadIdx.x; b < nframes; b += 256) {
        const Misc* m = (const Misc*)(ws + (size_t)b * P.frameBytes + P.offMisc);
        if (min(m->nSeg, maxLines) > cap) { atomicAdd(&cnt[0], 1); atomicMin(&first[0], b); }
        if (m->overflow) { atomicAdd(&cnt[1], 1); atomicMin(&first[1], b); }
    }
    __syncthreads();
    if (threadIdx.x == 0) { status[0] = cnt[0]; status[1] = first[0]; status[2] = cnt[1]; status[3] = first[1]; }
}

}  // namespace

// =============================================================== host side
struct sslam_lines {
    sslam_ctx* ctx;
    int maxLines;
    int planW = 0, planH = 0;
    LsdPlan plan;
    DevBuf dWs, dTabs, dTaps, dLgam, dGtab;
    size_t clFrame = 0; int clSlots = 0;      // layout of the last cluster-form launch (sslam_lines_debug_cluster reads it back)
    DevBuf dCl;                     // cluster form of the sequential core (lsd_cluster.h): chunk headers, shared map and list arenas of up to 8 frames
    int wsFrames = 0, lastFrames = 0;
    hipEvent_t coreEvent = nullptr;          // sslam_lines_set_core_event
    hipEvent_t coreWait = nullptr, coreDone = nullptr;      // sslam_lines_set_core_gate
    int lastN = -1;                 // lines of the last sslam_lines_extract (still resident in dKl/dDesc)
    DevBuf dImg, dKl, dDesc, dFn, dCounts;
    HostPinned hOut;
    bool constsUploaded = false;
    int blurVariant = 0;            // sslam_lines_set_blur_variant
    int nfaVariant = 1, lbdBitOrder = 1, lsdResize = 0;      // sslam_lines_set_nfa_variant / _lbd_bit_order / _resize_variant (decisions D11, D12, D7): D11 and D12 default to the OpenCV-as-recalled forms since round 5
    int seedOrder = 0;              // sslam_lines_set_seed_order (decision D2): 1 = the seeds are ordered by the host's std::sort
    int sMin = 0;                   // smallest |g|^2 of a defined pixel (k_grad_smin, with the gradient table)
    bool fusedGeometry = false;     // lines_build_plan: the plan admits k_lsd_grad_fused
    hipStream_t nfaStream = nullptr; hipEvent_t nfaFork = nullptr, nfaJoin = nullptr;      // the NFA stage next to the cluster form of the core (calls of up to 64 frames)
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

// OpenCV 3.4.0's 8-bit Gaussian taps (decision D6's alternative, oracle/cvleaf.h gauss_taps_340): the float kernel times 256, every tap rounded
static std::vector<int> taps_340(int n, double sigma) {
    std::vector<float> k(n); double sum = 0; const double s2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; k[i] = (float)std::exp(s2 * x * x); sum += k[i]; }
    sum = 1. / sum;
    std::vector<int> t(n);
    for (int i = 0; i < n; ++i) { k[i] = (float)(k[i] * sum); t[i] = (int)lrint((double)k[i] * 256.0); }
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
    P.nXB = (P.sw + 255) / 256;
    P.tileRows = std::min(64, std::max(1, TILE_PX / P.sw));
    P.nTiles = (P.sh + P.tileRows - 1) / P.tileRows;
    if (P.tileRows * P.nXB > MAX_TSEG) { set_error("image %dx%d: too many segments per counting-sort tile", w, h); return SSLAM_ERR_UNSUPPORTED; }
    const double ANG_TH = 22.5, QUANT = 2.0, SIGMA_SCALE = 0.6;
    P.prec = kPI * ANG_TH / 180; P.p = ANG_TH / 180;
    P.rho = QUANT / std::sin(P.prec);
    P.logNT = 5 * (std::log10((double)P.sw) + std::log10((double)P.sh)) / 2 + std::log10(11.0);
    P.minRegSize = (int)(size_t)(-P.logNT / std::log10(P.p));
    const double sigma = SIGMA_SCALE / SCALE;
    const unsigned hk = (unsigned)std::ceil(sigma * std::sqrt(2 * 3.0 * std::log(10.0)));
    if (hk != 3) { set_error("unexpected LSD kernel size"); return SSLAM_ERR_UNSUPPORTED; }
    // LSD's pre-blur (sigma 0.75: 0 4 56 136 56 4 0 under both variants) and LBD's (sigma 1: 14 62 104 62 14, or 14 63 103 63 14 under variant 1)
    std::vector<int> t7 = L->blurVariant == 1 ? taps_340(7, sigma) : taps_q8(7, sigma), t5 = L->blurVariant == 1 ? taps_340(5, 1.0) : taps_q8(5, 1.0);
    for (int t : t7) if (t < 0 || t > 255) { set_error("blur taps do not fit a byte"); return SSLAM_ERR_UNSUPPORTED; }
    for (int t : t5) if (t < 0 || t > 255) { set_error("blur taps do not fit a byte"); return SSLAM_ERR_UNSUPPORTED; }
    for (int i = 0; i < 7; ++i) P.blurTaps[i] = t7[i];
    for (int i = 0; i < 5; ++i) P.blur5Taps[i] = t5[i];
    // INTER_LINEAR_EXACT tables (D7): {source offset, q8 coefficient of the second tap}
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
    // D7's alternative (sslam_lines_set_resize_variant(1)): cv::resize(..., dsize, 0, 0, INTER_LINEAR) for 8u -- {source offset, a0 | a1 << 16} with the reference's own
    // float arithmetic: fx = (float)((dx + 0.5) * scale - 0.5), sx = floor, fx -= sx, clamped at both ends, each coefficient cvRound(c * 2048) on its own
    auto coeffs_linear = [&](int ssz, int dsz) {
        const double scale = 1.0 / ((double)dsz / ssz);
        for (int v = 0; v < dsz; ++v) {
            float fv = (float)(((double)v + 0.5) * scale - 0.5);
            int iv = (int)std::floor(fv);
            fv -= (float)iv;
            if (iv < 0) { fv = 0.f; iv = 0; }
            if (iv >= ssz - 1) { fv = 0.f; iv = ssz - 1; }
            const int a0 = (int)lrintf((1.f - fv) * 2048.f), a1 = (int)lrintf(fv * 2048.f);
            tabs.push_back(iv); tabs.push_back(a0 | (a1 << 16));
        }
    };
    P.nfaVariant = L->nfaVariant; P.lbdBitOrder = L->lbdBitOrder; P.lsdResize = L->lsdResize;
    P.tabX = 0; if (P.lsdResize) coeffs_linear(w, P.sw); else coeffs(SCALE, w, P.sw);
    P.tabY = (int)tabs.size(); if (P.lsdResize) coeffs_linear(h, P.sh); else coeffs(SCALE, h, P.sh);
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
    P.tW = P.sw; P.cW = P.sw;
    P.offT = take(sizeof(float) * (size_t)P.npx);
    P.offS = take(sizeof(int) * (size_t)P.npx);
    P.offCs = take(sizeof(float2) * (size_t)P.npx);
    P.offOrder = take(sizeof(unsigned) * (size_t)P.npx);
    P.offTileHist = take(sizeof(int) * (size_t)P.nTiles * N_BINS);
    P.offReg = take(sizeof(unsigned) * std::max((size_t)P.npx, (size_t)P.sh * P.nXB * 256));      // region lists beyond QCAP; before the core: the segments' lists
    P.offComp = P.offReg;
    P.offSegCnt = take(sizeof(int) * (size_t)P.sh * P.nXB);
    P.offSorted = take(sizeof(unsigned) * (size_t)P.sh * P.nXB * 256);      // the segments' lists again, every tile's entries sorted by bin (k_lsd_hist_sort -> k_lsd_scatter_runs)
    P.offSeg = take(sizeof(float4) * MAX_SEG);
    P.offCand = take(sizeof(double) * 12 * MAX_SEG);      // candidate rectangles (RectD) awaiting the NFA stage, seed order
    P.offFlag = take(sizeof(int) * MAX_SEG);
    P.offNfa = take(sizeof(NfaState) * MAX_SEG);
    P.offMisc = take(sizeof(Misc));
    P.offDxy = take(sizeof(unsigned) * (size_t)w * h);       // Sobel {dx, dy} of the sigma-1 blur, int16 pairs
    P.offKl = take(sizeof(sslam_keyline) * MAX_SEG);
    P.offLbdDir = take(sizeof(float2) * MAX_SEG);            // (cos, sin) of every output line's direction: k_keylines -> k_lbd
    P.frameBytes = align_up(off, 4096);
    if (!L->dGtab.p) {   // gradient -> {angle, cos, sin, |g|^2} table (rho depends only on LSD constants), and the smallest defined |g|^2 behind it
        if ((rc = L->dGtab.ensure(sizeof(float4) * ((size_t)GT * GT + 1)))) return rc;
        int* dSmin = (int*)(L->dGtab.as<float4>() + (size_t)GT * GT);
        const int big = 0x7FFFFFFF;
        SSLAM_HIP(hipMemcpy(dSmin, &big, sizeof(int), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_grad_table, dim3((GT * GT + 255) / 256), dim3(256), 0, L->ctx->stream, L->dGtab.as<float4>(), P.rho);
        hipLaunchKernelGGL(k_grad_smin, dim3((2 * 510 * 510 + 256) / 256), dim3(256), 0, L->ctx->stream, dSmin, P.rho);
        SSLAM_HIP(hipStreamSynchronize(L->ctx->stream));
        SSLAM_HIP(hipMemcpy(&L->sMin, dSmin, sizeof(int), hipMemcpyDeviceToHost));
    }
    P.sMin = L->sMin;
    {   // log-gamma table for nfa(): arguments are integers in [1, npx+2]
        const int nl = P.npx + 4;
        if ((rc = L->dLgam.ensure(sizeof(double) * (2 * (size_t)nl + 48)))) return rc;
        if ((rc = upload_nfa_tables(L->dLgam.as<double>(), nl, L->ctx->stream))) return rc;
    }
    // the fused blur + gradient kernel (lsd_front.h k_lsd_grad_fused) covers the 5-to-4 geometry with the plain table pattern and a five-tap blur; anything else keeps k_blur7 + k_lsd_grad
    {
        bool ok = (w % 4) == 0 && (P.sw % 4) == 0 && (P.sh % 4) == 0 && w == 5 * (P.sw / 4) && h == 5 * (P.sh / 4) && w >= 20 && h >= 10 && t7[0] == 0 && t7[6] == 0;
        for (int x = 0; ok && x < P.sw; ++x) ok = tabs[(size_t)P.tabX + 2 * x] == 5 * (x >> 2) + (x & 3);
        for (int y = 0; ok && y < P.sh; ++y) ok = tabs[(size_t)P.tabY + 2 * y] == 5 * (y >> 2) + (y & 3);
        L->fusedGeometry = ok;
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

extern "C" int sslam_lines_set_blur_variant(sslam_lines* L, int variant) {
    if (!L || (variant != 0 && variant != 1)) { set_error("sslam_lines_set_blur_variant: invalid arguments"); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::recursive_mutex> lk(L->ctx->mu);
    if (L->blurVariant != variant) { L->blurVariant = variant; L->planW = L->planH = 0; }      // the taps are part of the plan: rebuilt (and uploaded) by the next extraction
    return SSLAM_OK;
}

// The other stated decisions of the line path that have a selectable alternative (include/sslam_frontend.h).  Each is part of the plan: the next
// extraction rebuilds it.  which: 0 = nfa() first term (D11), 1 = LBD bit order (D12), 2 = the 0.8x rescale (D7).
static int lines_set_variant(sslam_lines* L, int which, int variant, const char* name) {
    if (!L || (variant != 0 && variant != 1)) { set_error("%s: invalid arguments", name); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::recursive_mutex> lk(L->ctx->mu);
    int& field = which == 0 ? L->nfaVariant : which == 1 ? L->lbdBitOrder : which == 2 ? L->lsdResize : L->seedOrder;
    if (field != variant) { field = variant; L->planW = L->planH = 0; }
    return SSLAM_OK;
}
extern "C" int sslam_lines_set_nfa_variant(sslam_lines* L, int variant) { return lines_set_variant(L, 0, variant, "sslam_lines_set_nfa_variant"); }
extern "C" int sslam_lines_set_lbd_bit_order(sslam_lines* L, int variant) { return lines_set_variant(L, 1, variant, "sslam_lines_set_lbd_bit_order"); }
extern "C" int sslam_lines_set_resize_variant(sslam_lines* L, int variant) { return lines_set_variant(L, 2, variant, "sslam_lines_set_resize_variant"); }
extern "C" int sslam_lines_set_seed_order(sslam_lines* L, int variant) { return lines_set_variant(L, 3, variant, "sslam_lines_set_seed_order"); }

// Decision D2's alternative (sslam_lines_set_seed_order(1)): upstream's ll_angle (imgproc/src/lsd.cpp, UPSTREAM-RECALL) fills a vector of {point, bin} in raster order over
// the (w - 1) x (h - 1) pixels that have a gradient and calls std::sort with `a.norm > b.norm` -- an unstable introsort, so the order inside a bin is whatever libstdc++'s
// algorithm leaves.  That permutation is the result of ~17 n dependent comparisons: it is produced HERE, on the host, by the same std::sort over the keys the device computed
// (k_lsd_grad<DENSE> left |g|^2 of every pixel in S), and uploaded as the frame's seed list in place of the device's counting sort.  Synchronous, ~15 ms per 640x480 frame.
static int lines_host_seed_order(sslam_lines* L, uint8_t* ws, int nframes, hipStream_t st) {
    const LsdPlan& P = L->plan;
    struct NormPoint { int x, y, norm; };
    std::vector<int> S((size_t)P.npx);
    std::vector<NormPoint> pts;
    std::vector<unsigned> order;
    SSLAM_HIP(hipStreamSynchronize(st));
    for (int b = 0; b < nframes; ++b) {
        uint8_t* base = ws + (size_t)b * P.frameBytes;
        Misc m;
        SSLAM_HIP(hipMemcpy(&m, base + P.offMisc, sizeof(m), hipMemcpyDeviceToHost));
        SSLAM_HIP(hipMemcpy(S.data(), base + P.offS, sizeof(int) * (size_t)P.npx, hipMemcpyDeviceToHost));
        const double maxGrad = m.maxS > 0 ? std::sqrt((double)m.maxS / 4.0) : -1.0;
        const double binCoef = maxGrad > 0 ? (double)(N_BINS - 1) / maxGrad : 0.0;
        pts.clear(); pts.reserve((size_t)(P.sw - 1) * (P.sh - 1));
        for (int y = 0; y < P.sh - 1; ++y)
            for (int x = 0; x < P.sw - 1; ++x) pts.push_back({x, y, (int)(std::sqrt((double)S[(size_t)y * P.sw + x] / 4.0) * binCoef)});
        std::sort(pts.begin(), pts.end(), [](const NormPoint& a, const NormPoint& c) { return a.norm > c.norm; });
        order.clear();
        for (const NormPoint& q : pts) if (S[(size_t)q.y * P.sw + q.x] >= P.sMin) order.push_back((unsigned)q.x | ((unsigned)q.y << 16));      // defined pixels only, as the device's list
        m.nDefined = (int)order.size();
        if (!order.empty()) SSLAM_HIP(hipMemcpy(base + P.offOrder, order.data(), sizeof(unsigned) * order.size(), hipMemcpyHostToDevice));
        SSLAM_HIP(hipMemcpy(base + P.offMisc + offsetof(Misc, nDefined), &m.nDefined, sizeof(int), hipMemcpyHostToDevice));
    }
    return SSLAM_OK;
}

// Whether a batch of `nframes` runs the sequential core in its guest form (lsd_regions.h): a co-running branch was announced (sslam_lines_set_core_event) and the batch is
// at least two rounds of the persistent grid of 18 (16 for smaller batches) workgroups per compute unit.  18: two SIMDs of a compute unit hold five core waves and keep 32
// registers free, two hold four and keep 128 -- since k_fast_cells needs 29 registers (round 6, call AQ) its waves fit into either, and the core's fifth wave on half of the
// SIMDs pays: 151.3 -> 149.3 ms per step (with the 51-register FAST: 153.2).  17 / 19 / 20 per compute unit: 150.8 / 149.4 / 150.0.
static bool lines_guest_form(const sslam_lines* L, int nframes, int* grid_out) {
    int grid = 18 * L->ctx->num_cus;
    if (2 * grid > nframes) grid = 16 * L->ctx->num_cus;
    bool guest = L->coreEvent != nullptr;
    if (const char* e = getenv("SSLAM_LSD_GUEST")) guest = atoi(e) != 0;
    if (const char* e = getenv("SSLAM_LSD_PERSIST")) { const int g = atoi(e) & ~7; if (g >= 8) grid = g; }
    if (grid_out) *grid_out = grid;
    // at least two full rounds of the persistent grid: a batch of 1.5 grids (sslam_frontend_batch's chunks of 6 144 frames) would run half of the slots twice and the
    // other half once -- measured on the bench's frame sequence, 24 576 frames through host memory: 61.1 k frames/s in this form against 66.0 k in the other (profiles/r06g_*)
    return guest && nframes >= 1024 && 2 * grid <= nframes;
}
extern "C" int sslam_lines_core_guest_form(sslam_lines* L, int nframes) {
    if (!L || nframes <= 0) return 0;
    return lines_guest_form(L, nframes, nullptr) ? 1 : 0;
}

extern "C" int sslam_lines_destroy(sslam_lines* L) {
    if (!L) return SSLAM_OK;
    (void)hipSetDevice(L->ctx->device);
    (void)hipStreamSynchronize(L->ctx->stream);
    if (L->nfaStream) { (void)hipStreamSynchronize(L->nfaStream); (void)hipStreamDestroy(L->nfaStream); (void)hipEventDestroy(L->nfaFork); (void)hipEventDestroy(L->nfaJoin); }
    DevBuf* bufs[] = {&L->dWs, &L->dCl, &L->dTabs, &L->dTaps, &L->dLgam, &L->dGtab, &L->dImg, &L->dKl, &L->dDesc, &L->dFn, &L->dCounts};
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
    std::lock_guard<std::recursive_mutex> lk(L->ctx->mu);      // plan, workspace and profile records are shared state
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
        float gl2[21];
        for (int i = 0; i < 21; ++i) { volatile float c = gl[i]; gl2[i] = c * c; }      // one fp32 rounding, as the device's __fmul_rn(c, c)
        SSLAM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(kGaussL), gl, sizeof(gl)));
        SSLAM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(kGaussL2), gl2, sizeof(gl2)));
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
    // LBD's gradient image (sigma-1 blur + Sobel of the SOURCE) does not depend on the segments; SSLAM_LBD_SOBEL=early launches it here, in
    // the prologue (8 ms with the chip to itself instead of 35 ms under the point branch) -- measured: the step does not respond to where a
    // kernel runs, only to how long the kernels take alone (194.1 vs 191.4 ms; docs/history/DESIGN_rounds_1-4.md 5g), so it stays behind the NFA stage
    static const bool sobelEarly = [] { const char* e = getenv("SSLAM_LBD_SOBEL"); return e && !strcmp(e, "early"); }();
    bool sobelDone = false;
    auto launch_blur_sobel = [&](hipStream_t s) {
        sslam::ProfScope _ps(L->ctx, "k_blur_sobel", s);
        hipLaunchKernelGGL(k_blur_sobel, dim3((((w + 3) / 4) * ((h + STRIP - 1) / STRIP) + 255) / 256, nframes), dim3(256), 0, s, d_images, pitch, image_stride, w, h,
                           (unsigned*)(ws + P.offDxy), P.frameBytes, taps + 8);
        sobelDone = true;
    };
    if (sobelEarly) launch_blur_sobel(st);
    // Work forked onto the side stream (L->nfaStream) is joined into `st` on EVERY way out of this function -- also the early error returns: the caller may reuse or
    // free the workspace as soon as `st` is idle, and the side stream's kernels write into it
    struct SideJoin {
        sslam_lines* L; hipStream_t st; bool forked = false, joined = false;
        void join() { if (forked && !joined) { (void)hipEventRecord(L->nfaJoin, L->nfaStream); (void)hipStreamWaitEvent(st, L->nfaJoin, 0); joined = true; } }
        ~SideJoin() { join(); }
    } side{L, st};
    auto side_stream_ready = [&]() -> int {
        if (!L->nfaStream) {
            SSLAM_HIP(hipStreamCreateWithFlags(&L->nfaStream, hipStreamNonBlocking));
            SSLAM_HIP(hipEventCreateWithFlags(&L->nfaFork, hipEventDisableTiming));
            SSLAM_HIP(hipEventCreateWithFlags(&L->nfaJoin, hipEventDisableTiming));
        }
        return SSLAM_OK;
    };
    // LSD: blur(7, 0.75) -> 0.8x -> gradient; one kernel where the geometry allows (SSLAM_LSD_FUSED=0: always two, A/B)
    const bool fused = L->fusedGeometry && GRAD_ROWS == 8 && ((uintptr_t)d_images & 3) == 0 && (pitch & 3) == 0 && (image_stride & 3) == 0 && pitch <= 0x7FFFFFFF &&
                       !(getenv("SSLAM_LSD_FUSED") && atoi(getenv("SSLAM_LSD_FUSED")) == 0);
    if (fused) {
      sslam::ProfScope _ps(L->ctx, "k_lsd_grad", st);
      const dim3 gg(P.nXB, (P.sh + 31) / 32, nframes);      // (frame-walking workgroups -- a grid of 1 024 / 2 048 -- were measured for this kernel too, call U: 171 / 175 ms per step against 158; the kernel keeps its frame loop, the grid covers the batch)
      const int* tabX = L->dTabs.as<int>() + P.tabX; const int* tabY = L->dTabs.as<int>() + P.tabY;
      switch ((P.lsdResize ? 1 : 0) | (L->seedOrder ? 2 : 0)) {
          case 0: hipLaunchKernelGGL(k_lsd_grad_fused<0>, gg, dim3(64, 4), 0, st, d_images, pitch, image_stride, ws, P, L->dGtab.as<float4>(), tabX, tabY, taps, nframes); break;
          case 1: hipLaunchKernelGGL(k_lsd_grad_fused<1>, gg, dim3(64, 4), 0, st, d_images, pitch, image_stride, ws, P, L->dGtab.as<float4>(), tabX, tabY, taps, nframes); break;
          case 2: hipLaunchKernelGGL(k_lsd_grad_fused<2>, gg, dim3(64, 4), 0, st, d_images, pitch, image_stride, ws, P, L->dGtab.as<float4>(), tabX, tabY, taps, nframes); break;
          default: hipLaunchKernelGGL(k_lsd_grad_fused<3>, gg, dim3(64, 4), 0, st, d_images, pitch, image_stride, ws, P, L->dGtab.as<float4>(), tabX, tabY, taps, nframes); break;
      }
    } else {
    { sslam::ProfScope _ps(L->ctx, "k_blur7", st); hipLaunchKernelGGL(k_blur7, dim3((((w + 3) / 4) * ((h + STRIP - 1) / STRIP) + 255) / 256, nframes), dim3(256), 0, st, d_images, pitch, image_stride,
                       ws + P.offBlur, bpitch, P.frameBytes, w, h, taps); }
    { sslam::ProfScope _ps(L->ctx, "k_lsd_grad", st);
      const dim3 gg(P.nXB, (P.sh + 4 * GRAD_ROWS - 1) / (4 * GRAD_ROWS), nframes);
      const int* tabX = L->dTabs.as<int>() + P.tabX; const int* tabY = L->dTabs.as<int>() + P.tabY;
      switch ((P.lsdResize ? 1 : 0) | (L->seedOrder ? 2 : 0)) {
          case 0: hipLaunchKernelGGL(k_lsd_grad<0>, gg, dim3(64, 4), 0, st, ws, P, L->dGtab.as<float4>(), bpitch, tabX, tabY); break;
          case 1: hipLaunchKernelGGL(k_lsd_grad<1>, gg, dim3(64, 4), 0, st, ws, P, L->dGtab.as<float4>(), bpitch, tabX, tabY); break;
          case 2: hipLaunchKernelGGL(k_lsd_grad<2>, gg, dim3(64, 4), 0, st, ws, P, L->dGtab.as<float4>(), bpitch, tabX, tabY); break;
          default: hipLaunchKernelGGL(k_lsd_grad<3>, gg, dim3(64, 4), 0, st, ws, P, L->dGtab.as<float4>(), bpitch, tabX, tabY); break;
      } }
    }
    if (L->seedOrder) { if ((rc = lines_host_seed_order(L, ws, nframes, st))) return rc; }
    else {
    // tile-sorted runs (lsd_front.h) where a sorted entry's packing fits (scaled image up to 2048 x 2048); SSLAM_LSD_SORT_RUNS=0: the round-1-5 kernels (A/B)
    const bool runs = P.sw <= (1 << SORT_XY_BITS) && P.sh <= (1 << SORT_XY_BITS) && !(getenv("SSLAM_LSD_SORT_RUNS") && atoi(getenv("SSLAM_LSD_SORT_RUNS")) == 0);
    if (runs) {
    { sslam::ProfScope _ps(L->ctx, "k_lsd_hist", st); hipLaunchKernelGGL(k_lsd_hist_sort, dim3(sort_grid(P.nTiles, nframes)), dim3(64), 0, st, ws, P, nframes); }
    { sslam::ProfScope _ps(L->ctx, "k_lsd_scan", st); hipLaunchKernelGGL(k_lsd_scan, dim3(nframes), dim3(1024), 0, st, ws, P); }
    { sslam::ProfScope _ps(L->ctx, "k_lsd_scatter", st); hipLaunchKernelGGL(k_lsd_scatter_runs, dim3(sort_grid(P.nTiles, nframes)), dim3(64), 0, st, ws, P, nframes); }
    } else {
    { sslam::ProfScope _ps(L->ctx, "k_lsd_hist", st); hipLaunchKernelGGL(k_lsd_hist, dim3(sort_grid(P.nTiles, nframes)), dim3(64), 0, st, ws, P, nframes); }
    { sslam::ProfScope _ps(L->ctx, "k_lsd_scan", st); hipLaunchKernelGGL(k_lsd_scan, dim3(nframes), dim3(1024), 0, st, ws, P); }
    { sslam::ProfScope _ps(L->ctx, "k_lsd_scatter", st); hipLaunchKernelGGL(k_lsd_scatter, dim3(sort_grid(P.nTiles, nframes)), dim3(64), 0, st, ws, P, nframes); }
    }
    }
    bool nfaStreamed = false; size_t nfaStageOff = 0;
    {
        size_t lds = sizeof(unsigned) * (QCAP + 4);      // + the sink slot behind the queue (region_grow_w)
        if (const char* e = getenv("SSLAM_LSD_LDS_PAD")) lds = std::max(lds, (size_t)atoi(e));      // experiment knob: cap resident region workgroups per CU
        if (lds > 48 * 1024) {
            SSLAM_HIP(hipFuncSetAttribute((const void*)k_lsd_regions<true, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            SSLAM_HIP(hipFuncSetAttribute((const void*)k_lsd_regions<false, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            SSLAM_HIP(hipFuncSetAttribute((const void*)k_lsd_regions<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            SSLAM_HIP(hipFuncSetAttribute((const void*)k_lsd_regions<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        if (L->coreWait) SSLAM_HIP(hipStreamWaitEvent(st, L->coreWait, 0));      // (sslam_lines_set_core_gate: another extractor's core has the wave slots until then)
        if (L->coreEvent) SSLAM_HIP(hipEventRecord(L->coreEvent, st));
        sslam::ProfScope _ps(L->ctx, "k_lsd_regions", st);
        bool lone = nframes < 1024;
        const bool spillFree = !(getenv("SSLAM_LSD_SPILLFREE") && atoi(getenv("SSLAM_LSD_SPILLFREE")) == 0);
        if (const char* e = getenv("SSLAM_LSD_FLAVOUR")) lone = e[0] == 'l' || e[0] == 'c';      // experiment knob: "cl" / "lat" / "thr"
        // Up to 64 frames (one to eight per XCD) whose frame-wide bitmap fits the main wave's LDS: the cluster form -- helper waves on several compute
        // units, results through global memory, monotonic pixel map (lsd_cluster.h).  SSLAM_LSD_CLUSTER=0 (or SSLAM_LSD_FLAVOUR=lat) takes lone
        // waves instead; SSLAM_CL_WGS = workgroups per frame (4 waves each), SSLAM_CL_WINDOW = how many sub-chunks of 16 seed positions the
        // helpers may run ahead, SSLAM_CL_SMAP = cell size (log2) of the shared map that steers their seed choice (-1: none).
        int CL_MAXFRAMES = 64;      // up to eight frames per XCD, four workgroups each.  Per call, cluster form against the multi-wave form of rounds 2-4 (tools/small_batch_probe.py, round 3):
                                    // 1 frame 5.9 / 7.9 ms, 8: 8.5 / 11.2, 16: 9.0 / 11.9, 24: 9.4 / 12.6, 32: 10.3 / 13.1, 64: 13.5 / 15.7, 96: 21.6 / 15.9
        if (const char* e = getenv("SSLAM_CL_MAXFRAMES")) CL_MAXFRAMES = std::max(1, std::min(128, atoi(e)));      // experiment knob
        const bool bigFrame = P.sw > TorusFrame::XMASK + 1 || P.sh > TorusFrame::YMASK + 1;      // the main wave's bitmap in global memory instead of LDS
        bool cluster = nframes <= CL_MAXFRAMES && P.sw <= TorusGlobal::XMASK + 1 && P.sh <= TorusGlobal::YMASK + 1;
        if (const char* e = getenv("SSLAM_LSD_FLAVOUR")) cluster = cluster && e[0] == 'c';      // "cl" / "lat" / "thr"
        if (const char* e = getenv("SSLAM_LSD_CLUSTER")) cluster = cluster && atoi(e) != 0;
        if (cluster) {
            int nWG = 10, window = 0, clShift = 0;
            if (const char* e = getenv("SSLAM_CL_WGS")) nWG = std::max(1, std::min(CL_MAXWG, atoi(e)));
            if (const char* e = getenv("SSLAM_CL_WINDOW")) window = std::max(-1, atoi(e)); else window = 640 / CL_SUB;      // in sub-chunks (640 seed positions); -1: no helpers at all (the main wave alone)
            if (const char* e = getenv("SSLAM_CL_SMAP")) clShift = atoi(e);
            if (getenv("SSLAM_CL_NO_FEEDER") && window >= 0) window |= 1 << 20;      // experiment knob: the main wave fetches everything itself
            const int clSpecWords = clShift < 0 ? 0 : (((P.sw + (1 << clShift) - 1) >> clShift) * ((P.sh + (1 << clShift) - 1) >> clShift) + 31) / 32;
            const size_t maxSubs = ((size_t)P.npx + CL_SUB - 1) / CL_SUB;
            const size_t zeroBytes = 512 + ((maxSubs * sizeof(ClSub) + 511) & ~(size_t)511) + 4 * (size_t)((clSpecWords + 127) & ~127) + (bigFrame ? 4 * (size_t)TorusGlobal::WORDS : 0);      // control block, sub-chunk states / flags, shared map (+ the main wave's bitmap)
            if (nframes > 8) nWG = std::max(2, std::min(nWG, 32 / ((nframes + 7) / 8)));      // the frames of an XCD share its 32 compute units
            // per frame: the zeroed head, two result records per seed position, one 256 KB list arena per HELPER THAT EXISTS ((nWG - 1) x CL_HPW: 27 by default;
            // rounds 1-3 sized it for 64).  One slot per frame of the call: blocks with b >= nframes return at once, so the XCD-aligned grid needs no padding slots
            // (a single 640x480 frame held 8 slots of 30 MB before).
            // The NFA stage runs NEXT TO the core on a second stream, on the rectangles the main wave has published so far (lsd_nfa.h, k_nfa_stream) -- the default since round 5
            // (whole GPU suite with the knob exported, single frames 6.02 / 7.18 -> 5.76 / 6.89 ms p50 / p90, calls of 2 .. 64 frames -8 .. -24 %: profiles/r05a_*).
            // SSLAM_NFA_STREAM=0: the stage behind the core as one launch (the round-4 default); =n > 1: n consumer waves per frame (default 16)
            int nfaStreamWaves = 16;
            if (const char* e = getenv("SSLAM_NFA_STREAM")) { nfaStreamWaves = atoi(e); if (nfaStreamWaves == 1) nfaStreamWaves = 16; nfaStreamWaves = std::max(0, std::min(64, nfaStreamWaves)); }
            const size_t stageOff = zeroBytes + maxSubs * CL_RES * sizeof(ClRec) + 4 * (size_t)CL_ARENA * (size_t)std::max(1, (nWG - 1) * CL_HPW);      // (k_lsd_regions_cl_stream: cl.candStage)
            const size_t clFrame = align_up(stageOff + (nfaStreamWaves ? sizeof(double) * 12 * (size_t)MAX_SEG : 0), 4096);
            L->clFrame = clFrame; L->clSlots = nframes;
            const size_t clSlots = (size_t)nframes;
            if (L->dCl.cap < clFrame * clSlots) { SSLAM_HIP(hipStreamSynchronize(st)); if ((rc = L->dCl.ensure(clFrame * clSlots))) return rc; }
            for (int f = 0; f < nframes; ++f) SSLAM_HIP(hipMemsetAsync(L->dCl.as<uint8_t>() + (size_t)f * clFrame, 0, zeroBytes, st));
            const size_t clLds = sizeof(unsigned) * std::max((size_t)QCAP + 4 + (bigFrame ? 0 : TorusFrame::WORDS) + CL_SCAN + CL_RING_WORDS, (size_t)CL_HPW * (CL_LIST + ClTorus::WORDS));      // the main wave's workgroup / a helper workgroup
            SSLAM_HIP(hipFuncSetAttribute((const void*)k_lsd_regions_cl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)clLds));
            if (nfaStreamWaves) {
                if ((rc = side_stream_ready())) return rc;
                long long spinTicks = 5000000;       // 50 ms of the 100 MHz clock (ten single-frame cores): a consumer wave that has seen no progress for that long leaves its blocks to the launch behind the core
                                                      // (0.2 s in round 5: on a GPU shared with another process the consumers can be resident before the producer, and the wait was the spike)
                if (const char* e = getenv("SSLAM_NFA_STREAM_TICKS")) spinTicks = std::max(0ll, atoll(e));
                size_t nfaLdsPad = 40 * 1024;        // (lines_nfa.hip: keeps the consumers off the main wave's and the helpers' compute units)
                int nfaTakeMax = NFA_STREAM_BLOCK;   // rectangles per claim at most (lsd_nfa.h)
                if (const char* e = getenv("SSLAM_NFA_STREAM_TAKE")) nfaTakeMax = atoi(e);
                int nfaSleep = 1;                     // s_sleep(127) between two polls of a waiting consumer (lsd_nfa.h)
                if (const char* e = getenv("SSLAM_NFA_STREAM_SLEEP")) nfaSleep = std::max(0, std::min(64, atoi(e)));
                if (const char* e = getenv("SSLAM_NFA_STREAM_LDS")) nfaLdsPad = (size_t)std::max(0, std::min(48 * 1024, atoi(e)));
                SSLAM_HIP(hipEventRecord(L->nfaFork, st));      // (the prologue's planes and the zeroed slot heads are what the consumers need)
                SSLAM_HIP(hipFuncSetAttribute((const void*)k_lsd_regions_cl_stream, hipFuncAttributeMaxDynamicSharedMemorySize, (int)clLds));
                hipLaunchKernelGGL(k_lsd_regions_cl_stream, dim3(8 * nWG * ((nframes + 7) / 8)), dim3(64 * CL_WAVES), clLds, st, ws, P, L->dCl.as<uint8_t>(), clFrame, nframes, nWG, clSpecWords, clShift, window);
                SSLAM_HIP(hipStreamWaitEvent(L->nfaStream, L->nfaFork, 0));
                side.forked = true;
                // LBD's gradient image depends on the source alone: on the second stream it runs under the core instead of behind the NFA stage (28 us of a single frame's
                // 5.4 ms; the join below orders it before k_lbd)
                if (!sobelDone && !getenv("SSLAM_LBD_SOBEL_MAIN")) launch_blur_sobel(L->nfaStream);      // (the knob: A/B, GPU call I -- 5.43 -> 5.38 ms p50)
                if ((rc = sslam::launch_nfa_stream(L->ctx, L->nfaStream, ws, &P, sizeof(P), L->dLgam.as<double>(), L->dCl.as<uint8_t>(), clFrame, stageOff, nframes, nfaStreamWaves, spinTicks, nfaLdsPad, nfaTakeMax, nfaSleep, nullptr))) return rc;
                nfaStreamed = true; nfaStageOff = stageOff;
            } else
            hipLaunchKernelGGL(k_lsd_regions_cl, dim3(8 * nWG * ((nframes + 7) / 8)), dim3(64 * CL_WAVES), clLds, st, ws, P, L->dCl.as<uint8_t>(), clFrame, nframes, nWG, clSpecWords, clShift, window);
        } else if (lone) {      // lone waves: shortest chain.  At most one wave per SIMD is resident, so the instantiation that spills nothing costs no occupancy (SSLAM_LSD_SPILLFREE=0: the six-wave one, A/B)
            if (spillFree) hipLaunchKernelGGL((k_lsd_regions<true, 4>), dim3(nframes), dim3(64), lds, st, ws, P, L->dLgam.as<double>(), nframes);
            else hipLaunchKernelGGL((k_lsd_regions<true, 6>), dim3(nframes), dim3(64), lds, st, ws, P, L->dLgam.as<double>(), nframes);
        } else {
            // A caller that announced a branch running beside the core (sslam_lines_set_core_event: the bench step's point branch waits for that event) gets the GUEST form
            // (lsd_regions.h): 16 - 18 persistent workgroups per compute unit of the four-wave instantiation, a third of the registers free for the other branch's waves.
            // SSLAM_LSD_GUEST=0 / 1 overrides (A/B), SSLAM_LSD_PERSIST=g sets the grid.  (LBD's blur + Sobel as one more guest under the core, on the side stream, was
            // measured too: its 126-VGPR waves take the slots FAST needs -- 167.7 ms per step against 160.8 with the kernel in the tail; profiles/r06d_*.)
            int grid = 0;
            if (lines_guest_form(L, nframes, &grid)) hipLaunchKernelGGL((k_lsd_regions<false, 4>), dim3(grid), dim3(64), lds, st, ws, P, L->dLgam.as<double>(), nframes);
            else if (spillFree && nframes <= 16 * L->ctx->num_cus)      // up to four waves per SIMD anyway (BASELINE configs[3]: 3 072 frames): the spill-free instantiation, one workgroup per frame
                hipLaunchKernelGGL((k_lsd_regions<false, 4>), dim3(nframes), dim3(64), lds, st, ws, P, L->dLgam.as<double>(), nframes);
            else hipLaunchKernelGGL((k_lsd_regions<false, 6>), dim3(nframes), dim3(64), lds, st, ws, P, L->dLgam.as<double>(), nframes);
        }
    }
    if (L->coreDone) SSLAM_HIP(hipEventRecord(L->coreDone, st));
    // LBD's gradient image needs the source alone and is bandwidth-bound; the NFA stage behind the core is latency-bound (a third of the vector pipes busy): on the side
    // stream the one runs beside the other instead of behind it -- for a caller WITHOUT a branch of its own beside this one (no core event): 187.5 against 190.1 ms per
    // one-stream step.  With the point branch still running there (two-stream step) the pair costs 3.5 ms instead (164.1 against 160.2: k_blur_sobel and k_nfa_all are 56 KB
    // and 54 KB of code, together more than the 64 KB instruction cache two compute units share, beside a third kernel); profiles/r06f_*.  SSLAM_LBD_SOBEL_MAIN=1: always behind.
    if (!sobelDone && !side.forked && nframes >= 1024 && !L->coreEvent && !getenv("SSLAM_LBD_SOBEL_MAIN")) {
        if ((rc = side_stream_ready())) return rc;
        SSLAM_HIP(hipEventRecord(L->nfaFork, st));
        SSLAM_HIP(hipStreamWaitEvent(L->nfaStream, L->nfaFork, 0));
        side.forked = true;
        launch_blur_sobel(L->nfaStream);
    }
    // the NFA stage: its kernels and launch forms live in lines_nfa.hip, a translation unit of its own (compiled with -mllvm -disable-machine-licm)
    if (nfaStreamed) {      // what the concurrent consumers left (nothing, unless they gave up waiting): the same kernel behind both, everything published, no waiting
        side.join();
        const int rc2 = sslam::launch_nfa_stream(L->ctx, st, ws, &P, sizeof(P), L->dLgam.as<double>(), L->dCl.as<uint8_t>(), L->clFrame, nfaStageOff, nframes, 16, 0, 0, NFA_STREAM_BLOCK, 1, "k_nfa_stream");
        if (rc2) return rc2;
    } else
    { const int rc = sslam::launch_nfa_stage(L->ctx, st, ws, &P, sizeof(P), L->dLgam.as<double>(), nframes); if (rc) return rc; }
    { sslam::ProfScope _ps(L->ctx, "k_keylines", st); hipLaunchKernelGGL(k_keylines, dim3(nframes), dim3(256), 0, st, ws, P, L->maxLines, d_kl, d_linefn, d_counts, cap); }
    // LBD: blur(5, 1) + Sobel fused (SSLAM_LBD_SOBEL=early: in the prologue) -> bands
    if (!sobelDone) launch_blur_sobel(st);
    side.join();
    {   // the walk's conversion form (lbd.h): images of up to 16 384 pixels a side; SSLAM_LBD_RPI=0 forces the previous form (A/B, tests)
        const bool rpiOff = getenv("SSLAM_LBD_RPI") && atoi(getenv("SSLAM_LBD_RPI")) == 0;
        const bool rpi = !rpiOff && P.w <= 16384 && P.h <= 16384;
        sslam::ProfScope _ps(L->ctx, "k_lbd", st);
        const dim3 grd(std::min(L->maxLines, cap), nframes);
        if (rpi) hipLaunchKernelGGL(k_lbd<true>, grd, dim3(64), 0, st, ws, P, d_kl, d_counts, d_ldesc, cap);
        else hipLaunchKernelGGL(k_lbd<false>, grd, dim3(64), 0, st, ws, P, d_kl, d_counts, d_ldesc, cap);
    }
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
    std::lock_guard<std::recursive_mutex> lk(L->ctx->mu);
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

extern "C" int sslam_lines_batch_status_dev(sslam_lines* L, int cap, int32_t* d_status4, void* stream_) {
    if (!L || !d_status4 || cap <= 0 || L->lastFrames <= 0) { set_error("sslam_lines_batch_status_dev: invalid arguments (or no batch yet)"); return SSLAM_ERR_INVALID; }
    SSLAM_HIP(hipSetDevice(L->ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : L->ctx->stream;
    hipLaunchKernelGGL(k_lines_status, dim3(1), dim3(256), 0, st, L->dWs.as<uint8_t>(), L->plan, L->lastFrames, L->maxLines, cap, d_status4);
    SSLAM_HIP(hipGetLastError());
    return SSLAM_OK;
}

extern "C" int sslam_lines_batch_status(sslam_lines* L, int cap, void* stream_, int* truncated_frames_out, int* unsupported_frames_out, int* first_frame_out) {
    if (!L) { set_error("sslam_lines_batch_status: null handle"); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::recursive_mutex> lk(L->ctx->mu);
    int rc;
    if ((rc = L->dCounts.ensure(16))) return rc;
    hipStream_t st = stream_ ? (hipStream_t)stream_ : L->ctx->stream;
    if ((rc = sslam_lines_batch_status_dev(L, cap, L->dCounts.as<int32_t>(), st))) return rc;
    int h[4] = {0, 0, 0, 0};
    SSLAM_HIP(hipMemcpyAsync(h, L->dCounts.p, sizeof(h), hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    if (truncated_frames_out) *truncated_frames_out = h[0];
    if (unsupported_frames_out) *unsupported_frames_out = h[2];
    if (first_frame_out) *first_frame_out = h[2] ? h[3] : h[0] ? h[1] : -1;
    if (h[2]) { set_error("sslam_lines_batch_status: %d frame(s) with more than %d candidate rectangles (first: frame %d)", h[2], MAX_SEG, h[3]); return SSLAM_ERR_UNSUPPORTED; }
    if (h[0]) { set_error("sslam_lines_batch_status: %d frame(s) hold more lines than the capacity %d (first: frame %d)", h[0], cap, h[1]); return SSLAM_ERR_CAPACITY; }
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

extern "C" int sslam_lines_set_core_event(sslam_lines* L, void* hip_event) {
    if (!L) return SSLAM_ERR_INVALID;
    L->coreEvent = (hipEvent_t)hip_event;
    return SSLAM_OK;
}

extern "C" int sslam_lines_set_core_gate(sslam_lines* L, void* wait_event, void* done_event) {
    if (!L) return SSLAM_ERR_INVALID;
    L->coreWait = (hipEvent_t)wait_event; L->coreDone = (hipEvent_t)done_event;
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

#ifdef SSLAM_TESTING      // everything from here to the matching #endif exists in libsslam_frontend_testing.so only (include/sslam_testing.h)
// counters of the cluster form's helpers for frame `frame` of the last call (lsd_cluster.h, ClCtl::stat; filled by builds with -DSSLAM_CL_CYCLES)
extern "C" int sslam_lines_debug_cluster(sslam_lines* L, int frame, long long* out8) {
    if (!L || frame < 0 || frame >= L->clSlots || !out8 || !L->dCl.p || !L->clFrame) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(L->ctx->device));
    SSLAM_HIP(hipDeviceSynchronize());
    ClCtl c;
    SSLAM_HIP(hipMemcpy(&c, L->dCl.as<uint8_t>() + (size_t)frame * L->clFrame, sizeof(c), hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i) out8[i] = c.stat[i];
    return SSLAM_OK;
}

// Self-test of the table-based exact division used in the NFA tail (tests/test_lines_gpu.py): returns the number of
// random (a, b) pairs, 1 <= a, b < n, whose quotient differs from the hardware IEEE division (must be 0).
namespace {
struct ScopedDev {            // device scratch of a self-test: freed on every return path
    void* p = nullptr;
    ~ScopedDev() { if (p) (void)hipFree(p); }
};
}  // namespace

extern "C" int sslam_selftest_exact_div(sslam_ctx* ctx, int n, long long pairs, long long* mismatches_out) {
    if (!ctx || n < 3 || pairs <= 0 || !mismatches_out) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(ctx->device));
    ScopedDev tabMem, badMem;
    SSLAM_HIP(hipMalloc(&tabMem.p, sizeof(double) * (2 * (size_t)n + 48)));
    SSLAM_HIP(hipMalloc(&badMem.p, sizeof(unsigned long long)));
    double* tab = (double*)tabMem.p; unsigned long long* bad = (unsigned long long*)badMem.p;
    SSLAM_HIP(hipMemset(bad, 0, sizeof(unsigned long long)));
    { const int rc = upload_nfa_tables(tab, n, ctx->stream); if (rc) return rc; }
    const int threads = 256 * 1024, iters = (int)((pairs + threads - 1) / threads);
    hipLaunchKernelGGL(k_selftest_div, dim3(1024), dim3(256), 0, ctx->stream, tab + n + 48, n, 0x1234567ull, iters, bad);
    unsigned long long h = 0;
    SSLAM_HIP(hipMemcpyAsync(&h, bad, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SSLAM_HIP(hipStreamSynchronize(ctx->stream));
    *mismatches_out = (long long)h;
    return SSLAM_OK;
}

// Self-test of the guarded fp32 early-exit test of the NFA tail (tests/test_lines_gpu.py): `samples` random inputs, half
// of them on the decision boundary.  disagree_out = decided cases whose verdict differs from the fp64 expression (must be
// 0); ambiguous_out = cases handed to the fp64 expression.
extern "C" int sslam_selftest_tail_test(sslam_ctx* ctx, long long samples, long long* disagree_out, long long* ambiguous_out) {
    if (!ctx || samples <= 0 || !disagree_out || !ambiguous_out) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(ctx->device));
    ScopedDev dMem;
    SSLAM_HIP(hipMalloc(&dMem.p, 2 * sizeof(unsigned long long)));
    unsigned long long* d = (unsigned long long*)dMem.p;
    SSLAM_HIP(hipMemset(d, 0, 2 * sizeof(unsigned long long)));
    const int threads = 256 * 1024, iters = (int)((samples + threads - 1) / threads);
    const double logNT = 5 * (std::log10(512.0) + std::log10(384.0)) / 2 + std::log10(11.0);
    hipLaunchKernelGGL(k_selftest_tail, dim3(1024), dim3(256), 0, ctx->stream, 0xABCDEF12345ull, iters, logNT, d);
    unsigned long long h[2] = {0, 0};
    SSLAM_HIP(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SSLAM_HIP(hipStreamSynchronize(ctx->stream));
    *disagree_out = (long long)h[0]; *ambiguous_out = (long long)h[1];
    return SSLAM_OK;
}

extern "C" int sslam_selftest_region_div(sslam_ctx* ctx, long long samples, long long* mismatches_out) {
    if (!ctx || samples <= 0 || !mismatches_out) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(ctx->device));
    ScopedDev dMem;
    SSLAM_HIP(hipMalloc(&dMem.p, 2 * sizeof(unsigned long long)));
    unsigned long long* d = (unsigned long long*)dMem.p;
    SSLAM_HIP(hipMemset(d, 0, 2 * sizeof(unsigned long long)));
    const int threads = 256 * 1024, iters = (int)((samples + threads - 1) / threads);
    hipLaunchKernelGGL(k_selftest_region_div, dim3(1024), dim3(256), 0, ctx->stream, 0x5EEDF00D1234ull, iters, d);
    unsigned long long h[2] = {0, 0};
    SSLAM_HIP(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SSLAM_HIP(hipStreamSynchronize(ctx->stream));
    mismatches_out[0] = (long long)h[0]; mismatches_out[1] = (long long)h[1];      // division, atan2
    return SSLAM_OK;
}

// sslam_selftest_sincos: sincos_0_2pi (common.h) against the library's sincos after the rounding to float, for every float in [0, 6.5]
namespace {
__global__ __launch_bounds__(256) void k_selftest_sincos(unsigned long long* __restrict__ bad) {
    unsigned long long nb = 0;
    const unsigned base = (blockIdx.x * 256u + threadIdx.x) << 6;
    for (unsigned i = 0; i < 64u; ++i) {
        const unsigned bits = base + i;
        if (bits > 0x40D00000u) break;                  // 6.5f
        const float x = __uint_as_float(bits);
        double s0, c0, s1, c1;
        sincos((double)x, &s0, &c0); sslam::sincos_0_2pi((double)x, s1, c1);
        if (__float_as_uint((float)s0) != __float_as_uint((float)s1) || __float_as_uint((float)c0) != __float_as_uint((float)c1)) ++nb;
    }
    if (nb) atomicAdd(bad, nb);
}
}
extern "C" int sslam_selftest_sincos(sslam_ctx* ctx, long long* mismatches_out) {
    if (!ctx || !mismatches_out) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(ctx->device));
    ScopedDev dMem;
    SSLAM_HIP(hipMalloc(&dMem.p, sizeof(unsigned long long)));
    unsigned long long* d = (unsigned long long*)dMem.p;
    SSLAM_HIP(hipMemset(d, 0, sizeof(unsigned long long)));
    const unsigned nthreads = (0x40D00000u >> 6) + 1;
    hipLaunchKernelGGL(k_selftest_sincos, dim3((nthreads + 255) / 256), dim3(256), 0, ctx->stream, d);
    unsigned long long h = 0;
    SSLAM_HIP(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SSLAM_HIP(hipStreamSynchronize(ctx->stream));
    *mismatches_out = (long long)h;
    return SSLAM_OK;
}

extern "C" int sslam_selftest_lsd_bin(sslam_ctx* ctx, int max_s, long long* mismatches_out) {
    if (!ctx || max_s < 0 || max_s >= (1 << 24) || !mismatches_out) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(ctx->device));
    ScopedDev dMem;
    SSLAM_HIP(hipMalloc(&dMem.p, sizeof(unsigned long long)));
    unsigned long long* d = (unsigned long long*)dMem.p;
    SSLAM_HIP(hipMemset(d, 0, sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_selftest_lsd_bin, dim3(2048), dim3(256), 0, ctx->stream, max_s, d);
    unsigned long long h = 0;
    SSLAM_HIP(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SSLAM_HIP(hipStreamSynchronize(ctx->stream));
    *mismatches_out = (long long)h;
    return SSLAM_OK;
}

extern "C" int sslam_selftest_lbd_round(sslam_ctx* ctx, long long* mismatches_out) {
    if (!ctx || !mismatches_out) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(ctx->device));
    ScopedDev dMem;
    SSLAM_HIP(hipMalloc(&dMem.p, 2 * sizeof(unsigned long long)));
    unsigned long long* d = (unsigned long long*)dMem.p;
    SSLAM_HIP(hipMemset(d, 0, 2 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_selftest_lbd_round, dim3(65536), dim3(256), 0, ctx->stream, d);      // 2^32 bit patterns, 256 per thread
    unsigned long long h[2] = {0, 0};
    SSLAM_HIP(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SSLAM_HIP(hipStreamSynchronize(ctx->stream));
    mismatches_out[0] = (long long)h[0]; mismatches_out[1] = (long long)h[1];      // against the previous form under the clamps, against roundf for x >= 0
    return SSLAM_OK;
}

// Issue-rate probe (profiles/README.md, round 3): every lane runs `iters` rounds of 16 independent VALU instructions of one kind
// (0: v_add_u32, 1: v_fma_f32, 2: v_add_f64, 3: v_bcnt_u32_b32) with 8 waves per SIMD resident, so that the rate is the pipe's, not a
// dependency chain's.  *ginst_per_s_out = wave-instructions per second over the whole chip (divide by SIMDs x clock for cycles per instruction).
namespace {
template <int KIND>
__global__ __launch_bounds__(256) void k_probe_valu(int iters, unsigned* __restrict__ sink) {
    unsigned a[16]; double d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x + i; d[i] = (double)(threadIdx.x + i); }
    const unsigned b = blockIdx.x | 1u; const float fb = 1.0000001f; const double db = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            else if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(fb));
            else if (KIND == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
            else asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += a[i] + (unsigned)d[i];
    if (acc == 0x12345u) *sink = acc;
}
}  // namespace
extern "C" int sslam_selftest_valu_rate(sslam_ctx* ctx, int kind, double* ginst_per_s_out) {
    if (!ctx || kind < 0 || kind > 3 || !ginst_per_s_out) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(ctx->device));
    ScopedDev sinkMem;
    SSLAM_HIP(hipMalloc(&sinkMem.p, 4));
    const int blocks = ctx->num_cus * 8, iters = 20000;      // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    SSLAM_HIP(hipEventCreate(&e0)); SSLAM_HIP(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {          // the first launch warms the clocks up
        (void)hipEventRecord(e0, ctx->stream);
        if (kind == 0) hipLaunchKernelGGL(k_probe_valu<0>, dim3(blocks), dim3(256), 0, ctx->stream, iters, (unsigned*)sinkMem.p);
        else if (kind == 1) hipLaunchKernelGGL(k_probe_valu<1>, dim3(blocks), dim3(256), 0, ctx->stream, iters, (unsigned*)sinkMem.p);
        else if (kind == 2) hipLaunchKernelGGL(k_probe_valu<2>, dim3(blocks), dim3(256), 0, ctx->stream, iters, (unsigned*)sinkMem.p);
        else hipLaunchKernelGGL(k_probe_valu<3>, dim3(blocks), dim3(256), 0, ctx->stream, iters, (unsigned*)sinkMem.p);
        (void)hipEventRecord(e1, ctx->stream);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); set_error("sslam_selftest_valu_rate: kernel failed"); return SSLAM_ERR_HIP; }
    }
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ginst_per_s_out = (double)blocks * 4.0 * iters * 16.0 / (ms * 1e-3) / 1e9;
    return SSLAM_OK;
}

// FETCH_SIZE calibration: reads `bytes` of a freshly allocated buffer once with 16 B/lane coalesced loads (mode 0) or
// issues bytes/16 scattered 16-B gathers over it (mode 1).  bytes_requested_out = 16 x loads issued.
extern "C" int sslam_selftest_fetch_probe(sslam_ctx* ctx, size_t bytes, int mode, long long* bytes_requested_out) {
    if (!ctx || bytes < (1u << 20) || (mode != 0 && mode != 1) || !bytes_requested_out) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(ctx->device));
    ScopedDev bufMem, sinkMem;
    SSLAM_HIP(hipMalloc(&bufMem.p, bytes));
    SSLAM_HIP(hipMalloc(&sinkMem.p, sizeof(float)));
    float4* buf = (float4*)bufMem.p; float* sink = (float*)sinkMem.p;
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
    return SSLAM_OK;
}

#endif      // SSLAM_TESTING

// Device-resident frame handle of the last sslam_lines_extract call (keylines + LBD descriptors), see sslam_frame_from_orb.
int sslam_frame_from_device(sslam_ctx* ctx, int kind, const void* d_feats, const uint8_t* d_desc, int n, const float bounds[4], sslam_frame** out);
extern "C" int sslam_frame_from_lines(sslam_lines* L, const float bounds[4], sslam_frame** out) {
    if (!L || !bounds || !out) { set_error("sslam_frame_from_lines: invalid arguments"); return SSLAM_ERR_INVALID; }
    if (L->lastN < 0) { set_error("sslam_frame_from_lines: no sslam_lines_extract call to snapshot"); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::recursive_mutex> lk(L->ctx->mu);
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

extern "C" sslam_ctx* sslam_lines_context(sslam_lines* L) { return L ? L->ctx : nullptr; }

#ifdef SSLAM_TESTING
// The host-evaluated nfa() tables exactly as upload_nfa_tables() sends them to the device (decision D8): lgam[0..n), then 16 x {log p,
// log(1-p), log10 p}, then 1/j.  Pure host code (no GPU needed): tests/test_oracle_cpu.py pins the bits against committed goldens, so a
// host libm that rounds log / sinh / pow differently is noticed before it can flip a borderline rectangle.
extern "C" int sslam_debug_nfa_tables(int n, double* out) {
    if (n < 2 || !out) return SSLAM_ERR_INVALID;
    for (int j = 0; j < n; ++j) out[j] = j >= 1 ? host_log_gamma((double)j) : 0.0;
    for (int j = 0; j < 16; ++j) { const double pp = std::ldexp(0.125, -j); out[n + 3 * j] = std::log(pp); out[n + 3 * j + 1] = std::log(1.0 - pp); out[n + 3 * j + 2] = std::log10(pp); }
    for (int j = 0; j < n; ++j) out[(size_t)n + 48 + j] = j >= 1 ? 1.0 / (double)j : 0.0;
    return SSLAM_OK;
}
#endif      // SSLAM_TESTING
