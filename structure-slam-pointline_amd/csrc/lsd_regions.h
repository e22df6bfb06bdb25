// LSD sequential core: region growing, region2rect, refine / reduce_region_radius, one persistent single-wave workgroup per frame.
// Part of lines.hip (included there, inside its anonymous namespace: one translation unit, so device helpers are shared
// without relocatable device code).  Not a standalone header.
#pragma once

// region point list: first QCAP entries in LDS, the rest in global memory.  entry = x | y<<16
struct RegQ {
    unsigned* lds; unsigned* glb;
    __device__ __forceinline__ unsigned get(int i) const { return i < QCAP ? lds[i] : glb[i]; }
    // n = current list length (wave-uniform): a list that fits in LDS is read with ds_read under a scalar branch; the per-lane select of
    // get() makes the compiler build a generic pointer and issue a flat load, which takes the vector-memory path even for LDS addresses
    __device__ __forceinline__ unsigned get_n(int i, int n) const {
        if (n <= QCAP) { unsigned v = lds[i]; asm volatile("" : "+v"(v)); return v; }      // the empty asm keeps the two loads from being merged into one flat load
        return get(i);
    }
    __device__ __forceinline__ void set(int i, unsigned v) const { if (i < QCAP) lds[i] = v; else glb[i] = v; }
};
// the per-pixel planes of one frame (lsd_plan.h): T = angle | used in the sign bit (8 x 4 tiles), Cs = {cos, sin} (4 x 4 tiles), S = |g|^2 (row-major)
struct Planes {
    float* T; const float2* Cs; const int* S; int tW, cW;
    __device__ __forceinline__ unsigned* Tb() const { return (unsigned*)T; }
    // element i of a plane through a 32-bit byte offset: the planes' base pointers are wave-uniform, so the load takes the scalar-base +
    // 32-bit-offset form (global_load ... v_off, s[base]) instead of a 64-bit address computed per lane (a frame's planes are < 2^31 bytes)
    __device__ __forceinline__ float ldT(int i) const { return *(const float*)((const char*)T + ((unsigned)i << 2)); }
    __device__ __forceinline__ void stT(int i, float v) const { *(float*)((char*)T + ((unsigned)i << 2)) = v; }
    __device__ __forceinline__ float2 ldCs(int i) const { return *(const float2*)((const char*)Cs + ((unsigned)i << CS_SHIFT)); }
    // (coordinates and pitch are < 2^16: v_mad_u32_u24, one full-rate instruction; v_mul_lo_u32 is quarter rate)
    __device__ __forceinline__ int ti(int x, int y) const { return (int)(__umul24((unsigned)y, (unsigned)tW) + (unsigned)x); }
    __device__ __forceinline__ int ti(unsigned e) const { return (int)(__umul24(e >> 16, (unsigned)tW) + (e & 0xFFFFu)); }

};
__device__ __forceinline__ bool t_free(float t) { return __float_as_int(t) >= 0; }      // defined and not part of a region
__device__ __forceinline__ float t_used(float t) { return __int_as_float(__float_as_int(t) | (int)USED_BIT); }
// helper waves (cluster form, lsd_cluster.h): a helper's private marks: a bit TORUS in LDS whatever the frame size (256 x 256 bits = 8 KB here; TorusWide for three helpers per workgroup).  A region that stays within +-126 pixels of its seed
// (tested neighbours: +-127) cannot alias on it; a helper abandons a region that reaches further and the main wave grows that one itself.
constexpr int MW_BM_WORDS = 256 * 256 / 32, MW_REACH = 126;
__device__ __forceinline__ int mw_bit(int x, int y) { return ((y & 255) << 8) | (x & 255); }
// Where a growing region's marks live (template argument of region_grow_w / rect_refine / remove_far_points_lds):
//   MARK_MAP   the pixel map itself (single-wave kernel);
//   MARK_SPEC  a helper's private bitmap: lists A / B / F are all kept for the later validation, growth gives up beyond capN points or
//              outside the bitmap's reach;
//   MARK_PRIV  a private bitmap that covers the whole frame (main wave of the cluster form): lists are handled as with MARK_MAP, the pixel
//              map is written once, when the outcome of the seed is committed -- pixels are never released in it.
// G = the bitmap's geometry (a torus of XMASK+1 x YMASK+1 bits; REACH = how far from its seed a region may go before bits could alias).
enum { MARK_MAP = 0, MARK_SPEC = 1, MARK_PRIV = 2 };
struct TorusHelper { static constexpr int XMASK = 255, YMASK = 255, YSHIFT = 8, REACH = MW_REACH, WORDS = MW_BM_WORDS; };
struct TorusWide { static constexpr int XMASK = 511, YMASK = 511, YSHIFT = 9, REACH = 254, WORDS = 512 * 512 / 32; };      // cluster form helpers: three per workgroup, 32 KB each
struct TorusFrame { static constexpr int XMASK = 1023, YMASK = 511, YSHIFT = 10, REACH = 1 << 20, WORDS = 1024 * 512 / 32; };      // no aliasing for frames up to 1024 x 512
// the same for frames up to 2048 x 1024 (the scaled 1280 x 960 of BASELINE configs[3]): 256 KB, kept in GLOBAL memory -- the marks are set and
// cleared with atomics (executed in the L2), so they are read with L1-bypassing loads
struct TorusGlobal { static constexpr int XMASK = 2047, YMASK = 1023, YSHIFT = 11, REACH = 1 << 20, WORDS = 2048 * 1024 / 32; static constexpr bool GLOBAL = true; };
template <class G, class = void> struct bm_is_global { static constexpr bool value = false; };
template <class G> struct bm_is_global<G, decltype((void)G::GLOBAL)> { static constexpr bool value = G::GLOBAL; };
template <class G> __device__ __forceinline__ unsigned bm_word(const unsigned* bm, int w) {
    if (bm_is_global<G>::value) return __hip_atomic_load(bm + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return bm[w];
}
template <class G> __device__ __forceinline__ int bm_bit(int x, int y) { return ((y & G::YMASK) << G::YSHIFT) | (x & G::XMASK); }
__device__ __forceinline__ double readlane_d(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// a / b for 0 <= a <= b, b in [2^-60, 2^60], a == 0 or a >= 2^-80 b: the hardware's IEEE division sequence (v_div_scale, v_rcp, three
// fused refinement steps, v_div_fmas, v_div_fixup) without the scaling and the special-case fix-up, which are the identity on this domain
// -- the same rcp and the same fused operations, hence the same correctly rounded quotient (sslam_selftest_region_div compares it with
// the `/` operator bit for bit).  The direction sums of a region are sums of float cos/sin values: |component| >= 6e-17 or exactly 0.
__device__ __forceinline__ float div_unit_range(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
    float q = a * r;
    q = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}

// cv::fastAtan2 as in common.h (branch-free form), with the division above
__device__ __forceinline__ float fast_atan2_deg_unit(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    const float c = div_unit_range(fminf(ax, ay), __fadd_rn(fmaxf(ax, ay), eps));      // the smaller magnitude over the larger one, either way
    const float c2 = __fmul_rn(c, c);
    float a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    if (ax < ay) a = __fsub_rn(90.f, a);
    // "if (x < 0) a = 180 - a; if (y < 0) a = 360 - a" without compares: (sign ? K : 0) - a is K - a or -a, and K - a >= 0.  The sums are
    // never -0 (they start at a nonzero cos or a +0 sin and x + (-x) rounds to +0), so the sign bit is exactly "x < 0".
    a = fabsf(__fsub_rn(__int_as_float((__float_as_int(x) >> 31) & 0x43340000), a));
    a = fabsf(__fsub_rn(__int_as_float((__float_as_int(y) >> 31) & 0x43b40000), a));
    return a;
}

#ifdef SSLAM_TESTING
// selftest: the trimmed division and the branch-free atan2 against the `/` operator and fast_atan2_deg on direction sums of every magnitude
// a region can produce (components from 6e-17 -- cos of 90.0 degrees in float -- up to 2^20 pixels, exact zeros, equal components)
__global__ void k_selftest_region_div(unsigned long long seed, int iters, unsigned long long* __restrict__ out) {
    unsigned long long x = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (float)((double)(x >> 11) * 0x1p-53); };
    unsigned long long bad = 0, bad2 = 0;
    for (int it = 0; it < iters; ++it) {
        const float ex = exp2f(-54.f + 74.f * rnd()), ey = exp2f(-54.f + 74.f * rnd());
        float sx = (1.f + rnd()) * ex, sy = (1.f + rnd()) * ey;
        const int mode = (int)(rnd() * 16.f);
        if (mode == 0) sy = 0.f; else if (mode == 1) sx = sy; else if (mode == 2) sx = 6.123234e-17f * (float)(1 + (int)(rnd() * 4000.f));
        else if (mode == 3) { sx = (float)(int)(rnd() * 100000.f); sy = (float)(int)(rnd() * 100000.f); if (sx == 0.f && sy == 0.f) sx = 1.f; }
        if ((x & 1) && sx != 0.f) sx = -sx;           // no -0: a region's sums cannot produce one (see fast_atan2_deg_unit)
        if ((x & 2) && sy != 0.f) sy = -sy;
        if (sx == 0.f && sy == 0.f) continue;
        const float eps = (float)2.2204460492503131e-16;
        const float ax = fabsf(sx), ay = fabsf(sy), mn = fminf(ax, ay), mx = __fadd_rn(fmaxf(ax, ay), eps);
        if (__float_as_int(div_unit_range(mn, mx)) != __float_as_int(__fdiv_rn(mn, mx))) ++bad;
        if (__float_as_int(fast_atan2_deg_unit(sy, sx)) != __float_as_int(fast_atan2_deg<false>(sy, sx))) ++bad2;
    }
    if (bad) atomicAdd(out, bad);
    if (bad2) atomicAdd(out + 1, bad2);
}
#endif

// LineSegmentDetectorImpl::region_grow by one wave.  Eight queue entries are staged at a time (8
// lanes each: the 3x3 neighbourhood in row-major order without its centre), so one global-load round
// trip serves up to eight points.  Lane order == the reference's visiting order, and the region angle only
// changes when a pixel is accepted, so ONE ballot over all staged lanes finds the next accepted
// pixel exactly as the sequential scan would; lanes before it are consumed, lanes after it are
// re-tested against the updated angle.
// LAT selects the flavour: v_readlane broadcasts, deferred stores and straight-line staging for a lone wave (single-frame latency),
// LDS permutes, immediate stores and masked staging loads with six waves per SIMD (throughput); each choice measured both ways.
// WIDE = false needs prec < pi/2: then "fold at 3pi/2, compare" is "|d| <= prec or ||d| - 2pi| <= prec" on the very same doubles (for
// |d| in (pi, 3pi/2] both forms say no, above 3pi/2 the first clause cannot hold) -- two compares instead of compare + select + compare.
//
// Drift-bounded decisions (SSLAM_LSD_DRIFT, tolerances up to 0.45 rad): the reference recomputes reg_angle = fastAtan2(sumdy, sumdx) after
// EVERY accepted pixel, 27 dependent instructions that one whole wave executes for one pixel.  The angle is a pure function of the two
// sums, so it only has to exist when a decision needs it.  Between two evaluations the wave keeps thetaRef (the last exact angle) and eps,
// a rigorous bound on how far the true direction of the sums can have turned since: adding a unit vector u that makes the angle alpha with
// S turns S by asin(|u| sin(alpha) / |S + u|) <= 1.006 alpha / max(|S'x|, |S'y|) (alpha <= 22.6 degrees because u was just accepted, and
// |S'| >= 2.5 is required), and alpha <= dc(u) + eps + E, E = 0.0096 degrees being the largest error of the fastAtan2 polynomial (measured
// over every float in [0, 1]).  A candidate whose circular distance dc to thetaRef is <= prec - eps - slack is accepted by the reference
// whatever the exact angle is, one with dc >= prec + eps + slack is rejected; only a candidate inside that band forces the exact angle
// (and resets eps).  The sums themselves are still added pixel by pixel in the reference's order, so the exact angle, whenever it is
// evaluated, is bit-identical to the reference's.  -DSSLAM_LSD_DRIFT_VERIFY cross-checks every shortcut decision against the exact test
// and counts disagreements in Misc::cyc[6] (cyc[7] = shortcuts << 32 | decisions; tools/lsd_drift_verify.py).
#ifndef SSLAM_LSD_DRIFT
#define SSLAM_LSD_DRIFT 1
#endif
// SPEC (helper waves of the cluster form, lsd_cluster.h): a helper wave grows a region AHEAD of the frame's main wave.  It never writes the pixel map: the
// pixels it takes are marked in its own bitmap `bm` (LDS), and it gives up (returns -n) when the list would outgrow `capN` points.
template <bool LAT, bool WIDE, int MARK = MARK_MAP, class G = TorusHelper>
__device__ int region_grow_w(int seedX, int seedY, float seedDeg, float seedCos, float seedSin, int sw, int sh, const Planes& pl, const RegQ& rq,
                             double prec, double& regAngleOut, long long* __restrict__ verifyCnt, unsigned* __restrict__ bm = nullptr, int capN = 0) {
#ifndef SSLAM_LSD_READLANE
#define SSLAM_LSD_READLANE 0
#endif
    static_assert(MARK == MARK_MAP || LAT, "private marks are written by the lone-wave flavour only");
    constexpr bool RL = LAT || SSLAM_LSD_READLANE;      // broadcasts of the accepted lane: v_readlane for the lone wave, LDS permutes with six waves per SIMD (measured: 35.4 vs 37.2 ms)
    constexpr bool DRIFT = !WIDE && SSLAM_LSD_DRIFT;
    const int lane = threadIdx.x & 63;
    int n = 1;
    double regAngle = (double)seedDeg * DEG2RAD;          // the seed's level-line angle, cos / sin: evaluated lane-parallel for a whole chunk of seed candidates (k_lsd_regions)
    float sumdx = seedCos, sumdy = seedSin;
    if (lane == 0) {
        rq.set(0, (unsigned)seedX | ((unsigned)seedY << 16));
        if (MARK) { const int bi = bm_bit<G>(seedX, seedY); atomicOr(&bm[bi >> 5], 1u << (bi & 31)); } else pl.stT(pl.ti(seedX, seedY), t_used(seedDeg));
    }
    const int g = lane >> 3, k8 = lane & 7;             // group (queue slot) and neighbour slot (centre skipped)
    const int k = k8 + (k8 >= 4 ? 1 : 0);                // row-major 3x3 position 0..8 without 4
    const int dy = k / 3 - 1, dx = k - (k / 3) * 3 - 1;
    // drift state (degrees): thetaRef = the last exact angle, band = eps + slack with eps the bound on the turn of the sums since thetaRef,
    // fresh = regAngle is the exact angle of the current sums
    const float precDeg = (float)(prec * (180.0 / kPI));
    const float minM = prec <= 0.45 ? 2.5f : 3.0e38f;    // wider tolerances (refine() can ask for them) always take the exact path
    constexpr float DRIFT_K = 1.007f, DRIFT_E = 0.011f, DRIFT_ADD = 2e-5f, DRIFT_SLACK = 0.05f;
    float thetaRef = seedDeg, band = DRIFT_SLACK;
    bool fresh = true;
    int i = 0;
#ifdef SSLAM_LSD_CYCLES
    long long* cycStage = verifyCnt - 1;      // Misc::cyc[5..]: staging wait, accept loops, stagings (the NFA statistics of SSLAM_LSD_STATS use the same slots)
#endif
    while (i < n) {
        if (MARK == MARK_SPEC && n + 64 > capN) return -n;
        const int np = min(8, n - i);
        // Lone wave: straight-line staging -- every lane loads (slots past the staged entries re-read the last entry, coordinates are
        // clamped into the image) and the three conditions (slot staged, neighbour inside the image, pixel neither NOTDEF nor USED) meet as
        // lane masks: fewer instructions and no branches on the one wave's path (14.2 -> 13.5 ms per frame).  With six waves per SIMD the
        // extra gather lanes cost more than the branches save (32.3 -> 32.8 ms per 6144 frames), so that flavour keeps the masked loads.
        int nidx = -1, xx = 0, yy = 0; float tv = NOTDEF_F; float2 cs = make_float2(0.f, 0.f);      // nidx: the pixel's element in the T plane (its identity)
        unsigned long long candM;
#ifdef SSLAM_LSD_CYCLES
        const long long tS0 = __builtin_readcyclecounter();
#endif
        if (LAT) {
            const unsigned e = rq.get_n(min(i + g, n - 1), n);
            xx = (int)(e & 0xFFFF) + dx; yy = (int)(e >> 16) + dy;
            const int cx = min(max(xx, 0), sw - 1), cy = min(max(yy, 0), sh - 1);
            nidx = pl.ti(cx, cy);
            tv = pl.ldT(nidx);                           // sign bit set: NOTDEF or already USED
            cs = pl.ldCs(nidx);                            // (same row pitch: one index serves both planes)
            candM = __builtin_amdgcn_ballot_w64(t_free(tv)) & __builtin_amdgcn_ballot_w64((unsigned)xx < (unsigned)sw) &
                    __builtin_amdgcn_ballot_w64((unsigned)yy < (unsigned)sh) & (np == 8 ? ~0ull : ((1ull << (np * 8)) - 1));
            if (MARK) { const int bi = bm_bit<G>(xx, yy); candM &= ~__builtin_amdgcn_ballot_w64((bm_word<G>(bm, bi >> 5) >> (bi & 31)) & 1u); }      // taken by this wave itself
        } else {
            bool cand = false;
            if (g < np) {
                const unsigned e = rq.get_n(i + g, n);
                xx = (int)(e & 0xFFFF) + dx; yy = (int)(e >> 16) + dy;
                if (xx >= 0 && yy >= 0 && xx < sw && yy < sh) {
                    nidx = pl.ti(xx, yy);      // y, pitch, x < 2^16: one full-rate instruction (v_mul_lo_u32 is quarter rate)
                    tv = pl.ldT(nidx);
                    cs = pl.ldCs(nidx);
                    cand = t_free(tv);
                }
            }
            candM = __builtin_amdgcn_ballot_w64(cand);
        }
#ifdef SSLAM_LSD_CYCLES
        const long long tS1 = __builtin_readcyclecounter();
#endif
        const int nBefore = n;
        unsigned long long accMask = 0;                    // lanes accepted from this staging, in lane (= acceptance) order
        // the live candidates and "lanes after the last accepted one" are wave-uniform 64-bit masks: scalar updates, no VALU
        auto aligned_mask = [&]() -> unsigned long long {
            if (WIDE) return __builtin_amdgcn_ballot_w64(is_aligned_val(tv, regAngle, prec));
            const double d = fabs(regAngle - (double)tv * DEG2RAD);
            return __builtin_amdgcn_ballot_w64(d <= prec) | __builtin_amdgcn_ballot_w64(fabs(d - M_2PI_) <= prec);
        };
        unsigned long long live = candM;                      // candidates that are still to be decided: the lanes above the last accepted one
        if (!DRIFT) {
            unsigned long long m = aligned_mask() & live;
            while (m) {                                          // bottom-tested: the next mask is computed right after the angle update
                const int sel = __ffsll((long long)m) - 1;       // wave-uniform
                const int selIdx = RL ? __builtin_amdgcn_readlane(nidx, sel) : __builtin_amdgcn_ds_bpermute(sel << 2, nidx);
                if (LAT) accMask |= 1ull << sel;               // lone wave: the used-map / queue stores follow the loop, all lanes at once
                else if (lane == sel) {                        // LDS slot QCAP is a sink, so the common case has no branch around the store
                    const unsigned v = (unsigned)xx | ((unsigned)yy << 16);
                    pl.stT(nidx, t_used(tv)); rq.lds[min(n, QCAP)] = v;
                    if (n >= QCAP) rq.glb[n] = v;
                }
                ++n;
                sumdx = __fadd_rn(sumdx, __int_as_float(RL ? __builtin_amdgcn_readlane(__float_as_int(cs.x), sel) : __builtin_amdgcn_ds_bpermute(sel << 2, __float_as_int(cs.x))));
                sumdy = __fadd_rn(sumdy, __int_as_float(RL ? __builtin_amdgcn_readlane(__float_as_int(cs.y), sel) : __builtin_amdgcn_ds_bpermute(sel << 2, __float_as_int(cs.y))));
                regAngle = (double)fast_atan2_deg_unit(sumdy, sumdx) * DEG2RAD;
                candM &= ~__builtin_amdgcn_ballot_w64(nidx == selIdx);      // the accepted pixel is now USED for every later visitor
                m = aligned_mask() & candM & (~1ull << sel);                // only lanes above sel
            }
        } else {
            // per lane and per reference angle: which side of the tolerance the candidate is on, and how far from it
            float dc, u; unsigned long long side;
            auto classify = [&]() {
                const float d = fabsf(__fsub_rn(thetaRef, tv));
                dc = fminf(d, __fsub_rn(360.f, d));                          // circular distance of the lane's level-line angle to thetaRef
                u = fabsf(__fsub_rn(dc, precDeg));
                side = __builtin_amdgcn_ballot_w64(dc <= precDeg);
            };
            classify();
            while (live) {
                const unsigned long long unc = __builtin_amdgcn_ballot_w64(u < band);       // the exact angle could decide these either way
                const unsigned long long cm = (side | unc) & live;                          // everything that is not certainly rejected
#ifdef SSLAM_LSD_DRIFT_VERIFY
                {
                    const double ra = (double)fast_atan2_deg_unit(sumdy, sumdx) * DEG2RAD;
                    const double d = fabs(ra - (double)tv * DEG2RAD);
                    const unsigned long long mx = (__builtin_amdgcn_ballot_w64(d <= prec) | __builtin_amdgcn_ballot_w64(fabs(d - M_2PI_) <= prec)) & live;
                    const int selX = mx ? __ffsll((long long)mx) - 1 : -1, selC = cm ? __ffsll((long long)cm) - 1 : -1;
                    const bool decided = selC < 0 || !((unc >> selC) & 1ull);
                    if (decided && selC != selX && lane == 0 && verifyCnt) atomicAdd((unsigned long long*)verifyCnt, 1ull);
                    if (lane == 0 && verifyCnt) atomicAdd((unsigned long long*)verifyCnt + 1, decided ? 0x100000001ull : 1ull);   // lo: decisions, hi: shortcuts
                }
#endif
                if (!cm) break;
                int sel = __ffsll((long long)cm) - 1;             // wave-uniform
                if ((unc >> sel) & 1ull) {                       // inside the band: this decision needs the exact angle of the current sums
                    if (!fresh) {
                        const float a = fast_atan2_deg_unit(sumdy, sumdx);
                        regAngle = (double)a * DEG2RAD; thetaRef = a; band = DRIFT_SLACK; fresh = true;
                        classify();
                    }
                    const unsigned long long m = aligned_mask() & live;
                    if (!m) break;
                    sel = __ffsll((long long)m) - 1;
                }
                const int selIdx = RL ? __builtin_amdgcn_readlane(nidx, sel) : __builtin_amdgcn_ds_bpermute(sel << 2, nidx);
                if (LAT) accMask |= 1ull << sel;
                else if (lane == sel) {
                    const unsigned v = (unsigned)xx | ((unsigned)yy << 16);
                    pl.stT(nidx, t_used(tv)); rq.lds[min(n, QCAP)] = v;
                    if (n >= QCAP) rq.glb[n] = v;
                }
                ++n;
                const float dcsEarly = __int_as_float(RL ? __builtin_amdgcn_readlane(__float_as_int(dc), sel) : __builtin_amdgcn_ds_bpermute(sel << 2, __float_as_int(dc)));
                sumdx = __fadd_rn(sumdx, __int_as_float(RL ? __builtin_amdgcn_readlane(__float_as_int(cs.x), sel) : __builtin_amdgcn_ds_bpermute(sel << 2, __float_as_int(cs.x))));
                sumdy = __fadd_rn(sumdy, __int_as_float(RL ? __builtin_amdgcn_readlane(__float_as_int(cs.y), sel) : __builtin_amdgcn_ds_bpermute(sel << 2, __float_as_int(cs.y))));
                {   // the sums turned by at most K (dc(sel) + eps + E) / max(|S'x|, |S'y|) + ADD degrees (eps + E = band - slack + E)
                    const float dcs = dcsEarly;
                    const float M = fmaxf(fabsf(sumdx), fabsf(sumdy));
                    const float aK = __builtin_fmaf(__fadd_rn(dcs, band), DRIFT_K, (DRIFT_E - DRIFT_SLACK) * DRIFT_K);
                    const float r = M < minM ? 1.0e9f : __builtin_amdgcn_rcpf(M);
                    band = __fadd_rn(__builtin_fmaf(aK, r, band), DRIFT_ADD);
                    fresh = false;
                }
                candM &= ~__builtin_amdgcn_ballot_w64(nidx == selIdx);      // the accepted pixel is now USED for every later visitor
                live = candM & (~1ull << sel);                              // only lanes above sel
            }
        }
        if (LAT && accMask != 0 && ((accMask >> lane) & 1ull)) {
            if (MARK) { const int bi = bm_bit<G>(xx, yy); atomicOr(&bm[bi >> 5], 1u << (bi & 31)); } else pl.stT(nidx, t_used(tv));
            rq.set(nBefore + mbcnt(accMask), (unsigned)xx | ((unsigned)yy << 16));
        }
        if (MARK == MARK_SPEC && __builtin_amdgcn_ballot_w64(((accMask >> lane) & 1ull) && (abs(xx - seedX) > G::REACH || abs(yy - seedY) > G::REACH))) return -n;      // leaves the torus
#ifdef SSLAM_LSD_CYCLES
        if (verifyCnt) { const long long tS2 = __builtin_readcyclecounter(); cycStage[0] += tS1 - tS0; cycStage[1] += tS2 - tS1; cycStage[2] += 1; }
#endif
        i += np;
    }
    if (DRIFT && !fresh) regAngle = (double)fast_atan2_deg_unit(sumdy, sumdx) * DEG2RAD;
    regAngleOut = regAngle;
    return n;
}

template <bool LAT, int MARK = MARK_MAP, class G = TorusHelper>
__device__ __forceinline__ int region_grow_m(int seedX, int seedY, float seedDeg, float seedCos, float seedSin, int sw, int sh, const Planes& pl, const RegQ& rq,
                                             double prec, double& regAngleOut, long long* __restrict__ verifyCnt, unsigned* __restrict__ bm = nullptr) {
    // the first growth runs at 22.5 degrees; refine()'s tolerance (two standard deviations of the angles) is normally smaller still
    if (prec < 1.5) return region_grow_w<LAT, false, MARK, G>(seedX, seedY, seedDeg, seedCos, seedSin, sw, sh, pl, rq, prec, regAngleOut, verifyCnt, bm, 0);
    return region_grow_w<LAT, true, MARK, G>(seedX, seedY, seedDeg, seedCos, seedSin, sw, sh, pl, rq, prec, regAngleOut, verifyCnt, bm, 0);
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
__device__ void region2rect_m(const RegQ& rq, int n, int sw, const int* __restrict__ S, double regAngle, double prec, double p, RectD& rec, double* __restrict__ red) {
    const int lane = threadIdx.x & 63;
    OrdSum S1; S1.acc = 0;
    double wgt0 = 0; int px0 = 0, py0 = 0;                  // the first 64 points stay in registers for the second pass (most regions are that short)
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        double fx = 0, fy = 0, wgt = 0;
        if (i < n) {
            const unsigned e = rq.get_n(i, n);
            const int px = e & 0xFFFF, py = e >> 16;
            wgt = sqrt((double)*(const int*)((const char*)S + ((__umul24((unsigned)py, (unsigned)sw) + (unsigned)px) << S_SHIFT)) / 4.0);
            fx = (double)px * wgt; fy = (double)py * wgt;
            if (base == 0) { wgt0 = wgt; px0 = px; py0 = py; }
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
            int px = px0, py = py0; double wgt = wgt0;
            if (base != 0) {
                const unsigned e = rq.get_n(i, n);
                px = e & 0xFFFF; py = e >> 16;
                wgt = sqrt((double)*(const int*)((const char*)S + ((__umul24((unsigned)py, (unsigned)sw) + (unsigned)px) << S_SHIFT)) / 4.0);
            }
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
        const unsigned e = rq.get_n(i, n);
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

// reduce_region_radius's removal pass for a point list that lives in LDS (n <= QCAP).  The reference walks the list once, overwrites a
// far point with the LAST point and tests that one again; that walk is a two-pointer partition: with K points kept, the holes below K
// receive -- in ascending order -- the kept points above K in descending order (checked against the sequential walk on 2*10^5 random
// keep patterns before it was written down here, and by the parity tests since).  n/64 wave steps instead of n.
// scratch: 1536 bytes of LDS (keep masks of <= 16 chunks, then the hole positions as u16: holes <= min(K, n - K) <= 512).
template <int MARK = MARK_MAP, class G = TorusHelper>
__device__ int remove_far_points_lds(unsigned* __restrict__ q, int n, double xc, double yc, double radSq, const Planes& pl,
                                     void* scratch, unsigned* __restrict__ bm = nullptr) {
    const int lane = threadIdx.x & 63;
    unsigned long long* km = (unsigned long long*)scratch;
    unsigned short* holeIdx = (unsigned short*)(km + 16);
    const int nch = (n + 63) >> 6;
    int K = 0;
    for (int c = 0; c < nch; ++c) {
        const int i = c * 64 + lane;
        bool keep = false;
        if (i < n) {
            const unsigned e = q[i];
            const int px = e & 0xFFFF, py = e >> 16;
            const double d2 = ((double)px - xc) * ((double)px - xc) + ((double)py - yc) * ((double)py - yc);
            keep = !(d2 > radSq);
            if (!keep) {                                                          // NOTUSED again
                if (MARK) { const int bi = bm_bit<G>(px, py); atomicAnd(&bm[bi >> 5], ~(1u << (bi & 31))); }
                else { unsigned* t = pl.Tb() + pl.ti(px, py); *t &= ~USED_BIT; }
            }
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        if (lane == 0) km[c] = m;
        K += __popcll(m);
    }
    int run = 0;
    for (int c = 0; c * 64 < K; ++c) {
        const int i = c * 64 + lane;
        const bool hole = i < K && !((km[c] >> lane) & 1ull);
        const unsigned long long hm = __builtin_amdgcn_ballot_w64(hole);
        if (hole) holeIdx[run + mbcnt(hm)] = (unsigned short)i;
        run += __popcll(hm);
    }
    int runB = 0;
    for (int c = nch - 1; c >= 0 && c * 64 + 63 >= K; --c) {
        const int j = c * 64 + lane;
        const bool bk = j >= K && ((km[c] >> lane) & 1ull);                        // bits beyond n are clear
        const unsigned long long bm = __builtin_amdgcn_ballot_w64(bk);
        if (bk) q[holeIdx[runB + __popcll((bm >> lane) >> 1)]] = q[j];            // rank counted from the back
        runB += __popcll(bm);
    }
    return K;
}

__device__ __forceinline__ double dist_d(double x1, double y1, double x2, double y2) {
    return sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1));
}

// ------------------------------------------------------------------ records shared with the cluster form (lsd_cluster.h)
// (Rounds 2-4 also shipped a multi-wave form of this kernel -- one main wave + helper waves inside ONE workgroup, results through LDS: DESIGN history, section 5c.  The
// cluster form superseded it for calls of up to 64 frames and the lone-wave flavour is as fast from 65 frames on, so it was removed in round 5; the record a helper
// publishes per seed and the validation rules (b) / (c) live on in lsd_cluster.h.)
// A published seed.  Lists in the helper's ring from `off`: A = the region as first grown (nA points); if refine() ran, B = the region
// re-grown at the refined tolerance (nB), and if reduce_region_radius ran, F = what it left of B (nF).  The pixels that end up USED are the
// last list's; A and B are what the helper accepted on the way, i.e. what must still be unused for the result to stand.
constexpr int MW_REFINED = 1, MW_REDUCED = 2, MW_EMIT = 4;
struct MwRes {                      // 20 bytes; lo = x0 | y0 << 16, hi = x1 | y1 << 16: box of A and B; the rectangle (MW_EMIT) follows the lists
    unsigned w0, w1, w2, lo, hi;    // lane | flags << 8 | (startSeq & 0xFFFF) << 16,  off | nA << 16,  nB | nF << 16
    __device__ __forceinline__ int lane() const { return w0 & 0xFF; }
    __device__ __forceinline__ int flags() const { return (w0 >> 8) & 0xFF; }
    __device__ __forceinline__ int startSeq() const { return w0 >> 16; }
    __device__ __forceinline__ int off() const { return w1 & 0xFFFF; }
    __device__ __forceinline__ int nA() const { return w1 >> 16; }
    __device__ __forceinline__ int nB() const { return w2 & 0xFFFF; }
    __device__ __forceinline__ int nF() const { return w2 >> 16; }
};
struct SpecLists { unsigned* bm; unsigned* free; int cap; int nB; bool reduced, gaveUp; };      // helper side of rect_refine
__device__ __forceinline__ int lds_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int lds_cas_uniform(int* p, int expect, int want, int lane) {      // one CAS per wave, result broadcast
    int got = expect;
    if (lane == 0) got = atomicCAS(p, expect, want);
    return __builtin_amdgcn_readfirstlane(got);
}
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b) { typedef unsigned short u16x2 __attribute__((ext_vector_type(2))); return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b))); }
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) { typedef unsigned short u16x2 __attribute__((ext_vector_type(2))); return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b))); }
// bounding box of a point list (entries x | y << 16 are already packed u16 pairs)
__device__ __forceinline__ void list_bbox(const unsigned* __restrict__ lst, int n, int lane, unsigned& lo, unsigned& hi) {
    unsigned mn = 0xFFFFFFFFu, mx = 0u;
    for (int i = lane; i < n; i += 64) { const unsigned e = lst[i]; mn = pk_min_u16(mn, e); mx = pk_max_u16(mx, e); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = pk_min_u16(mn, (unsigned)__shfl_xor((int)mn, o, 64)); mx = pk_max_u16(mx, (unsigned)__shfl_xor((int)mx, o, 64)); }
    lo = mn; hi = mx;
}
__device__ __forceinline__ bool boxes_meet(unsigned lo, unsigned hi, unsigned elo, unsigned ehi, int grow) {
    const int x0 = (int)(lo & 0xFFFF) - grow, y0 = (int)(lo >> 16) - grow, x1 = (int)(hi & 0xFFFF) + grow, y1 = (int)(hi >> 16) + grow;
    const int ex0 = elo & 0xFFFF, ey0 = elo >> 16, ex1 = ehi & 0xFFFF, ey1 = ehi >> 16;
    return !(x1 < ex0 || ex1 < x0 || y1 < ey0 || ey1 < y0);
}

// stage clocks for tools/lsd_cycles.py: compiled in with -DSSLAM_LSD_CYCLES only (an s_memtime + wait per read sits on the one wave's path)
#ifdef SSLAM_LSD_CYCLES
#define SSLAM_CLK() __builtin_readcyclecounter()
#else
#define SSLAM_CLK() 0ll
#endif

// region2rect + refine() (LSD_REFINE_STD) of one grown region: returns whether a rectangle goes to the NFA stage.  n / rq hold the region on
// entry and what is left USED on exit.  Main-wave form: releases and re-marks pixels in the pixel map (WANTBOX: the box of everything touched,
// of the re-gather of the chunk's remaining candidates).  SPEC form (helper wave): marks live in the private bitmap, the re-grown list goes BEHIND the first one in the
// arena and reduce_region_radius works on a copy, so that everything the helper ever accepted can be validated later.
template <bool LAT, int MARK, bool WANTBOX, class G = TorusHelper>
__device__ bool rect_refine(const LsdPlan& P, const float4 sd, int& n, double& regAngle, RegQ& rq, const Planes& pl, double* __restrict__ red,
                            RectD& rec, bool& refined, unsigned& evLo, unsigned& evHi, long long* cycs, SpecLists* sl, long long* verifyCnt) {
    const int lane = threadIdx.x & 63;
    const int sw = P.sw, sh = P.sh;
    const double prec = P.prec, p = P.p, DENSITY_TH = 0.7;
    const long long tA = SSLAM_CLK();
    region2rect_m(rq, n, sw, pl.S, regAngle, prec, p, rec, red);
    cycs[0] = SSLAM_CLK() - tA;
    // ---- refine (LSD_REFINE_STD part)
    double density = (double)n / (dist_d(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density >= DENSITY_TH) return true;
    refined = true;
    if (WANTBOX) { if (n <= QCAP) list_bbox(rq.lds, n, lane, evLo, evHi); else { evLo = 0u; evHi = 0xFFFFFFFFu; } }
    const unsigned e0 = rq.get(0);
    const int x0 = e0 & 0xFFFF, y0 = e0 >> 16;
    const double xc = (double)x0, yc = (double)y0;
    const double ang_c = (double)sd.x * DEG2RAD;      // reg[0] is the seed
    OrdSum SR; SR.acc = 0; int cnt = 0;
    for (int bs = 0; bs < n; bs += 64) {
        const int i = bs + lane;
        double ad = 0; bool in = false;
        if (i < n) {
            const unsigned e = rq.get_n(i, n);
            const int px = e & 0xFFFF, py = e >> 16, id = pl.ti(px, py);
            const float aOrig = fabsf(pl.ldT(id));      // the angle whatever the used bit says (a helper's view may be racing the main wave's marks)
            if (MARK) { const int bi = bm_bit<G>(px, py); atomicAnd(&sl->bm[bi >> 5], ~(1u << (bi & 31))); }
            else pl.stT(id, aOrig);                  // NOTUSED again
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
    if (MARK == MARK_SPEC) {
        const int capN = min(QCAP, sl->cap);
        if (capN <= 64) { sl->gaveUp = true; return false; }
        rq.lds = sl->free;
        n = tau < 1.5 ? region_grow_w<true, false, MARK_SPEC, G>(x0, y0, sd.x, sd.y, sd.z, sw, sh, pl, rq, tau, regAngle, nullptr, sl->bm, capN)
                      : region_grow_w<true, true, MARK_SPEC, G>(x0, y0, sd.x, sd.y, sd.z, sw, sh, pl, rq, tau, regAngle, nullptr, sl->bm, capN);
        if (n < 0) {
            for (int i = lane; i < -n; i += 64) { const unsigned e = rq.lds[i]; const int bi = bm_bit<G>((int)(e & 0xFFFF), (int)(e >> 16)); atomicAnd(&sl->bm[bi >> 5], ~(1u << (bi & 31))); }
            sl->gaveUp = true; return false;
        }
        sl->nB = n; sl->free += n; sl->cap -= n;
    } else n = region_grow_m<LAT, MARK, G>(x0, y0, sd.x, sd.y, sd.z, sw, sh, pl, rq, tau, regAngle, verifyCnt, MARK ? sl->bm : nullptr);
    if (WANTBOX) {                                // the re-grown region can reach outside the first one, and reduce_region_radius releases from it
        unsigned l2, h2;
        if (n <= QCAP) { list_bbox(rq.lds, n, lane, l2, h2); evLo = pk_min_u16(evLo, l2); evHi = pk_max_u16(evHi, h2); } else { evLo = 0u; evHi = 0xFFFFFFFFu; }
    }
    if (n < 2) return false;
    region2rect_m(rq, n, sw, pl.S, regAngle, prec, p, rec, red);
    density = (double)n / (dist_d(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density < DENSITY_TH) {
        const long long tr0 = SSLAM_CLK();
        if (MARK == MARK_SPEC) {                  // work on a copy: list B stays as grown
            if (sl->cap < n) {
                for (int i = lane; i < n; i += 64) { const unsigned e = rq.lds[i]; const int bi = bm_bit<G>((int)(e & 0xFFFF), (int)(e >> 16)); atomicAnd(&sl->bm[bi >> 5], ~(1u << (bi & 31))); }
                sl->gaveUp = true; return false;
            }
            for (int i = lane; i < n; i += 64) sl->free[i] = rq.lds[i];
            rq.lds = sl->free; sl->reduced = true;
        }
        // reduce_region_radius: sequential swap-with-last removal (the order feeds later sums)
        const double r1 = (rec.x1 - xc) * (rec.x1 - xc) + (rec.y1 - yc) * (rec.y1 - yc);
        const double r2 = (rec.x2 - xc) * (rec.x2 - xc) + (rec.y2 - yc) * (rec.y2 - yc);
        double radSq = r1 > r2 ? r1 : r2;
        bool good = true;
        while (density < DENSITY_TH) {
            radSq *= 0.75 * 0.75;
            if (MARK == MARK_SPEC) n = remove_far_points_lds<MARK_SPEC, G>(rq.lds, n, xc, yc, radSq, pl, red, sl->bm);
            else if (n <= QCAP) n = remove_far_points_lds<MARK, G>(rq.lds, n, xc, yc, radSq, pl, red, MARK ? sl->bm : nullptr);
            else for (int i = 0; i < n; ++i) {
                const unsigned e = rq.get(i);
                const int px = e & 0xFFFF, py = e >> 16;
                const double d2 = ((double)px - xc) * ((double)px - xc) + ((double)py - yc) * ((double)py - yc);
                if (d2 > radSq) {
                    const unsigned last = rq.get(n - 1);
                    if (lane == 0) {
                        if (MARK) { const int bi = bm_bit<G>(px, py); atomicAnd(&sl->bm[bi >> 5], ~(1u << (bi & 31))); }
                        else { unsigned* t = pl.Tb() + pl.ti(px, py); *t &= ~USED_BIT; }
                        rq.set(i, last);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                    --n; --i;
                }
            }
            if (n < 2) { good = false; break; }
            region2rect_m(rq, n, sw, pl.S, regAngle, prec, p, rec, red);
            density = (double)n / (dist_d(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        }
        cycs[2] = SSLAM_CLK() - tr0;
        if (!good) return false;
    }
    return true;
}

// One persistent single-wave workgroup per frame: the flsd() main loop replayed in order.
template <bool LAT>
__device__ __forceinline__ void lsd_regions_body(uint8_t* __restrict__ ws, const LsdPlan& P, int b, unsigned* __restrict__ dynLds, double* __restrict__ red, float4* __restrict__ seedStash) {
    const int lane = threadIdx.x & 63;
    uint8_t* base = ws + (size_t)b * P.frameBytes;
    Planes pl; pl.T = (float*)(base + P.offT); pl.Cs = (const float2*)(base + P.offCs); pl.S = (const int*)(base + P.offS); pl.tW = P.tW; pl.cW = P.cW;
    const unsigned* order = (const unsigned*)(base + P.offOrder);          // seed candidates in LSD's order, x | y << 16
    double* candOut = (double*)(base + P.offCand);
    Misc* misc = (Misc*)(base + P.offMisc);
    const int sw = P.sw, sh = P.sh;
    RegQ rq; rq.lds = dynLds; rq.glb = (unsigned*)(base + P.offReg);
    const int nOrd = misc->nDefined;
    const double prec = P.prec;
    int nSeg = 0;
    long long cyc0 = 0, cyc1 = 0, cyc2 = 0, cyc3 = 0;
    const long long tStart = SSLAM_CLK();
    for (int pos0 = 0; pos0 < nOrd; pos0 += 64) {
        const int q = pos0 + lane;
        const bool have = q < nOrd;
        const unsigned idx = have ? order[q] : 0xFFFFFFFFu;
        const int tiSeed = have ? pl.ti(idx) : 0;
        // One gather per chunk of 64 seed candidates: T is the candidate's level-line angle while it is unused.  What a region start
        // needs from its seed -- coordinates, the angle, cos and sin of it (two fp64 evaluations that the whole wave used to execute for ONE
        // seed, after a dependent load of the angle) -- is computed here for all 64 candidates at once and parked in LDS.
        const float a0 = have ? pl.ldT(tiSeed) : NOTDEF_F;
        unsigned long long unM = __ballot(t_free(a0));        // candidates of this chunk that are still unused (wave-uniform, kept up to date below)
        if (!unM) continue;
        {
            const double ar = (double)a0 * DEG2RAD;
            seedStash[lane] = make_float4(a0, (float)cos(ar), (float)sin(ar), __int_as_float((int)idx));
        }
        while (unM) {
            const int first = __ffsll((long long)unM) - 1;
            unM &= unM - 1;                                   // the seed itself is consumed whatever happens
            const float4 sd = seedStash[first];               // wave-uniform address: one broadcast read
            double regAngle;
            long long t0 = SSLAM_CLK();
            const int sxy = __float_as_int(sd.w), sx = sxy & 0xFFFF, sy = sxy >> 16;
            int n = region_grow_m<LAT>(sx, sy, sd.x, sd.y, sd.z, sw, sh, pl, rq, prec, regAngle, &misc->cyc[6]);
            long long t1 = SSLAM_CLK(); cyc0 += t1 - t0;
            if (n < P.minRegSize) {
                // too small: rejected, its pixels stay used.  Which candidates of this chunk did it take?  Compare them with the (few) points of
                // the region instead of gathering 64 pixel records again.
                for (int k = 1; k < n; ++k) {
                    const unsigned e = rq.lds[k];             // minRegSize < QCAP
                    unM &= ~__ballot(idx == e);
                }
                continue;
            }
            RectD rec;
            bool refined = false; unsigned evLo = 0xFFFFFFFFu, evHi = 0u;      // everything refine() may have released lies inside this box
            long long cycs[3] = {0, 0, 0};
            const int nGrown = n;
            const bool emit = rect_refine<LAT, false, true>(P, sd, n, regAngle, rq, pl, red, rec, refined, evLo, evHi, cycs, nullptr, &misc->cyc[6]);
            cyc1 += cycs[0]; const long long t2 = t1 + cycs[0]; cyc3 += cycs[2];
            if (!refined) { if (nGrown <= QCAP) list_bbox(rq.lds, nGrown, lane, evLo, evHi); else { evLo = 0u; evHi = 0xFFFFFFFFu; } }      // the list is the region as grown
            // ---- hand the rectangle to the NFA stage (rect_improve reads only the static angle map and never touches
            // `used`, so it is not part of the sequential dependency chain: k_nfa_all evaluates all candidates in parallel)
            long long t3 = SSLAM_CLK(); cyc2 += t3 - t2;
            // a region of this size -- kept or not, refine() may have released pixels again -- can have changed the candidates of the chunk that
            // lie inside the box of what it touched (everything it marked or released is inside; pixels of other regions are never released):
            // only those are gathered again.  (Rounds 1-2 gathered all 64: 27 k scattered lines per frame, a quarter of the core's L1 misses.)
            {
                const int ix = (int)(idx & 0xFFFF), iy = (int)(idx >> 16);
                const bool chk = have && ((unM >> lane) & 1ull) && ix >= (int)(evLo & 0xFFFF) && ix <= (int)(evHi & 0xFFFF) && iy >= (int)(evLo >> 16) && iy <= (int)(evHi >> 16);
                bool usedNow = false;
                if (chk) usedNow = !t_free(pl.T[tiSeed]);
                unM &= ~__ballot(usedNow);
            }
            if (!emit) continue;
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
        misc->cyc[0] = cyc0; misc->cyc[1] = cyc1; misc->cyc[2] = cyc2; misc->cyc[3] = cyc3; misc->cyc[4] = SSLAM_CLK() - tStart;
    }
}

// MW = waves per SIMD the register allocator leaves room for:
//   6  (80 VGPRs, 11 dwords spilled): every wave slot the register file has -- the fastest form when the core has the chip to itself (77.8 ms per 12 288 frames; 5 and 4: 86 / 85)
//   4  (97 VGPRs, nothing spilled), launched as a PERSISTENT grid of 16 - 18 workgroups per compute unit (lines.hip, lines_guest_form) that claim frames dynamically: a third of every SIMD's registers
//      stays free, so the kernels of another branch (the point branch of the bench step) are co-resident from the first millisecond instead of waiting for core waves to
//      retire -- lines.hip picks it when the caller announced such a branch (sslam_lines_set_core_event); profiles/r06c_*: 160.5 ms per step against 167.5
template <bool LAT, int MW>
__global__ __launch_bounds__(64, MW) void k_lsd_regions(uint8_t* __restrict__ ws, LsdPlan P, const double* __restrict__ lgam, int nframes) {
    extern __shared__ __align__(16) unsigned dynLds[];           // region queue (first QCAP points)
    __shared__ double red[3 * 64];                                 // addends of the ordered fp64 sums
    __shared__ float4 seedStash[64];                               // per seed candidate of the current chunk: angle, cos, sin, x | y << 16
    // gridDim.x == nframes: one frame per workgroup.  A smaller grid makes the workgroups persistent: each takes the next unclaimed frame when it
    // has finished one (Misc::claim of frame 0, zeroed by k_zero_misc), and the wave slots the grid does not fill stay free for the other
    // branch's kernels for the whole launch
    int* claim = &((Misc*)(ws + P.offMisc))->claim;
    for (int i = blockIdx.x; i < nframes;) {
        lsd_regions_body<LAT>(ws, P, xcd_mix_frame(i, nframes), dynLds, red, seedStash);
        if (gridDim.x >= (unsigned)nframes) break;
        int nxt = 0;
        if (threadIdx.x == 0) nxt = atomicAdd(claim, 1);
        i = (int)gridDim.x + __builtin_amdgcn_readfirstlane(nxt);
    }
}
