// LSD rectangle validation (rect_improve / rect_nfa / nfa): host-evaluated tables, exact division, guarded early-exit test,
// aligned-point counting, lane-dynamic NFA evaluation, in-order acceptance; self-test and calibration kernels.
// Part of lines.hip (included there, inside its anonymous namespace: one translation unit, so device helpers are shared
// without relocatable device code).  Not a standalone header.
#pragma once

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
#if defined(SSLAM_TESTING) && !defined(SSLAM_NFA_STAGE_ONLY)
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
#endif
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
    if (!(term > 1e-280)) return -1;      // the guard bands below are relative: not near the denormals (decision D11's variant 1 reaches them)
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

#if defined(SSLAM_TESTING) && !defined(SSLAM_NFA_STAGE_ONLY)
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
#endif
// LineSegmentDetectorImpl::nfa() split for lane-dynamic scheduling: nfa_setup() covers everything before the binomial-tail
// loop, tail_block() advances the loop by up to eight terms, and the caller finishes with -log10(bin_tail) - logNT.
struct TailState { double term, bin_tail, p_term; int n, i; };           // i = next term index (k+1 .. n)

// returns true when the tail loop has to run; otherwise v is the function value
// variant (decision D11, LsdPlan::nfaVariant): 0 = the first term of log1term is log_gamma(n + 1) (the binomial coefficient, von Gioi's lsd.c); 1 = it is (double(n) + 1), as
// OpenCV's lsd.cpp is recalled to have it.  Under 1 the term is smaller by e^(lgamma(n+1) - (n+1)): it is a denormal for n around 200-240 and an exact 0 beyond.
__device__ __forceinline__ bool nfa_setup(int n, int k, double p, double logNT, int variant, const double* __restrict__ lgam, const PLog* __restrict__ plog, TailState& S, double& v) {
    if (n == 0 || k == 0) { v = -logNT; return false; }
    const int h = 1020 - ((__double2hiint(p) >> 20) & 0x7FF);        // p is an exact power of two
    const bool tab = h >= 0 && h < 16 && p == ldexp(0.125, -h);
    if (n == k) { v = -logNT - (double)n * (tab ? plog[h].l10p : log10(p)); return false; }
    const double p_term = p / (1 - p);
    const double first = variant ? (double)(n + 1) : lgam[n + 1];
    const double log1term = first - lgam[k + 1] - lgam[n - k + 1] + (double)k * (tab ? plog[h].lp : log(p)) + (double)(n - k) * (tab ? plog[h].l1mp : log(1.0 - p));
    const double term = exp(log1term);
    // double_equal(term, 0): |term| / max(|term|, DBL_MIN) <= 100 * DBL_EPSILON -- an exact zero, or a denormal of at most 100 units (m * 2^-1074 / 2^-1022 = m * 2^-52 <= 100 * 2^-52).
    // (Rounds 1-4 tested `term == 0.0`: the two differ for log1term in [-745.1, -739.9], which decision D11's variant 1 reaches on ordinary frames; found by tests/test_variants_cpu.py.)
    if (term <= 0x1.9p-1068) {
        v = ((double)k > (double)n * p) ? -log1term / 2.30258509299404568402 - logNT : -logNT;
        return false;
    }
    S.term = term; S.bin_tail = term; S.p_term = p_term; S.n = n; S.i = k + 1;
    return true;
}

// up to eight terms of the tail loop; true when the loop is over (early exit or i > n)
constexpr int TBK = 8;      // terms per block (four: 20 fewer registers in the evaluator, 27.5 -> 31.3 ms under D11 = 0, GPU call AB)
__device__ __forceinline__ bool tail_block(TailState& S, double logNT, const double* __restrict__ rcp) {
    const double tolerance = 0.1;
    const int n = S.n, i0 = S.i;
    double term = S.term, bin_tail = S.bin_tail;
    double mt[TBK];
#pragma unroll
    for (int j = 0; j < TBK; ++j) {           // independent divisions: issue back to back
        const int i = min(i0 + j, n);
        mt[j] = exact_div((double)(n - i + 1), (double)i, rcp[i]) * S.p_term;
    }
    bool done = false;
#pragma unroll
    for (int j = 0; j < TBK; ++j) {
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
    S.term = term; S.bin_tail = bin_tail; S.i = i0 + TBK;
    return done || S.i > n;
}

// ------------------------------------------------------------------ rect_improve as staged, fully parallel kernels
// rect_improve (LSD_REFINE_ADV) evaluates the rectangle, then five refinement stages of up to five candidate rectangles
// each; inside a stage the candidates do not depend on which of them is accepted.  Per stage two launches cover every
// candidate of every frame: k_nfa_count (one wave per rectangle: aligned-point counts of the stage's candidates) and
// k_nfa_eval (the binomial-tail NFAs of all candidates, lanes scheduled dynamically) + k_nfa_accept (the reference's sequential acceptance).
struct NfaState { double logNfa; int done, nc; int cnt[6][2]; double val[6]; };      // per rectangle; cnt[k] = {total, aligned}, val[j] = NFA of candidate j


// Everything below is the stage itself: compiled by lines_nfa.hip only (round 4 compiled it into lines.hip as well -- a second k_nfa_all with 81 spilled VGPRs that nothing launched).
#ifdef SSLAM_NFA_STAGE_ONLY
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
// `small` as a run-time (wave-uniform) flag: one instance of the counters instead of two
__device__ __forceinline__ void nfa_row_range(const NfaGeom& g, int y, int sw, bool small, int& xa, int& xb) {
    if (small) {
        const int steps = y - g.y0;
        int nl2 = 0, nr2 = 0;
        if (steps > 0) { nl2 = max(0, y - max(g.ly, g.y0)); nr2 = max(0, y - max(g.ry, g.y0)); }
        xa = max(g.mx + (steps - nl2) * g.fl + nl2 * g.sl, 0); xb = min(g.mx + (steps - nr2) * g.fr + nr2 * g.sr, sw - 1);
    } else nfa_row_range<false>(g, y, sw, xa, xb);
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

// wave-wide vote on ONE comparison: the compare writes the lane mask itself.  (HIP's __ballot takes an int, and a vote on a conjunction
// makes the compiler rebuild a 0/1 vector from the scalar masks -- v_cndmask + v_cmp per vote; conjunctions are done on the masks instead.)
__device__ __forceinline__ unsigned long long vote(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// k_nfa_count: one wave walks a frame's rectangles.  The corner bookkeeping of rect_nfa (nfa_geom: sorting, slopes, integer
// divisions) is the same few hundred instructions whether one lane or sixty-four execute it, so it runs lane-parallel for a
// batch of (rectangle, candidate) items whose results are parked in LDS; the wave then counts the items one after
// the other with all lanes on the pixels.  Counters are wave-uniform (ballot + popcount), so nothing is reduced at the end.
//
// Round 3: the pixels a candidate covers are no longer counted by votes -- that total is the sum of its row widths, accumulated by the
// first lane of every row -- and the angle plane is T (|T|: the used bit is the sign; NOTDEF becomes 1024).
// (An integer form of the alignment test -- at most two windows of fp32 angle bit patterns per (theta, tolerance) -- was exact and 4 % slower: the counter is bound by
// its row-range bookkeeping and its dependent loads, not by the predicate.  Removed in round 6; docs/history has the measurement.)
constexpr int EVAL_CH = 1024;          // rectangles per item-list chunk (k_nfa_count, k_nfa_eval)
constexpr int EVAL_REFILL = 16;
constexpr int CNT_NEST = 64;
constexpr unsigned PIX_NONE = 0x7FFFFFFFu;      // what a lane without a pixel holds: above every interval
struct CntItem { NfaGeom g; int c, j, lg;      // lg: log2 of the lanes sharing a row
                 double theta, prec, p;
};
__device__ __forceinline__ double align_dist_min(float aDeg, double theta) {
    const double n_theta = fabs(theta - (double)aDeg * DEG2RAD);
    return fmin(n_theta, fabs(n_theta - M_2PI_));
}

// A rectangle's geometry is the same for every lane of the wave that counts it, but it comes out of LDS, i.e. in vector registers: handed to the scalar file explicitly
// (nine registers per candidate, forty-five in count_rect5), the row-range arithmetic takes them as scalar operands
__device__ __forceinline__ NfaGeom geom_uniform(const NfaGeom& g) {
    NfaGeom u;
    u.mx = __builtin_amdgcn_readfirstlane(g.mx); u.y0 = __builtin_amdgcn_readfirstlane(g.y0); u.y1 = __builtin_amdgcn_readfirstlane(g.y1);
    u.ly = __builtin_amdgcn_readfirstlane(g.ly); u.ry = __builtin_amdgcn_readfirstlane(g.ry); u.fl = __builtin_amdgcn_readfirstlane(g.fl);
    u.sl = __builtin_amdgcn_readfirstlane(g.sl); u.fr = __builtin_amdgcn_readfirstlane(g.fr); u.sr = __builtin_amdgcn_readfirstlane(g.sr);
    return u;
}
__device__ __forceinline__ double uniform_f64(double x) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}

// Pixel walk shared by the two counters below.  A row is shared by 2^lg lanes (lg picked per rectangle from its widest
// row: tall thin rectangles put 32 rows in flight, flat ones spread one row over the whole wave); each lane owns a
// contiguous run of the row and the wave steps through the runs twelve pixels at a time.
// T of pixel x of the row that starts at element yb, RAW: the used bit is masked off where the value is consumed (T_ABS), not here --
// inside the conditional load the mask needs the loaded value, so the compiler waits for every load before it issues the next one
// (measured: the counter 33 % slower, twelve dependent round trips per run instead of twelve loads in flight)
__device__ __forceinline__ unsigned t_raw(const unsigned* __restrict__ Tb, int yb, int x) { return Tb[yb + x]; }
// the angle whatever the used bit says; 1024 for NOTDEF, PIX_NONE stays above every hi.  The empty asm pins the raw value in a register at
// the point of use: without it the compiler sinks the mask back into the conditional load
__device__ __forceinline__ unsigned t_abs_u(unsigned v) { asm volatile("" : "+v"(v)); return v & 0x7FFFFFFFu; }
#define T_ABS(v) ((int)t_abs_u(v))
struct __attribute__((packed, aligned(4))) T4 { unsigned v[4]; };      // four pixels of a row of T from any dword

// aligned-point counts of one rectangle for six nested tolerances (precs[k], around theta); total = pixels visited.
__device__ __forceinline__ void count_item(const NfaGeom& gIn, int lg, const unsigned* __restrict__ Tb, int tW, int sw, bool small,
                                           int lane, int& totalOut, int (&alg)[6], double theta = 0, const double* precs = nullptr) {
    constexpr int K = 6;
    const NfaGeom g = geom_uniform(gIn);
    const int nrows = g.y1 - g.y0 + 1;
    const int rowsPer = 64 >> lg, r = lane >> lg, sub = lane & ((1 << lg) - 1);
    int total = 0;
    for (int t0 = 0; t0 < nrows; t0 += rowsPer) {
        const int t = t0 + r;
        int xa = 0, xb = -1; const int y = g.y0 + t;
        if (t < nrows) nfa_row_range(g, y, sw, small, xa, xb);
        const int width = max(xb - xa + 1, 0);
        total += sub == 0 ? width : 0;
        const int share = (width + (1 << lg) - 1) >> lg;
        const int xs = xa + sub * share;
        const int mine = max(min(share, xb - xs + 1), 0);
        const int yb = tix(0, max(y, 0), tW);
        for (int c0 = 0; vote(c0 < mine) != 0; c0 += 12) {
            unsigned a[12];
            // a lane's run as three 16-byte loads (dword-aligned: the hardware takes them) instead of twelve dword loads: every lane of a gather is on a row of its own, so
            // the address unit spends its cycles per INSTRUCTION and lane, not per byte.  A load may run up to three pixels past the lane's run -- into the neighbour's run, the
            // rest of the row or the plane behind T, all inside the frame's workspace --; those slots are masked where the votes are taken (`valid`), not here (a select on the
            // loaded value would make every load a round trip of its own, see t_raw).
#pragma unroll
            for (int gq = 0; gq < 3; ++gq) {
                if (gq == 0 || vote(c0 + 4 * gq < mine)) {
                    T4 t = {{PIX_NONE, PIX_NONE, PIX_NONE, PIX_NONE}};
                    if (c0 + 4 * gq < mine) t = *(const T4*)(Tb + yb + xs + c0 + 4 * gq);
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[4 * gq + q] = t.v[q];
                }
            }
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const bool mineq = c0 + q < mine;
                if (!vote(mineq)) break;
                const float af = __uint_as_float(t_abs_u(a[q]));
                // a slot that holds no pixel of this lane's run (NOTDEF is 1024 here; past the run: PIX_NONE or, with 16-byte loads, somebody else's pixel) gets the distance
                // +inf ONCE -- one select on the upper half -- instead of a mask that every one of the six votes below is and-ed with: 13 + 27 -> 14 + 15 vector + scalar
                // instructions per slot (round 6: the stage's time is the SUM of the two, 299 k + 239 k per frame x 4 cycles = its 12.5 ms)
                const double d0 = align_dist_min(af, theta);
                const double dd = __hiloint2double((mineq && af < 1000.f) ? __double2hiint(d0) : 0x7FF00000, __double2loint(d0));
#pragma unroll
                for (int k = 0; k < K; ++k) alg[k] += __popcll(vote(dd <= precs[k]));
            }
        }
    }
    totalOut = wave_sum_dpp(total);
}
// (Round 5, GPU call Q, measured and removed: under D11 = 1 and p = 1 / 8 the initial evaluation of a rectangle of n >= 256 pixels is a density test -- log1term <= (n + 1) -
// lgamma(n + 1) <= -910 sends exp() to an exact 0 whatever k is, and nfa() then accepts iff 8 k > n --, n follows from the row widths alone, so the count could stop at the
// first row group that settles it.  Exact (the whole line suite ran with it) and no faster: too few candidates are that large, the pass that finds n costs what the exit saves:
// initial count 5.2 -> 5.6 ms, k_nfa_all 12.0 -> 12.3 ms per 12 288 frames.)

// Stages 1-3: the (up to five) candidates of a rectangle differ by half-pixel width / offset steps and share theta and the
// tolerance, so they are counted in ONE pass over the union of their rows: the angle test runs once per pixel, membership in
// candidate j is two integer compares against that candidate's own row range (rect_nfa's edge stepping, per candidate).
__device__ __forceinline__ void count_rect5(const CntItem* __restrict__ it5, int nc, int lg, const unsigned* __restrict__ Tb, int tW, int sw, bool small, int lane,
                                            int (&total)[MAXC], int (&alg)[MAXC]) {
    const double theta = uniform_f64(it5[0].theta), prec = uniform_f64(it5[0].prec);
    NfaGeom g[MAXC];
#pragma unroll
    for (int j = 0; j < MAXC; ++j) g[j] = geom_uniform(it5[j < nc ? j : 0].g);
    int y0u = g[0].y0, y1u = g[0].y1;
#pragma unroll
    for (int j = 1; j < MAXC; ++j) if (j < nc) { y0u = min(y0u, g[j].y0); y1u = max(y1u, g[j].y1); }
    const int nrows = y1u - y0u + 1;
    const int rowsPer = 64 >> lg, r = lane >> lg, sub = lane & ((1 << lg) - 1);
#pragma unroll
    for (int j = 0; j < MAXC; ++j) total[j] = 0;
    for (int t0 = 0; t0 < nrows; t0 += rowsPer) {
        const int t = t0 + r;
        const int y = y0u + t;
        int xaj[MAXC], xbj[MAXC];
        int xa = 0x7fffffff, xb = -1;
        unsigned wj[MAXC];                                              // xbj - xaj of a candidate's row; an empty row: xaj far to the right of every pixel, width 0
#pragma unroll
        for (int j = 0; j < MAXC; ++j) {
            xaj[j] = 1; xbj[j] = 0;
            if (j < nc && t < nrows && y >= g[j].y0 && y <= g[j].y1) {
                nfa_row_range(g[j], y, sw, small, xaj[j], xbj[j]);
                if (xbj[j] >= xaj[j]) { xa = min(xa, xaj[j]); xb = max(xb, xbj[j]); }
                else { xaj[j] = 1; xbj[j] = 0; }
            }
            total[j] += sub == 0 ? xbj[j] - xaj[j] + 1 : 0;             // the candidate's pixels in this row (0 for an empty row: xaj = 1, xbj = 0)
            const bool emptyRow = xbj[j] < xaj[j];
            wj[j] = emptyRow ? 0u : (unsigned)(xbj[j] - xaj[j]);
            if (emptyRow) xaj[j] = 0x40000000;
        }
        const int width = xb >= xa ? xb - xa + 1 : 0;
        const int share = (width + (1 << lg) - 1) >> lg;
        const int xs = xa + sub * share;
        const int mine = width > 0 ? max(min(share, xb - xs + 1), 0) : 0;
        const int yb = tix(0, max(y, 0), tW);
        for (int c0 = 0; vote(c0 < mine) != 0; c0 += 12) {
            unsigned a[12];
            // three 16-byte loads per run, as in count_item (round 6: this counter still issued twelve dword loads per run -- it is where the time of the refinement stages
            // goes, i.e. most of the stage under decision D11 = 0: 26.6 -> 26.4 ms there, 12.2 -> 12.0 under D11 = 1, GPU call H); slots past the lane's run hold a neighbour's pixel and are masked at the vote (`valid`)
#pragma unroll
            for (int gq = 0; gq < 3; ++gq) {
                if (gq == 0 || vote(c0 + 4 * gq < mine)) {
                    T4 t = {{PIX_NONE, PIX_NONE, PIX_NONE, PIX_NONE}};
                    if (c0 + 4 * gq < mine) t = *(const T4*)(Tb + yb + xs + c0 + 4 * gq);
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[4 * gq + q] = t.v[q];
                }
            }
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const bool mineq = c0 + q < mine;
                if (!vote(mineq)) break;
                const float af = __uint_as_float(t_abs_u(a[q]));
                // "aligned, defined, mine" moves the pixel's column to where no candidate's row reaches (one select) instead of masking each candidate's two votes; membership
                // in candidate j's row [xaj, xbj] is ONE unsigned compare of x - xaj with the row's width
                const bool al = mineq && af < 1000.f && align_dist_min(af, theta) <= prec;
                const unsigned x = al ? (unsigned)(xs + c0 + q) : 0x80000000u;
#pragma unroll
                for (int j = 0; j < MAXC; ++j) {
                    if (j < nc) alg[j] += __popcll(vote(x - (unsigned)xaj[j] <= wj[j]));
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < MAXC; ++j) total[j] = wave_sum_dpp(total[j]);
}


// stage 0 is merged with the initial evaluation: same rectangle, six precisions (p, p/2 .. p/32); stage 4 likewise has one
// geometry and five precisions.  Stages 1-3 change the rectangle itself: up to five candidates per rectangle.
#ifndef SSLAM_COUNT_MINWAVES
#define SSLAM_COUNT_MINWAVES 4
#endif
// LDS of one wave of the NFA stage: the counter's item batch + active list, or the evaluator's item list (never both at once)
template <int CH>
struct NfaCountLdsT {
    CntItem its[64];
    unsigned short act[CH];
};
template <int CH> union NfaLdsT { NfaCountLdsT<CH> c; unsigned short items[CH * 5]; };      // items: (rect - chunk) << 3 | candidate; CH = rectangles per chunk
typedef NfaCountLdsT<EVAL_CH> NfaCountLds;
typedef NfaLdsT<EVAL_CH> NfaLds;

// the inner synchronisation of the per-wave stage bodies: only the wave's own LDS arrays are at stake, and every kernel of this stage is a single-wave workgroup
__device__ __forceinline__ void nfa_wave_sync() { __syncthreads(); }

// aligned-point counts of the stage's candidates for the rectangles [part * per, ...) of one frame: the body of one wave
// (the bodies come as *_range over the rectangles [c0, c1) -- what the streaming form below hands out block by block -- and as *_body over a wave's share of the frame)
template <int CH>
__device__ __forceinline__ void nfa_count_range(uint8_t* __restrict__ base, const LsdPlan& P, int stage, int c0, int c1, int lane, NfaCountLdsT<CH>& L) {
    CntItem* its = L.its;
    unsigned short* act = L.act;
    Misc* misc = (Misc*)(base + P.offMisc);
    const unsigned* Tb = (const unsigned*)(base + P.offT);
    const double* rects = (const double*)(base + P.offCand);
    NfaState* st = (NfaState*)(base + P.offNfa);
    const int sw = P.sw, sh = P.sh, tW = P.tW;
    const bool nested = stage == 0 || stage == 4;
    const bool small = sw < 32768 && sh < 32768;
    const int rpb = nested ? CNT_NEST : 12;                        // rectangles per batch (stages 1-3: five lanes each)
    for (int chunk = c0; chunk < c1; chunk += CH) {
        const int cend = min(chunk + CH, c1);
        int nAct = 0;
        for (int cb = chunk; cb < cend; cb += 64) {               // rectangles still being refined
            const int c = cb + lane;
            const bool on = c < cend && (stage == 0 || !st[c].done);
            const unsigned long long m = __ballot(on);
            if (on) act[nAct + mbcnt(m)] = (unsigned short)(c - chunk);
            nAct += __popcll(m);
        }
        nfa_wave_sync();
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
                        I.g = nfa_geom(r, sh);
                        const int need = (nfa_max_width(I.g) + (nested ? 0 : 3) + 11) / 12;       // lanes per row so that a run is <= 12 pixels
                        int lg = 1; while ((1 << lg) < need && lg < 6) ++lg;
                        I.lg = lg;
                        I.theta = r.theta; I.prec = r.prec; I.p = r.p;
                    }
                }
                const unsigned long long vm = __ballot(valid);
                if (lane < nIt) {
                    if (nested) { if (!valid) st[c].nc = 0; }
                    else if (j == 0) st[c].nc = __popcll((vm >> lane) & 31ull);
                }
            }
            nfa_wave_sync();
            if (!nested) {
                for (int ri = 0; ri < nr; ++ri) {
                    const CntItem* it5 = its + ri * MAXC;
                    int nc = 0;
#pragma unroll
                    for (int j = 0; j < MAXC; ++j) nc += it5[j].j >= 0 ? 1 : 0;           // valid candidates form a prefix
                    if (nc == 0) continue;
                    int total[MAXC], alg[MAXC], tot2[MAXC];
#pragma unroll
                    for (int j = 0; j < MAXC; ++j) alg[j] = 0;
                    count_rect5(it5, nc, it5[0].lg, Tb, tW, sw, small, lane, total, alg);
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
                const int c = its[it].c;
                int total, tot2, alg[6] = {0, 0, 0, 0, 0, 0};
                const int lg = its[it].lg;
                double precs[6];
                for (int k = 0; k < 6; ++k) precs[k] = stage == 0 ? (k == 0 ? its[it].prec : ldexp(its[it].p, -k) * kPI) : ldexp(its[it].p, -(k + 1)) * kPI;
                count_item(g, lg, Tb, tW, sw, small, lane, total, alg, its[it].theta, precs); (void)tot2;
                if (lane == 0) {
                    const int K = stage == 0 ? 6 : 5;
#pragma unroll
                    for (int k = 0; k < 6; ++k) if (k < K) { st[c].cnt[k][0] = total; st[c].cnt[k][1] = alg[k]; }
                    st[c].nc = K;
                }
            }
            nfa_wave_sync();
        }
    }
}

template <int CH>
__device__ __forceinline__ void nfa_count_body(uint8_t* __restrict__ base, const LsdPlan& P, int stage, int part, int nparts, int lane, NfaCountLdsT<CH>& L) {
    const int nCand = ((const Misc*)(base + P.offMisc))->nCand;
    const int per = (nCand + nparts - 1) / nparts;
    const int c0 = part * per, c1 = min(c0 + per, nCand);
    nfa_count_range<CH>(base, P, stage, c0, c1, lane, L);
}

__global__ __launch_bounds__(64, SSLAM_COUNT_MINWAVES) void k_nfa_count(uint8_t* __restrict__ ws, LsdPlan P, int stage) {
    __shared__ NfaCountLds L;
    const int b = gridDim.x == 1 ? xcd_mix_frame(blockIdx.y, gridDim.y) : blockIdx.y;
    nfa_count_body<EVAL_CH>(ws + (size_t)b * P.frameBytes, P, stage, blockIdx.x, gridDim.x, threadIdx.x, L);
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
// log10 behind a call (SSLAM_NFA_LOG10_CALL): inlined into the fused kernel its polynomial constants are hoisted out of the evaluation loop into
// registers the loop does not have, spilled, and reloaded from scratch one by one with a wait each (six dependent round trips per use)
__device__ __noinline__ double nfa_log10(double x) { return log10(x); }
template <int CH>
__device__ __forceinline__ void nfa_eval_range(uint8_t* __restrict__ base, const LsdPlan& P, int stage, const double* __restrict__ lgam, int c0, int c1, int lane,
                                                unsigned short* __restrict__ items) {
    Misc* misc = (Misc*)(base + P.offMisc); (void)misc;
    const double* rects = (const double*)(base + P.offCand);
    NfaState* st = (NfaState*)(base + P.offNfa);
    const PLog* plog = (const PLog*)(lgam + P.npx + 4);
    const double* rcp = lgam + P.npx + 4 + 48;
#ifdef SSLAM_LSD_STATS
    long long useful = 0, executed = 0, evals = 0;
#endif
    for (int chunk = c0; chunk < c1; chunk += CH) {
        const int cend = min(chunk + CH, c1);
        int nItems = 0;
        for (int cb = chunk; cb < cend; cb += 64) {
            const int c = cb + lane;
            const int cnt = c < cend ? stage_ncand(st[c], stage) : 0;
            const int incl = wave_incl_scan(cnt);
            const int ex = nItems + incl - cnt;
            for (int j = 0; j < cnt; ++j) items[ex + j] = (unsigned short)(((c - chunk) << 3) | j);
            nItems += __builtin_amdgcn_readlane(incl, 63);
        }
        nfa_wave_sync();
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
                    if (needLog) v = -nfa_log10(S.bin_tail) - P.logNT;
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
                            needLog = nfa_setup(n, k, p, P.logNT, P.nfaVariant, lgam, plog, S, v);
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
        nfa_wave_sync();
    }
#ifdef SSLAM_LSD_STATS
    // cyc[5] = useful tail iterations (upper bound: whole blocks), cyc[6] = lane-iterations the wave executed, cyc[7] = evaluations
    atomicAdd((unsigned long long*)&misc->cyc[5], (unsigned long long)useful); atomicAdd((unsigned long long*)&misc->cyc[6], (unsigned long long)executed);
    atomicAdd((unsigned long long*)&misc->cyc[7], (unsigned long long)evals);
#endif
}

template <int CH>
__device__ __forceinline__ void nfa_eval_body(uint8_t* __restrict__ base, const LsdPlan& P, int stage, const double* __restrict__ lgam, int part, int nparts, int lane,
                                               unsigned short* __restrict__ items) {
    const int nCand = ((const Misc*)(base + P.offMisc))->nCand;
    const int per = (nCand + nparts - 1) / nparts;
    const int c0 = part * per, c1 = min(c0 + per, nCand);
    nfa_eval_range<CH>(base, P, stage, lgam, c0, c1, lane, items);
}

__global__ __launch_bounds__(64) void k_nfa_eval(uint8_t* __restrict__ ws, LsdPlan P, int stage, const double* __restrict__ lgam) {
    __shared__ unsigned short items[EVAL_CH * 5];
    const int b = gridDim.x == 1 ? xcd_mix_frame(blockIdx.y, gridDim.y) : blockIdx.y;
    nfa_eval_body<EVAL_CH>(ws + (size_t)b * P.frameBytes, P, stage, lgam, blockIdx.x, gridDim.x, threadIdx.x, items);
}

// rect_improve's acceptance, in candidate order, one lane per rectangle (the candidates of a stage do not depend on which of
// them is accepted, so they were all evaluated up front).
__device__ __forceinline__ void nfa_accept_range(uint8_t* __restrict__ base, const LsdPlan& P, int stage, int c0, int c1, int tid, int nthreads) {
    double* rects = (double*)(base + P.offCand);
    NfaState* st = (NfaState*)(base + P.offNfa);
    for (int c = c0 + tid; c < c1; c += nthreads) {
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

__device__ __forceinline__ void nfa_accept_body(uint8_t* __restrict__ base, const LsdPlan& P, int stage, int tid, int nthreads) {
    nfa_accept_range(base, P, stage, 0, ((const Misc*)(base + P.offMisc))->nCand, tid, nthreads);
}

__global__ __launch_bounds__(256) void k_nfa_accept(uint8_t* __restrict__ ws, LsdPlan P, int stage) {
    nfa_accept_body(ws + (size_t)blockIdx.y * P.frameBytes, P, stage, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

__device__ __forceinline__ void nfa_finish_range(uint8_t* __restrict__ base, const LsdPlan& P, int c0, int c1, int tid, int nthreads) {
    const double* rects = (const double*)(base + P.offCand);
    const NfaState* st = (const NfaState*)(base + P.offNfa);
    float4* seg = (float4*)(base + P.offSeg);
    int* flag = (int*)(base + P.offFlag);
    for (int c = c0 + tid; c < c1; c += nthreads) {
        const bool ok = st[c].logNfa > 0.0;
        flag[c] = ok ? 1 : 0;
        if (ok) {
            const double* o = rects + (size_t)c * 12;
            const double SCALE = 0.8;
            seg[c] = make_float4((float)((o[0] + 0.5) / SCALE), (float)((o[1] + 0.5) / SCALE), (float)((o[2] + 0.5) / SCALE), (float)((o[3] + 0.5) / SCALE));
        }
    }
}

__device__ __forceinline__ void nfa_finish_body(uint8_t* __restrict__ base, const LsdPlan& P, int tid, int nthreads) {
    nfa_finish_range(base, P, 0, ((const Misc*)(base + P.offMisc))->nCand, tid, nthreads);
}

__global__ __launch_bounds__(256) void k_nfa_finish(uint8_t* __restrict__ ws, LsdPlan P) {
    nfa_finish_body(ws + (size_t)blockIdx.y * P.frameBytes, P, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// The whole NFA stage of one frame in ONE launch (round 4).  count -> evaluate -> accept are frame-local, so a frame's stages need no
// launch boundary between them: W waves of one workgroup (W = blockDim.x / 64; 1 in the batch form, where the frames themselves fill the
// chip) walk the frame's rectangles stage after stage with a workgroup barrier in between.  18 dependent launches became one: the line
// stream queues once behind the point branch's grids instead of 18 times, and a frame no longer waits at every stage for the slowest
// frame of the batch -- only the kernel's one tail is left.  Results are those of the separate launches (same bodies, same order).
template <int CH>
__device__ __forceinline__ void nfa_all_body(uint8_t* __restrict__ base, const LsdPlan& P, const double* __restrict__ lgam, int wave, int nwaves, int lane, NfaLdsT<CH>& L) {
    const int nthreads = nwaves * 64;
#pragma unroll 1
    for (int it = -1; it <= 4; ++it) {
        // Every body derives its addresses from (base, lgam, lane).  Made opaque per use, none of that is loop invariant any more: hoisted out of
        // this loop the bodies' address arithmetic was live across all of them (131 spilled SGPRs, 14 spilled VGPRs, 60 bytes of scratch).
#define NFA_OPAQUE() asm volatile("" : "+s"(base), "+s"(lgam), "+v"(lane))
        NFA_OPAQUE();
        if (it != 0) {                                   // stage 0's counts came with the initial evaluation's (nested tolerances, one pass)
            nfa_count_body<CH>(base, P, it < 0 ? 0 : it, wave, nwaves, lane, L.c);
            __syncthreads();
        }
        NFA_OPAQUE();
        nfa_eval_body<CH>(base, P, it, lgam, wave, nwaves, lane, L.items);
        __syncthreads();
        NFA_OPAQUE();
        nfa_accept_body(base, P, it, wave * 64 + lane, nthreads);
        __syncthreads();
    }
    nfa_finish_body(base, P, wave * 64 + lane, nthreads);
#undef NFA_OPAQUE
}

#ifndef SSLAM_NFA_ALL_MINWAVES
#define SSLAM_NFA_ALL_MINWAVES 4
#endif
// CH: rectangles per item-list chunk = the kernel's LDS (10 bytes per rectangle): 768 -> 7.5 KB, twenty workgroups = five waves per SIMD on a compute unit's 160 KB, which the
// 92 vector registers of the kernel allow since the rectangles' geometry moved to scalar registers (geom_uniform: 128 registers and 16 bytes of scratch before).
// Per 12 288 frames (GPU calls AA, AB): 13.4 -> 12.5 ms (D11 = 1), 29.6 -> 27.5 (D11 = 0); six waves (80 registers, 32 - 48 bytes of scratch, chunks of 640) 12.6 / 27.8: no further gain
template <int CH>
__global__ __launch_bounds__(64, SSLAM_NFA_ALL_MINWAVES) void k_nfa_all(uint8_t* __restrict__ ws, LsdPlan P, const double* __restrict__ lgam) {
    __shared__ NfaLdsT<CH> L;
    const int b = xcd_mix_frame(blockIdx.x, gridDim.x);
    nfa_all_body<CH>(ws + (size_t)b * P.frameBytes, P, lgam, 0, 1, threadIdx.x, L);
}


// ------------------------------------------------------------------ streaming form (single frames / calls of up to 64 frames in the cluster form; the default since round 5, SSLAM_NFA_STREAM=0 turns it off)
// A single frame's NFA stage is 0.6 ms of launches behind a 5 ms core that produces its rectangles one after the other -- and no rectangle's verdict
// feeds back into the core (a rejected rectangle's pixels stay USED).  So the stage can run WHILE the core runs: the cluster form's main wave writes its
// rectangle records into a staging array of the frame's cluster slot with L1-bypassing stores and publishes the number of complete records with every rectangle
// (lsd_cluster.h, cl_main<G, true>); waves of this kernel, launched on a second stream next to the core, claim 1 .. NFA_STREAM_BLOCK published rectangles (CAS on a
// cursor), copy their records into the workspace and run the whole chain -- count / evaluate / accept for stages -1 .. 4, then the segment output -- on them:
// the chain of a rectangle depends on nothing but the rectangle.  What is left behind the core is the chain of the last rectangle.
//   * The main wave never waits for a consumer, and a consumer waits for nothing but the main wave's progress (bounded: `spinTicks` of the 100 MHz
//     clock, then it leaves).  The SAME kernel is launched once more behind the core with spinTicks = 0: everything is published by then, its waves take
//     whatever blocks are unclaimed (none, normally) and return.  So the result does not depend on the consumers having run at all.
//   * Visibility: records and counters cross compute units and possibly XCDs -> agent-scope (sc1) stores by the main wave, records before
//     s_waitcnt vmcnt(0) before the counter; agent-scope loads here, counter before records.  The block's working copy in the workspace, the NfaState and the
//     outputs are private to the claiming wave until the kernel ends.
//   * Results are those of the launches: same bodies (nfa_*_range), same order per rectangle.
__device__ __forceinline__ int ns_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ns_ld64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(64, SSLAM_NFA_ALL_MINWAVES) void k_nfa_stream(uint8_t* __restrict__ ws, LsdPlan P, const double* __restrict__ lgam, uint8_t* __restrict__ clArea,
                                                                          size_t clFrameBytes, size_t stageOff, long long spinTicks, int takeMax, int sleepReps) {
    __shared__ NfaLds L;
    const int f = blockIdx.y;
    int lane = threadIdx.x;
    uint8_t* base = ws + (size_t)f * P.frameBytes;
    uint8_t* area = clArea + (size_t)f * clFrameBytes;
    NfaStreamCtl* ns = (NfaStreamCtl*)(area + NFA_STREAM_CTL_OFF);
    const unsigned long long* staged = (const unsigned long long*)(area + stageOff);
    unsigned long long* rects = (unsigned long long*)(base + P.offCand);
    long long t0 = (long long)wall_clock64();                                       // (of the last progress this wave saw)
    for (;;) {
        int c0 = -1, c1 = 0;
        if (lane == 0) {
            for (;;) {
                const int fin = ns_ld(&ns->candFinal);                               // (read before candReady: a set flag makes the count final)
                const int ready = fin ? fin - 1 : ns_ld(&ns->candReady);
                const int cur = ns_ld(&ns->claim);
                if (cur < ready) {
                    // A quarter of what is waiting, 1 .. takeMax rectangles: while the core runs they arrive one at a time and each goes to a wave of its own at once (a
                    // rectangle's chain is ~50 us of dependent steps; what is behind the core in the end is the chain of its last rectangle), a backlog goes out in larger pieces.
                    const int take = min(takeMax, max(1, (ready - cur) >> 2));
                    if (atomicCAS(&ns->claim, cur, cur + take) == cur) { c0 = cur; c1 = cur + take; break; }
                    continue;
                }
                if (fin) break;                                                      // everything is handed out
                if ((long long)wall_clock64() - t0 >= spinTicks) { atomicAdd(&ns->expired, 1); break; }
                for (int q = 0; q < sleepReps; ++q) __builtin_amdgcn_s_sleep(127);      // ~3.4 us each: every poll is three loads that go past the caches
            }
        }
        c0 = __builtin_amdgcn_readfirstlane(c0); c1 = __builtin_amdgcn_readfirstlane(c1);
        if (c0 < 0) return;
        for (int i = lane; i < (c1 - c0) * 12; i += 64) rects[(size_t)c0 * 12 + i] = ns_ld64(staged + (size_t)c0 * 12 + i);
        __syncthreads();
#pragma unroll 1
        for (int it = -1; it <= 4; ++it) {
#define NFA_OPAQUE() asm volatile("" : "+s"(base), "+s"(lgam), "+v"(lane))       // (as in nfa_all_body: nothing of a body's address arithmetic may be hoisted across the chain)
            NFA_OPAQUE();
            if (it != 0) { nfa_count_range<EVAL_CH>(base, P, it < 0 ? 0 : it, c0, c1, lane, L.c); __syncthreads(); }
            NFA_OPAQUE();
            nfa_eval_range<EVAL_CH>(base, P, it, lgam, c0, c1, lane, L.items);
            __syncthreads();
            NFA_OPAQUE();
            nfa_accept_range(base, P, it, c0, c1, lane, 64);
            __syncthreads();
#undef NFA_OPAQUE
        }
        nfa_finish_range(base, P, c0, c1, lane, 64);
        __syncthreads();
        t0 = (long long)wall_clock64();
    }
}
#endif      // SSLAM_NFA_STAGE_ONLY
