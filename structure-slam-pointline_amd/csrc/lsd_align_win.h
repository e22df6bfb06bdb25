// rect_nfa's isAligned as integer intervals.
//
// rect_nfa (OpenCV 3.4 lsd.cpp, reached from /root/reference/src/ExtractLineSegment.cpp:38-40; restated at oracle/lsd_oracle.cpp:68-76)
// tests every pixel of a candidate rectangle with
//     n = |theta - a|;  if (n > 3pi/2) n = |n - 2pi|;  aligned = n <= prec            (a = the pixel's level-line angle in radians, double)
// where a = (double)aDeg * DEG2RAD for the stored fp32 angle aDeg in [0, 360).  theta and prec are fixed per (rectangle, candidate), and
// every step is monotone in aDeg: fl(aDeg * D) is non-decreasing, so g(aDeg) = fl(theta - fl(aDeg * D)) is non-increasing, and for
// prec < pi/2 the predicate is  -prec <= g <= prec   or   -prec <= fl(|g| - 2pi) <= prec  -- three windows of g (around 0, +2pi, -2pi),
// each the preimage of an interval under a monotone map, i.e. an INTERVAL OF FLOAT BIT PATTERNS of aDeg (non-negative floats order like
// their bits).  g spans 2pi and the windows are 2pi - 2 prec apart, so at most two of them are non-empty.  The end points are found
// with the fp64 expression itself (an estimate from the inverse map, then galloping + bisection on the bit pattern: 2-3 evaluations per
// end point), once per (rectangle, candidate); the per-pixel test is then two integer compares per window on the 4-byte angle instead of
// eight fp64 instructions (round 2: 8 of k_nfa_count's ~13 vector instructions per pixel slot).
// sslam_selftest_align_windows (lines.hip) and tests/test_align_windows_cpu.py (this header compiled by g++) compare windows and
// predicate over every angle the gradient table can produce, random bit patterns and the neighbours of every end point.
#pragma once
#ifdef __HIPCC__
#define SSLAM_HD __host__ __device__ __forceinline__
#define SSLAM_HD_OUTLINE __host__ __device__ __attribute__((noinline))
#else
#include <cmath>
#include <cstring>
#define SSLAM_HD inline
#define SSLAM_HD_OUTLINE inline
#endif

namespace alnwin {

constexpr double A_PI = 3.14159265358979323846, A_D2R = A_PI / 180, A_2PI = 2 * A_PI, A_3_2PI = (3 * A_PI) / 2;
constexpr int BMAX = 0x43B40000;            // bits of 360.0f: the stored angles are below it; NOTDEF (|-1024.f| = 0x44800000) is above

SSLAM_HD float bits_to_float(int b) {
#ifdef __HIP_DEVICE_COMPILE__
    return __int_as_float(b);
#else
    float f; memcpy(&f, &b, 4); return f;
#endif
}
SSLAM_HD int float_to_bits(float f) {
#ifdef __HIP_DEVICE_COMPILE__
    return __float_as_int(f);
#else
    int b; memcpy(&b, &f, 4); return b;
#endif
}

// the reference predicate (fold at 3pi/2), for the tests
SSLAM_HD bool aligned_ref(float aDeg, double theta, double prec) {
    double n = fabs(theta - (double)aDeg * A_D2R);
    if (n > A_3_2PI) n = fabs(n - A_2PI);
    return n <= prec;
}

// boundary predicates, each monotone false -> true in the bit pattern b of aDeg (kinds 0/1: window around g = 0, 2/3: g = +2pi, 4/5: g = -2pi;
// even kinds: "at or past the lower end", odd kinds: "past the upper end")
template <int KIND>
SSLAM_HD bool bq(int b, double theta, double prec) {
    const double g = theta - (double)bits_to_float(b) * A_D2R;
    if (KIND == 0) return g <= prec;
    if (KIND == 1) return g < -prec;
    if (KIND == 2) return g - A_2PI <= prec;
    if (KIND == 3) return g - A_2PI < -prec;
    if (KIND == 4) return -g - A_2PI >= -prec;
    return -g - A_2PI > prec;
}

// smallest b in [0, BMAX] with bq<KIND>(b), BMAX + 1 if there is none; est = a guess (any value: only the number of evaluations depends on it)
template <int KIND>
SSLAM_HD int first_true(double theta, double prec, double estDeg) {
    const float ef = (float)fmin(fmax(estDeg, 0.0), 360.0);
    const int est = float_to_bits(ef);
    int lo, hi;                                  // bq(lo) false or lo == -1;  bq(hi) true or hi == BMAX + 1
    if (bq<KIND>(est, theta, prec)) {
        hi = est; lo = -1;
        for (int step = 1; hi - step >= 0; step = step < (1 << 29) ? step << 1 : step) {
            const int c = hi - step;
            if (bq<KIND>(c, theta, prec)) hi = c; else { lo = c; break; }
        }
    } else {
        lo = est; hi = BMAX + 1;
        for (int step = 1; lo + step <= BMAX; step = step < (1 << 29) ? step << 1 : step) {
            const int c = lo + step;
            if (!bq<KIND>(c, theta, prec)) lo = c; else { hi = c; break; }
        }
    }
    while (hi - lo > 1) {
        const int mid = lo + ((hi - lo) >> 1);
        if (bq<KIND>(mid, theta, prec)) hi = mid; else lo = mid;
    }
    return hi;
}

// The aligned set of (theta, prec), prec < pi/2, as up to two closed intervals [lo0, hi0], [lo1, hi1] of bit patterns (n = how many are
// non-empty; an empty one is lo = hi + 1 above every angle).  ok = false if all three windows came out non-empty, which the argument above
// excludes (callers flag the frame instead of counting wrong).
// (Six searches inlined per call site: k_nfa_count calls it from ONE place, a loop over the tolerances -- two call sites doubled its code.)
struct Win { int n, lo0, hi0, lo1, hi1, ok; };
SSLAM_HD Win windows(double theta, double prec) {
    Win w; w.n = 0; w.lo0 = w.lo1 = BMAX + 1; w.hi0 = w.hi1 = BMAX; w.ok = 1;
    const double slack = 1e-6;                                // g lies in [theta - 2pi(1 + 1e-15), theta]: a window that g cannot reach is skipped without evaluating
    auto put = [&](int l, int h1) {
        if (l >= h1) return;
        if (w.n == 0) { w.lo0 = l; w.hi0 = h1 - 1; } else if (w.n == 1) { w.lo1 = l; w.hi1 = h1 - 1; } else w.ok = 0;
        ++w.n;
    };
    if (theta >= -prec - slack && theta - A_2PI <= prec + slack)
        put(first_true<0>(theta, prec, (theta - prec) / A_D2R), first_true<1>(theta, prec, (theta + prec) / A_D2R));
    if (theta >= A_2PI - prec - slack)
        put(first_true<2>(theta, prec, (theta - A_2PI - prec) / A_D2R), first_true<3>(theta, prec, (theta - A_2PI + prec) / A_D2R));
    if (theta <= prec + slack)
        put(first_true<4>(theta, prec, (theta + A_2PI - prec) / A_D2R), first_true<5>(theta, prec, (theta + A_2PI + prec) / A_D2R));
    if (w.n > 2) w.n = 2;
    return w;
}
SSLAM_HD bool windows(double theta, double prec, int& n, int (&lo)[2], int (&hi)[2]) {
    const Win w = windows(theta, prec);
    n = w.n; lo[0] = w.lo0; hi[0] = w.hi0; lo[1] = w.lo1; hi[1] = w.hi1;
    return w.ok != 0;
}

}  // namespace alnwin
