// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// CPU restatement of the un-vendored OpenCV 3.4 leaf functions that the
// reference's hot path calls.  OpenCV is NOT present in
// /root/reference nor installed here, so these follow the published algorithms
// of opencv 3.4.x (modules/features2d/src/fast.cpp, fast_score.cpp,
// modules/imgproc/src/resize.cpp, smooth.cpp, deriv.cpp, copy.cpp,
// modules/core/src/mathfuncs_core.simd.hpp) as restated from their published
// source.  PARITY UNPINNED FOR THIS FILE: the reference ships no tests/golden vectors and the
// real library cannot be run here; this file *defines* the behaviour the HIP
// path must match bit-for-bit.  (The reference's OWN code around these leaves is pinned since round 4:
// oracle/ref_pin compiles src/ORBextractor.cc unmodified against a stub cv:: layer that forwards to
// this header, and the result equals orb_oracle.cpp byte for byte.)  What could be checked against an independent
// implementation offline IS checked (tests/test_oracle_cpu.py): the FAST-9/16 corner set,
// cornerScore (= largest passing threshold) and the intensity-centroid orientation against
// scikit-image 0.18.3 (fixtures in tests/golden/skimage_fast_orient.npz), Sobel 3x3 and the
// LSD NFA against scipy.  Resize, GaussianBlur, NMS, LSD region logic and LBD have no such
// counterpart here and stay unpinned.  Call sites in the reference:
//   src/ORBextractor.cc:81,103 (cvRound, fastAtan2), :809,814 (FAST),
//   :1086 (GaussianBlur), :1120 (resize), :1122,1127 (copyMakeBorder).
//
// Determinism decisions:
//   D4  no FMA contraction anywhere (build with -ffp-contract=off)
//   D5  cosf/sinf/atan2f := correctly rounded float of the real function,
//       evaluated as (float)f((double)x)
//   D6  8-bit GaussianBlur := OpenCV's bit-exact fixed-point path (>=3.4.1 with the
//       error-diffused 8.8 kernel whose taps sum to exactly 256)
//   D7  LSD's internal 0.8x resize := INTER_LINEAR_EXACT (8.8 fixed point)
#pragma once
#include <cstdint>
#include <cmath>
#include <cfloat>
#include <cstring>
#include <vector>
#include <algorithm>

namespace orc {

struct Img8 {                 // simple owning 8-bit image
    int w = 0, h = 0;
    std::vector<uint8_t> d;
    Img8() {}
    Img8(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_) {}
    uint8_t* row(int y) { return d.data() + (size_t)y * w; }
    const uint8_t* row(int y) const { return d.data() + (size_t)y * w; }
    uint8_t at(int y, int x) const { return d[(size_t)y * w + x]; }
};

// cvRound: round-half-to-even (SSE cvtss2si / lrint).  A.2
static inline int cv_round(double v) { return (int)std::lrint(v); }
static inline int cv_roundf(float v) { return (int)std::lrintf(v); }
static inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }

// D5 helpers
static inline float cr_cosf(float x) { return (float)std::cos((double)x); }
static inline float cr_sinf(float x) { return (float)std::sin((double)x); }
static inline float cr_atan2f(float y, float x) { return (float)std::atan2((double)y, (double)x); }

// cv::fastAtan2 (scalar atan_f32), degrees in [0,360).  A.3
static inline float fast_atan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = std::fabs(x), ay = std::fabs(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// BORDER_REFLECT_101 index map.  A.4
static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; }
    return i;
}

// cv::copyMakeBorder(src, dst, b,b,b,b, BORDER_REFLECT_101)
static inline Img8 copy_make_border101(const Img8& s, int b) {
    Img8 o(s.w + 2 * b, s.h + 2 * b);
    for (int y = 0; y < o.h; ++y) {
        const uint8_t* sr = s.row(reflect101(y - b, s.h));
        uint8_t* dr = o.row(y);
        for (int x = 0; x < o.w; ++x) dr[x] = sr[reflect101(x - b, s.w)];
    }
    return o;
}

// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for CV_8UC1: 11-bit fixed-point
// coefficients, horizontal pass to int32, vertical pass with the 8u special-case
// rounding.  A.5
static inline Img8 resize_linear_8u(const Img8& s, int dw, int dh) {
    Img8 o(dw, dh);
    const double inv_scale_x = (double)dw / s.w, inv_scale_y = (double)dh / s.h;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= s.w - 1) { fx = 0; sx = s.w - 1; }
        xofs[dx] = sx;
        float c0 = 1.f - fx, c1 = fx;
        ialpha[dx * 2] = (short)cv_roundf(c0 * 2048);
        ialpha[dx * 2 + 1] = (short)cv_roundf(c1 * 2048);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        float c0 = 1.f - fy, c1 = fy;
        ibeta[dy * 2] = (short)cv_roundf(c0 * 2048);
        ibeta[dy * 2 + 1] = (short)cv_roundf(c1 * 2048);
    }
    std::vector<int> r0(dw), r1(dw);
    for (int dy = 0; dy < dh; ++dy) {
        int sy0 = std::min(std::max(yofs[dy], 0), s.h - 1);
        int sy1 = std::min(std::max(yofs[dy] + 1, 0), s.h - 1);
        const uint8_t *S0 = s.row(sy0), *S1 = s.row(sy1);
        for (int dx = 0; dx < dw; ++dx) {
            int sx = xofs[dx], sx1 = std::min(sx + 1, s.w - 1);
            int a0 = ialpha[dx * 2], a1 = ialpha[dx * 2 + 1];
            r0[dx] = S0[sx] * a0 + S0[sx1] * a1;
            r1[dx] = S1[sx] * a0 + S1[sx1] * a1;
        }
        int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t* D = o.row(dy);
        for (int dx = 0; dx < dw; ++dx)
            D[dx] = (uint8_t)((((b0 * (r0[dx] >> 4)) >> 16) + ((b1 * (r1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    return o;
}

// D7: cv::resize(..., INTER_LINEAR_EXACT) for CV_8UC1 with fx=fy given (dsize
// derived by saturate_cast<int>(ssize*inv_scale)): 8.8 fixed-point coefficients,
// 8.8 intermediate, 16.16 vertical accumulate, round to nearest.
static inline Img8 resize_linear_exact_8u(const Img8& s, double inv_scale_x, double inv_scale_y) {
    int dw = cv_round(s.w * inv_scale_x), dh = cv_round(s.h * inv_scale_y);
    Img8 o(dw, dh);
    auto coeffs = [](double inv_scale, int ssz, int dsz, std::vector<int>& ofs, std::vector<int>& c1) {
        double scale = 1.0 / inv_scale;
        ofs.assign(dsz, 0); c1.assign(dsz, 0);
        for (int v = 0; v < dsz; ++v) {
            double fv = scale * ((double)v + 0.5) - 0.5;
            int iv = cv_floor(fv);
            if (iv >= 0 && ssz > 1) {
                if (iv < ssz - 1) { ofs[v] = iv; c1[v] = cv_round((fv - iv) * 256.0); }
                else { ofs[v] = ssz - 1; c1[v] = 0; }       // replicate last
            } else { ofs[v] = 0; c1[v] = 0; }               // replicate first
        }
    };
    std::vector<int> xo, xc, yo, yc;
    coeffs(inv_scale_x, s.w, dw, xo, xc);
    coeffs(inv_scale_y, s.h, dh, yo, yc);
    std::vector<int> r0(dw), r1(dw);
    for (int dy = 0; dy < dh; ++dy) {
        const uint8_t* S0 = s.row(yo[dy]);
        const uint8_t* S1 = s.row(std::min(yo[dy] + 1, s.h - 1));
        for (int dx = 0; dx < dw; ++dx) {
            int sx = xo[dx], sx1 = std::min(sx + 1, s.w - 1), c = xc[dx];
            r0[dx] = S0[sx] * (256 - c) + S0[sx1] * c;   // 8.8
            r1[dx] = S1[sx] * (256 - c) + S1[sx1] * c;
        }
        int c = yc[dy];
        uint8_t* D = o.row(dy);
        for (int dx = 0; dx < dw; ++dx) {
            uint32_t v = (uint32_t)r0[dx] * (256 - c) + (uint32_t)r1[dx] * c;   // 16.16
            D[dx] = (uint8_t)((v + (1u << 15)) >> 16);
        }
    }
    return o;
}

// D6: bit-exact 8.8 fixed-point Gaussian taps with error diffusion so that the
// taps sum to exactly 256 (OpenCV getGaussianKernelFixedPoint_ED).
static inline std::vector<int> gauss_taps_q8(int n, double sigma) {
    std::vector<double> k(n);
    double sum = 0;
    double scale2x = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; ++i) {
        double x = i - (n - 1) * 0.5;
        k[i] = std::exp(scale2x * x * x);
        sum += k[i];
    }
    for (int i = 0; i < n; ++i) k[i] /= sum;
    std::vector<int> t(n);
    double err = 0;
    int64_t isum = 0;
    int n2 = n / 2;
    for (int i = 0; i < n2; ++i) {
        double adj = k[i] * 256.0 + err;
        int v = cv_round(adj);
        err = adj - v;
        t[i] = v; t[n - 1 - i] = v;
        isum += v;
    }
    t[n2] = (int)(256 - 2 * isum);
    return t;
}

// The ALTERNATIVE of decision D6 (selectable: orc_set_gauss_variant(1), sslam_orb_set_blur_variant(orb, 1)): OpenCV 3.4.0 -- the version the
// reference's README.md:22 names -- has no bit-exact 8-bit Gaussian yet; its separable filter engine (filter.cpp, createSeparableLinearFilter,
// bits = 8) takes the FLOAT kernel of getGaussianKernel(n, sigma, CV_32F), scales it by 256 and rounds EACH tap, so the taps need not sum to 256:
// sigma 2, n 7 gives 18 34 49 55 49 34 18 (sum 257) against the error-diffused 18 34 48 56 48 34 18.  Same 16.16 accumulation and rounding, result
// saturated to 255 (the sum can reach 257 / 256 of a flat area).  UPSTREAM-RECALL like the rest of this file.
static inline std::vector<int> gauss_taps_340(int n, double sigma) {
    std::vector<float> k(n); double sum = 0; const double s2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; k[i] = (float)std::exp(s2 * x * x); sum += k[i]; }
    sum = 1. / sum;
    std::vector<int> t(n);
    for (int i = 0; i < n; ++i) { k[i] = (float)(k[i] * sum); t[i] = cv_round(k[i] * 256.0); }
    return t;
}

// cv::GaussianBlur(src8u, ksize n x n, sigma, BORDER_REFLECT_101), D6 variant:
// horizontal 8.8 accumulate, vertical 16.16 accumulate, round-to-nearest.  A.6
static inline Img8 gaussian_blur_8u_taps(const Img8& s, const std::vector<int>& t) {
    const int n = (int)t.size(), r = n / 2;
    std::vector<uint32_t> tmp((size_t)s.w * s.h);
    for (int y = 0; y < s.h; ++y) {
        const uint8_t* S = s.row(y);
        for (int x = 0; x < s.w; ++x) {
            uint32_t acc = 0;
            for (int k = -r; k <= r; ++k) acc += (uint32_t)S[reflect101(x + k, s.w)] * t[k + r];
            tmp[(size_t)y * s.w + x] = acc;      // <= 255 * sum(taps)
        }
    }
    Img8 o(s.w, s.h);
    for (int y = 0; y < s.h; ++y) {
        uint8_t* D = o.row(y);
        for (int x = 0; x < s.w; ++x) {
            uint32_t acc = 0;
            for (int k = -r; k <= r; ++k)
                acc += tmp[(size_t)reflect101(y + k, s.h) * s.w + x] * t[k + r];
            D[x] = (uint8_t)std::min<uint32_t>((acc + (1u << 15)) >> 16, 255u);
        }
    }
    return o;
}
static inline Img8 gaussian_blur_8u(const Img8& s, int n, double sigma, int variant = 0) {
    return gaussian_blur_8u_taps(s, variant == 1 ? gauss_taps_340(n, sigma) : gauss_taps_q8(n, sigma));
}

// ---- FAST-9/16 -----------------------------------------------------------
struct FastKp { int x, y, score; };

static const int kFastRing[16][2] = {{0,3},{1,3},{2,2},{3,1},{3,0},{3,-1},{2,-2},{1,-3},
                                     {0,-3},{-1,-3},{-2,-2},{-3,-1},{-3,0},{-3,1},{-2,2},{-1,3}};

// cornerScore<16> (fast_score.cpp).  A.1
static inline int fast_corner_score16(const uint8_t* p, int stride, int threshold) {
    const int K = 8, N = K * 3 + 1;
    int v = p[0];
    short d[N];
    for (int k = 0; k < N; ++k) {
        const int* o = kFastRing[k % 16];
        d[k] = (short)(v - p[o[0] + o[1] * stride]);
    }
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]); a = std::min(a, (int)d[k + 5]);
        a = std::min(a, (int)d[k + 6]); a = std::min(a, (int)d[k + 7]);
        a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]); b = std::max(b, (int)d[k + 4]);
        b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]); b = std::max(b, (int)d[k + 7]);
        b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

// cv::FAST(view, kps, threshold, nonmaxSuppression=true), TYPE_9_16, on the view
// [x0,x1) x [y0,y1) of img.  Emits view-relative coordinates in raster order. A.1
static inline void fast9_view(const Img8& img, int x0, int y0, int x1, int y1,
                              int threshold, std::vector<FastKp>& out, std::vector<uint8_t>* score_map_out = nullptr) {
    const int cols = x1 - x0, rows = y1 - y0, stride = img.w;
    out.clear();
    if (cols < 7 || rows < 7) return;
    threshold = std::min(std::max(threshold, 0), 255);
    std::vector<uint8_t> sc((size_t)cols * rows, 0);       // 0 = not a corner at this threshold
    for (int i = 3; i < rows - 3; ++i) {
        for (int j = 3; j < cols - 3; ++j) {
            const uint8_t* p = img.d.data() + (size_t)(y0 + i) * stride + (x0 + j);
            int v = p[0];
            // fast.cpp's early rejection: a 9-arc needs one of every opposite ring pair on its side of the threshold
            {
                const int lo = v - threshold, hi = v + threshold;
                auto cls = [&](int k) { int x = p[kFastRing[k][0] + kFastRing[k][1] * stride]; return (x < lo ? 1 : 0) | (x > hi ? 2 : 0); };
                int d = cls(0) | cls(8);
                if (d == 0) continue;
                d &= cls(2) | cls(10); d &= cls(4) | cls(12); d &= cls(6) | cls(14);
                if (d == 0) continue;
                d &= cls(1) | cls(9); d &= cls(3) | cls(11); d &= cls(5) | cls(13); d &= cls(7) | cls(15);
                if (d == 0) continue;
            }
            int cb = 0, cd = 0;
            bool corner = false;
            for (int k = 0; k < 25 && !corner; ++k) {
                const int* o = kFastRing[k % 16];
                int x = p[o[0] + o[1] * stride];
                if (x > v + threshold) { if (++cb > 8) corner = true; } else cb = 0;
                if (x < v - threshold) { if (++cd > 8) corner = true; } else cd = 0;
            }
            if (corner) sc[(size_t)i * cols + j] = (uint8_t)fast_corner_score16(p, stride, threshold);
        }
    }
    if (score_map_out) *score_map_out = sc;              // test tap: the score of every pixel that passed the segment test, before NMS
    // A pixel that passed the segment test always has score >= threshold; with
    // threshold >= 1 a stored 0 therefore means "not a corner".
    for (int i = 3; i < rows - 3; ++i)
        for (int j = 3; j < cols - 3; ++j) {
            int s = sc[(size_t)i * cols + j];
            if (s == 0 && threshold > 0) continue;
            // need the segment test to have passed: score>=threshold identifies it
            if (s < threshold) continue;
            const uint8_t* c = &sc[(size_t)i * cols + j];
            if (s > c[-1] && s > c[1] && s > c[-cols - 1] && s > c[-cols] && s > c[-cols + 1] &&
                s > c[cols - 1] && s > c[cols] && s > c[cols + 1])
                out.push_back({j, i, s});
        }
}

// cv::Sobel(src8u, dst, CV_16S, dx, dy, 3) with BORDER_REFLECT_101 (exact ints)
static inline void sobel3_s16(const Img8& s, std::vector<int16_t>& gx, std::vector<int16_t>& gy) {
    gx.assign((size_t)s.w * s.h, 0); gy.assign((size_t)s.w * s.h, 0);
    for (int y = 0; y < s.h; ++y) {
        const uint8_t* r0 = s.row(reflect101(y - 1, s.h));
        const uint8_t* r1 = s.row(y);
        const uint8_t* r2 = s.row(reflect101(y + 1, s.h));
        for (int x = 0; x < s.w; ++x) {
            int xm = reflect101(x - 1, s.w), xp = reflect101(x + 1, s.w);
            gx[(size_t)y * s.w + x] = (int16_t)((r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]));
            gy[(size_t)y * s.w + x] = (int16_t)((r2[xm] - r0[xm]) + 2 * (r2[x] - r0[x]) + (r2[xp] - r0[xp]));
        }
    }
}

}  // namespace orc
