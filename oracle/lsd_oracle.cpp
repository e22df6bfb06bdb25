// ORACLE — TEST INFRASTRUCTURE ONLY (see cvleaf.h header).  PARITY UNPINNED.
//
// CPU restatement of the line detector behind LineSegment::ExtractLineSegment
// (reference src/ExtractLineSegment.cpp:18-69):
//   cv::line_descriptor::LSDDetector::detect(img, keylines, scale, numOctaves)   (:38-40)
//     -> cv::createLineSegmentDetector(LSD_REFINE_ADV)->detect(octave image)
// Neither lives in /root/reference (un-vendored OpenCV 3.4 imgproc/lsd.cpp and
// opencv_contrib line_descriptor/src/LSDDetector.cpp); they are restated here from
// their published algorithm (von Gioi et al. LSD as ported in OpenCV; SURVEY.md
// Appendix A.7-A.8), including OpenCV's own quirks (integer edge slopes and the
// `tailp->p.x` comparisons in rect_nfa).
//
// Determinism decisions: D2 seed order = descending gradient bin, raster order
// inside a bin (stable); D4 no FMA; D5 cosf/sinf/atan2f correctly rounded;
// D6 fixed-point GaussianBlur; D7 INTER_LINEAR_EXACT for the 0.8x rescale.
#include "oracle.h"
#include "cvleaf.h"
#include "lines_types.h"

namespace orc {
int g_lsdResize = 0;        // 0: decision D7 (INTER_LINEAR_EXACT); 1: INTER_LINEAR (error bar only)
int g_lsdSeedSort = 0;      // 0: decision D2 (stable: raster order inside a bin); 1: upstream's std::sort (error bar only)
int g_lsdNfaVariant = 1;    // decision D11 -- nfa()'s first term.  1 (DEFAULT since round 5): `(double(n) + 1)`, as imgproc/src/lsd.cpp is recalled to read by two independent
                            // recollections (the round-4 review's and the builder's, the latter including the comment block above the line: "bincoef(n,k) = gamma(n+1) /
                            // ( gamma(k+1) * gamma(n-k+1) ).  We use this to compute the first term.  Actually the log of it." followed by a first term WITHOUT the log_gamma);
                            // 0: log_gamma(n + 1), the binomial coefficient of von Gioi's lsd.c (the default of rounds 1-4).  UPSTREAM-RECALL either way: orc_set_lsd_nfa_variant();
                            // the library's switch is sslam_lines_set_nfa_variant().


static const double NOTDEF = -1024.0;
static const double M_3_2_PI = (3 * M_PI) / 2, M_2__PI = 2 * M_PI, DEG_TO_RADS = M_PI / 180;
static const double RELATIVE_ERROR_FACTOR = 100.0;

struct RegionPoint { int x, y; double angle, modgrad; };
struct Rect { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; };

static inline double distSq(double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); }
static inline double dist(double x1, double y1, double x2, double y2) { return std::sqrt(distSq(x1, y1, x2, y2)); }
static inline double angle_diff_signed(double a, double b) {
    double diff = a - b;
    while (diff <= -M_PI) diff += M_2__PI;
    while (diff > M_PI) diff -= M_2__PI;
    return diff;
}
static inline double angle_diff(double a, double b) { return std::fabs(angle_diff_signed(a, b)); }
static inline bool double_equal(double a, double b) {
    if (a == b) return true;
    double abs_diff = std::fabs(a - b), aa = std::fabs(a), bb = std::fabs(b);
    double abs_max = aa > bb ? aa : bb;
    if (abs_max < DBL_MIN) abs_max = DBL_MIN;
    return (abs_diff / abs_max) <= (RELATIVE_ERROR_FACTOR * DBL_EPSILON);
}
static inline double log_gamma_windschitl(double x) {
    return 0.918938533204673 + (x - 0.5) * std::log(x) - x + 0.5 * x * std::log(x * std::sinh(1 / x) + 1 / (810.0 * std::pow(x, 6.0)));
}
static inline double log_gamma_lanczos(double x) {
    static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
    double a = (x + 0.5) * std::log(x + 5.5) - (x + 5.5);
    double b = 0;
    for (int n = 0; n < 7; ++n) { a -= std::log(x + double(n)); b += q[n] * std::pow(x, double(n)); }
    return a + std::log(b);
}
static inline double log_gamma(double x) { return x > 15 ? log_gamma_windschitl(x) : log_gamma_lanczos(x); }

struct Lsd {
    // LSD_REFINE_ADV defaults of createLineSegmentDetector
    const double SCALE = 0.8, SIGMA_SCALE = 0.6, QUANT = 2.0, ANG_TH = 22.5, LOG_EPS = 0, DENSITY_TH = 0.7;
    const int N_BINS = 1024;
    int w = 0, h = 0;
    double LOG_NT = 0;
    Img8 scaled;
    std::vector<double> angles, modgrad;
    std::vector<uint8_t> used;
    std::vector<int> order;         // pixel indices (y*w+x), D2 order

    bool isAligned(int x, int y, double theta, double prec) const {
        if (x < 0 || y < 0 || x >= w || y >= h) return false;
        const double a = angles[(size_t)y * w + x];
        if (a == NOTDEF) return false;
        double n_theta = theta - a;
        if (n_theta < 0) n_theta = -n_theta;
        if (n_theta > M_3_2_PI) { n_theta -= M_2__PI; if (n_theta < 0) n_theta = -n_theta; }
        return n_theta <= prec;
    }

    void ll_angle(double threshold) {
        angles.assign((size_t)w * h, NOTDEF);
        modgrad.assign((size_t)w * h, 0.0);
        double max_grad = -1;
        for (int y = 0; y < h - 1; ++y) {
            const uint8_t* r0 = scaled.row(y); const uint8_t* r1 = scaled.row(y + 1);
            for (int x = 0; x < w - 1; ++x) {
                int DA = r1[x + 1] - r0[x], BC = r0[x + 1] - r1[x];
                int gx = DA + BC, gy = DA - BC;
                double norm = std::sqrt((gx * gx + gy * gy) / 4.0);
                modgrad[(size_t)y * w + x] = norm;
                if (norm <= threshold) angles[(size_t)y * w + x] = NOTDEF;
                else {
                    angles[(size_t)y * w + x] = fast_atan2(float(gx), float(-gy)) * DEG_TO_RADS;
                    if (norm > max_grad) max_grad = norm;
                }
            }
        }
        double bin_coef = (max_grad > 0) ? double(N_BINS - 1) / max_grad : 0;
        order.clear();
        if (g_lsdSeedSort == 1) {
            // Decision D2's alternative, for its error bar only (orc_set_lsd_seed_sort(1)): upstream (UPSTREAM-RECALL, imgproc/src/lsd.cpp ll_angle) fills a vector of
            // {point, bin} in raster order and calls std::sort with `a.norm > b.norm` -- an UNSTABLE introsort, so the order inside a bin is whatever libstdc++'s
            // algorithm leaves (median-of-three quicksort to depth 2 lg n, insertion sort below 16 elements; unchanged since GCC 4).  The library cannot follow it:
            // the permutation is the result of ~17 n dependent comparisons per frame.
            struct NormPoint { int x, y, norm; };
            std::vector<NormPoint> pts; pts.reserve((size_t)(w - 1) * (h - 1));
            for (int y = 0; y < h - 1; ++y)
                for (int x = 0; x < w - 1; ++x) {
                    int i = int(modgrad[(size_t)y * w + x] * bin_coef);
                    pts.push_back({x, y, i});
                }
            std::sort(pts.begin(), pts.end(), [](const NormPoint& a, const NormPoint& b) { return a.norm > b.norm; });
            for (const NormPoint& q : pts) order.push_back(q.y * w + q.x);
        } else {
        // counting sort, descending bin, raster order inside a bin (D2)
        std::vector<std::vector<int>> bins(N_BINS);
        for (int y = 0; y < h - 1; ++y)
            for (int x = 0; x < w - 1; ++x) {
                int i = int(modgrad[(size_t)y * w + x] * bin_coef);
                if (i < 0) i = 0;
                if (i >= N_BINS) i = N_BINS - 1;
                bins[i].push_back(y * w + x);
            }
        for (int b = N_BINS - 1; b >= 0; --b) order.insert(order.end(), bins[b].begin(), bins[b].end());
        }
    }

    void region_grow(int sx, int sy, std::vector<RegionPoint>& reg, double& reg_angle, double prec) {
        reg.clear();
        RegionPoint seed; seed.x = sx; seed.y = sy;
        reg_angle = angles[(size_t)sy * w + sx];
        seed.angle = reg_angle; seed.modgrad = modgrad[(size_t)sy * w + sx];
        reg.push_back(seed);
        float sumdx = float(std::cos(reg_angle)), sumdy = float(std::sin(reg_angle));
        used[(size_t)sy * w + sx] = 1;
        for (size_t i = 0; i < reg.size(); ++i) {
            const int px = reg[i].x, py = reg[i].y;
            int xx_min = std::max(px - 1, 0), xx_max = std::min(px + 1, w - 1);
            int yy_min = std::max(py - 1, 0), yy_max = std::min(py + 1, h - 1);
            for (int yy = yy_min; yy <= yy_max; ++yy)
                for (int xx = xx_min; xx <= xx_max; ++xx) {
                    uint8_t& is_used = used[(size_t)yy * w + xx];
                    if (is_used != 1 && isAligned(xx, yy, reg_angle, prec)) {
                        const double angle = angles[(size_t)yy * w + xx];
                        is_used = 1;
                        RegionPoint rp; rp.x = xx; rp.y = yy; rp.modgrad = modgrad[(size_t)yy * w + xx]; rp.angle = angle;
                        reg.push_back(rp);
                        sumdx += cr_cosf(float(angle));       // D5
                        sumdy += cr_sinf(float(angle));
                        reg_angle = fast_atan2(sumdy, sumdx) * DEG_TO_RADS;
                    }
                }
        }
    }

    double get_theta(const std::vector<RegionPoint>& reg, double x, double y, double reg_angle, double prec) const {
        double Ixx = 0, Iyy = 0, Ixy = 0;
        for (size_t i = 0; i < reg.size(); ++i) {
            const double regx = reg[i].x, regy = reg[i].y, weight = reg[i].modgrad;
            double dx = regx - x, dy = regy - y;
            Ixx += dy * dy * weight; Iyy += dx * dx * weight; Ixy -= dx * dy * weight;
        }
        double lambda = 0.5 * (Ixx + Iyy - std::sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
        double theta = (std::fabs(Ixx) > std::fabs(Iyy)) ? double(fast_atan2(float(lambda - Ixx), float(Ixy)))
                                                         : double(fast_atan2(float(Ixy), float(lambda - Iyy)));
        theta *= DEG_TO_RADS;
        if (angle_diff(theta, reg_angle) > prec) theta += M_PI;
        return theta;
    }

    void region2rect(const std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec) const {
        double x = 0, y = 0, sum = 0;
        for (size_t i = 0; i < reg.size(); ++i) {
            const double weight = reg[i].modgrad;
            x += double(reg[i].x) * weight; y += double(reg[i].y) * weight; sum += weight;
        }
        x /= sum; y /= sum;
        double theta = get_theta(reg, x, y, reg_angle, prec);
        double dx = std::cos(theta), dy = std::sin(theta);
        double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
        for (size_t i = 0; i < reg.size(); ++i) {
            double regdx = double(reg[i].x) - x, regdy = double(reg[i].y) - y;
            double l = regdx * dx + regdy * dy, ww = -regdx * dy + regdy * dx;
            if (l > l_max) l_max = l; else if (l < l_min) l_min = l;
            if (ww > w_max) w_max = ww; else if (ww < w_min) w_min = ww;
        }
        rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy; rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
        rec.width = w_max - w_min; rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy; rec.prec = prec; rec.p = p;
        if (rec.width < 1.0) rec.width = 1.0;
    }

    bool reduce_region_radius(std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec, double density, double density_th) {
        double xc = double(reg[0].x), yc = double(reg[0].y);
        double radSq1 = distSq(xc, yc, rec.x1, rec.y1), radSq2 = distSq(xc, yc, rec.x2, rec.y2);
        double radSq = radSq1 > radSq2 ? radSq1 : radSq2;
        while (density < density_th) {
            radSq *= 0.75 * 0.75;
            for (size_t i = 0; i < reg.size(); ++i) {
                if (distSq(xc, yc, double(reg[i].x), double(reg[i].y)) > radSq) {
                    used[(size_t)reg[i].y * w + reg[i].x] = 0;
                    std::swap(reg[i], reg[reg.size() - 1]);
                    reg.pop_back();
                    --i;
                }
            }
            if (reg.size() < 2) return false;
            region2rect(reg, reg_angle, prec, p, rec);
            density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        }
        return true;
    }

    bool refine(std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec, double density_th) {
        double density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        if (density >= density_th) return true;
        double xc = double(reg[0].x), yc = double(reg[0].y);
        const double ang_c = reg[0].angle;
        double sum = 0, s_sum = 0;
        int n = 0;
        for (size_t i = 0; i < reg.size(); ++i) {
            used[(size_t)reg[i].y * w + reg[i].x] = 0;
            if (dist(xc, yc, reg[i].x, reg[i].y) < rec.width) {
                double ang_d = angle_diff_signed(reg[i].angle, ang_c);
                sum += ang_d; s_sum += ang_d * ang_d; ++n;
            }
        }
        double mean_angle = sum / double(n);
        double tau = 2.0 * std::sqrt((s_sum - 2.0 * mean_angle * sum) / double(n) + mean_angle * mean_angle);
        int sx = reg[0].x, sy = reg[0].y;
        region_grow(sx, sy, reg, reg_angle, tau);
        if (reg.size() < 2) return false;
        region2rect(reg, reg_angle, prec, p, rec);
        density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        if (density < density_th) return reduce_region_radius(reg, reg_angle, prec, p, rec, density, density_th);
        return true;
    }

    double nfa(int n, int k, double p) const {
        if (n == 0 || k == 0) return -LOG_NT;
        if (n == k) return -LOG_NT - double(n) * std::log10(p);
        double p_term = p / (1 - p);
        const double first = g_lsdNfaVariant == 1 ? (double(n) + 1) : log_gamma(double(n) + 1);          // decision D11 (see g_lsdNfaVariant)
        double log1term = first - log_gamma(double(k) + 1) - log_gamma(double(n - k) + 1)
                          + double(k) * std::log(p) + double(n - k) * std::log(1.0 - p);
        double term = std::exp(log1term);
        if (double_equal(term, 0)) {
            if (k > n * p) return -log1term / M_LN10 - LOG_NT;
            else return -LOG_NT;
        }
        double bin_tail = term;
        double tolerance = 0.1;
        for (int i = k + 1; i <= n; ++i) {
            double bin_term = double(n - i + 1) / double(i);
            double mult_term = bin_term * p_term;
            term *= mult_term;
            bin_tail += term;
            if (bin_term < 1) {
                double err = term * ((1 - std::pow(mult_term, double(n - i + 1))) / (1 - mult_term) - 1);
                if (err < tolerance * std::fabs(-std::log10(bin_tail) - LOG_NT) * bin_tail) break;
            }
        }
        return -std::log10(bin_tail) - LOG_NT;
    }

    struct Edge { int x, y; bool taken; };

    double rect_nfa(const Rect& rec) const {
        int total_pts = 0, alg_pts = 0;
        double half_width = rec.width / 2.0;
        double dyhw = rec.dy * half_width, dxhw = rec.dx * half_width;
        Edge ox[4];
        ox[0] = {int(rec.x1 - dyhw), int(rec.y1 + dxhw), false};
        ox[1] = {int(rec.x2 - dyhw), int(rec.y2 + dxhw), false};
        ox[2] = {int(rec.x2 + dyhw), int(rec.y2 - dxhw), false};
        ox[3] = {int(rec.x1 + dyhw), int(rec.y1 - dxhw), false};
        // 4-element sort by (x, then y): insertion sort (a stable total order; ties are identical points)
        for (int i = 1; i < 4; ++i) {
            Edge e = ox[i]; int j = i - 1;
            while (j >= 0 && (ox[j].x > e.x || (ox[j].x == e.x && ox[j].y > e.y))) { ox[j + 1] = ox[j]; --j; }
            ox[j + 1] = e;
        }
        Edge* min_y = &ox[0]; Edge* max_y = &ox[0];
        for (unsigned i = 1; i < 4; ++i) {
            if (min_y->y > ox[i].y) min_y = &ox[i];
            if (max_y->y < ox[i].y) max_y = &ox[i];
        }
        min_y->taken = true;
        Edge* leftmost = 0;
        for (unsigned i = 0; i < 4; ++i) if (!ox[i].taken) { if (!leftmost) leftmost = &ox[i]; else if (leftmost->x > ox[i].x) leftmost = &ox[i]; }
        leftmost->taken = true;
        Edge* rightmost = 0;
        for (unsigned i = 0; i < 4; ++i) if (!ox[i].taken) { if (!rightmost) rightmost = &ox[i]; else if (rightmost->x < ox[i].x) rightmost = &ox[i]; }
        rightmost->taken = true;
        Edge* tailp = 0;
        for (unsigned i = 0; i < 4; ++i) if (!ox[i].taken) { if (!tailp) tailp = &ox[i]; else if (tailp->x > ox[i].x) tailp = &ox[i]; }
        tailp->taken = true;
        // integer divisions and the p.y-vs-p.x comparisons are OpenCV's own
        double flstep = (min_y->y != leftmost->y) ? (min_y->x - leftmost->x) / (min_y->y - leftmost->y) : 0;
        double slstep = (leftmost->y != tailp->x) ? (leftmost->x - tailp->x) / (leftmost->y - tailp->x) : 0;
        double frstep = (min_y->y != rightmost->y) ? (min_y->x - rightmost->x) / (min_y->y - rightmost->y) : 0;
        double srstep = (rightmost->y != tailp->x) ? (rightmost->x - tailp->x) / (rightmost->y - tailp->x) : 0;
        double lstep = flstep, rstep = frstep;
        double left_x = min_y->x, right_x = min_y->x;
        int min_iter = min_y->y, max_iter = max_y->y;
        for (int y = min_iter; y <= max_iter; ++y) {
            if (y < 0 || y >= h) continue;          // NB: also skips the edge stepping below (as upstream)
            for (int x = int(left_x); x <= int(right_x); ++x) {
                if (x < 0 || x >= w) continue;
                ++total_pts;
                if (isAligned(x, y, rec.theta, rec.prec)) ++alg_pts;
            }
            if (y >= leftmost->y) lstep = slstep;
            if (y >= rightmost->y) rstep = srstep;
            left_x += lstep; right_x += rstep;
        }
        return nfa(total_pts, alg_pts, rec.p);
    }

    double rect_improve(Rect& rec) const {
        double delta = 0.5, delta_2 = delta / 2.0;
        double log_nfa = rect_nfa(rec);
        if (log_nfa > LOG_EPS) return log_nfa;
        Rect r = rec;
        for (int n = 0; n < 5; ++n) {
            r.p /= 2; r.prec = r.p * M_PI;
            double v = rect_nfa(r);
            if (v > log_nfa) { log_nfa = v; rec = r; }
        }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (unsigned n = 0; n < 5; ++n) {
            if ((r.width - delta) >= 0.5) {
                r.width -= delta;
                double v = rect_nfa(r);
                if (v > log_nfa) { rec = r; log_nfa = v; }
            }
        }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (unsigned n = 0; n < 5; ++n) {
            if ((r.width - delta) >= 0.5) {
                r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2; r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2;
                r.width -= delta;
                double v = rect_nfa(r);
                if (v > log_nfa) { rec = r; log_nfa = v; }
            }
        }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (unsigned n = 0; n < 5; ++n) {
            if ((r.width - delta) >= 0.5) {
                r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2; r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2;
                r.width -= delta;
                double v = rect_nfa(r);
                if (v > log_nfa) { rec = r; log_nfa = v; }
            }
        }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (unsigned n = 0; n < 5; ++n) {
            if ((r.width - delta) >= 0.5) {
                r.p /= 2; r.prec = r.p * M_PI;
                double v = rect_nfa(r);
                if (v > log_nfa) { rec = r; log_nfa = v; }
            }
        }
        return log_nfa;
    }

    // LineSegmentDetectorImpl::flsd
    void detect(const Img8& image, std::vector<Seg4f>& lines) {
        lines.clear();
        const double prec = M_PI * ANG_TH / 180, p = ANG_TH / 180;
        const double rho = QUANT / std::sin(prec);
        const double sigma = SIGMA_SCALE / SCALE;
        const double sprec = 3;
        const unsigned hk = (unsigned)(std::ceil(sigma * std::sqrt(2 * sprec * std::log(10.0))));
        Img8 g = gaussian_blur_8u(image, 1 + 2 * hk, sigma, g_gaussVariant);          // D6 (sigma 0.75, n 7: both variants give 0 4 56 136 56 4 0)
        // D7; its alternative (orc_set_lsd_resize(1), error bar only): plain INTER_LINEAR, the 11-bit fixed-point resize of the ORB pyramid, with the same output size
        scaled = g_lsdResize == 1 ? resize_linear_8u(g, cv_round(g.w * SCALE), cv_round(g.h * SCALE)) : resize_linear_exact_8u(g, SCALE, SCALE);
        w = scaled.w; h = scaled.h;
        ll_angle(rho);
        LOG_NT = 5 * (std::log10(double(w)) + std::log10(double(h))) / 2 + std::log10(11.0);
        const size_t min_reg_size = size_t(-LOG_NT / std::log10(p));
        used.assign((size_t)w * h, 0);
        std::vector<RegionPoint> reg;
        for (size_t i = 0; i < order.size(); ++i) {
            const int idx = order[i], px = idx % w, py = idx / w;
            if (used[idx] == 0 && angles[idx] != NOTDEF) {
                double reg_angle;
                region_grow(px, py, reg, reg_angle, prec);
                if (reg.size() < min_reg_size) continue;
                Rect rec;
                region2rect(reg, reg_angle, prec, p, rec);
                if (!refine(reg, reg_angle, prec, p, rec, DENSITY_TH)) continue;
                double log_nfa = rect_improve(rec);
                if (log_nfa <= LOG_EPS) continue;
                rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
                rec.x1 /= SCALE; rec.y1 /= SCALE; rec.x2 /= SCALE; rec.y2 /= SCALE; rec.width /= SCALE;
                lines.push_back({float(rec.x1), float(rec.y1), float(rec.x2), float(rec.y2)});
            }
        }
    }
};

// LSDDetector::detectImpl (opencv_contrib line_descriptor/src/LSDDetector.cpp), numOctaves = 1
void lsd_detect_keylines(const Img8& image, std::vector<KeyLine>& keylines, std::vector<Seg4f>* raw) {
    keylines.clear();
    Lsd lsd;
    std::vector<Seg4f> segs;
    lsd.detect(image, segs);
    if (raw) *raw = segs;
    int class_counter = -1;
    const float octaveScale = 1.0f;      // pow((float)scale, 0)
    for (size_t k = 0; k < segs.size(); ++k) {
        float e[4] = {segs[k].x1, segs[k].y1, segs[k].x2, segs[k].y2};
        // checkLineExtremes
        if (e[0] < 0) e[0] = 0;
        if (e[0] >= image.w) e[0] = (float)image.w - 1.0f;
        if (e[2] < 0) e[2] = 0;
        if (e[2] >= image.w) e[2] = (float)image.w - 1.0f;
        if (e[1] < 0) e[1] = 0;
        if (e[1] >= image.h) e[1] = (float)image.h - 1.0f;
        if (e[3] < 0) e[3] = 0;
        if (e[3] >= image.h) e[3] = (float)image.h - 1.0f;
        KeyLine kl;
        kl.startPointX = e[0] * octaveScale; kl.startPointY = e[1] * octaveScale;
        kl.endPointX = e[2] * octaveScale; kl.endPointY = e[3] * octaveScale;
        kl.sPointInOctaveX = e[0]; kl.sPointInOctaveY = e[1]; kl.ePointInOctaveX = e[2]; kl.ePointInOctaveY = e[3];
        double ddx = (double)(e[0] - e[2]), ddy = (double)(e[1] - e[3]);
        kl.lineLength = (float)std::sqrt(ddx * ddx + ddy * ddy);
        // LineIterator(img, Point(cvRound), Point(cvRound)).count, 8-connected, endpoints already inside the image
        int ax = cv_roundf(e[0]), ay = cv_roundf(e[1]), bx = cv_roundf(e[2]), by = cv_roundf(e[3]);
        kl.numOfPixels = std::max(std::abs(bx - ax), std::abs(by - ay)) + 1;
        kl.angle = cr_atan2f(kl.endPointY - kl.startPointY, kl.endPointX - kl.startPointX);      // D5
        kl.class_id = ++class_counter;
        kl.octave = 0;
        kl.size = (kl.endPointX - kl.startPointX) * (kl.endPointY - kl.startPointY);
        kl.response = kl.lineLength / (float)std::max(image.w, image.h);
        kl.pt_x = (kl.endPointX + kl.startPointX) / 2; kl.pt_y = (kl.endPointY + kl.startPointY) / 2;
        keylines.push_back(kl);
    }
}

// test taps: the NFA of (n, k, p) for an image of w x h pixels, and cv::Sobel 3x3 as the LBD stage calls it
extern "C" int orc_set_lsd_resize(int v) { const int old = orc::g_lsdResize; orc::g_lsdResize = v == 1 ? 1 : 0; return old; }
extern "C" int orc_set_lsd_nfa_variant(int v) { const int old = orc::g_lsdNfaVariant; orc::g_lsdNfaVariant = v == 1 ? 1 : 0; return old; }
extern "C" int orc_set_lsd_seed_sort(int v) { const int old = orc::g_lsdSeedSort; orc::g_lsdSeedSort = v == 1 ? 1 : 0; return old; }
extern "C" double orc_lsd_nfa(int w, int h, int n, int k, double p) {
    Lsd lsd; lsd.w = w; lsd.h = h;
    lsd.LOG_NT = 5 * (std::log10(double(w)) + std::log10(double(h))) / 2 + std::log10(11.0);
    return lsd.nfa(n, k, p);
}
extern "C" double orc_log_gamma(double x) { return log_gamma(x); }
extern "C" int orc_sobel3(const uint8_t* src, int w, int h, int16_t* gx, int16_t* gy) {
    Img8 im(w, h); std::memcpy(im.d.data(), src, (size_t)w * h);
    std::vector<int16_t> x, y; sobel3_s16(im, x, y);
    std::memcpy(gx, x.data(), 2 * (size_t)w * h); std::memcpy(gy, y.data(), 2 * (size_t)w * h);
    return 0;
}

// stage taps used by the parity tests
void lsd_debug_scaled(const Img8& image, Img8& scaled_out) {
    Lsd lsd;
    const double sigma = lsd.SIGMA_SCALE / lsd.SCALE;
    const unsigned hk = (unsigned)(std::ceil(sigma * std::sqrt(2 * 3.0 * std::log(10.0))));
    scaled_out = resize_linear_exact_8u(gaussian_blur_8u(image, 1 + 2 * hk, sigma, g_gaussVariant), lsd.SCALE, lsd.SCALE);
}

}  // namespace orc
