// TEST / DESIGN-STUDY INFRASTRUCTURE (not product): walks the CPU oracle's LSD main loop (oracle/lsd_oracle.cpp, itself a restatement of
// OpenCV's LineSegmentDetectorImpl::flsd) with counters, to size the work of the sequential core before a GPU kernel is shaped around it:
// seeds grown, region sizes, accepts per staged queue point for staging widths 1/2/4/8, candidate rectangles, refine / radius reductions.
// Built by tests/sim/lsd_pack_sim.py with g++; never linked into the library.
#include "../../oracle/lsd_oracle.cpp"
namespace orc { int g_gaussVariant = 0; }      // (defined in orb_oracle.cpp, which this single-file build does not link)

namespace {
struct Trace {
    long long nDefined = 0, seeds = 0, accepted = 0, regionsGE = 0, refines = 0, reduceIters = 0, rectPoints = 0, rectCalls = 0, regrowAccepted = 0;
    long long stagings[4] = {0, 0, 0, 0}, ticks[4] = {0, 0, 0, 0};      // staging width 1, 2, 4, 8 queue points
    long long sizeHist[12] = {0};                                        // region size buckets: 1, 2, 3-4, 5-8, ... (log2)
    long long accHist[9] = {0};                                          // accepts per queue point 0..8
};

// region_grow with per-queue-point accept counts
static void grow_traced(orc::Lsd& L, int sx, int sy, std::vector<orc::RegionPoint>& reg, double& reg_angle, double prec, Trace& T, std::vector<int>& accPerPoint) {
    using namespace orc;
    reg.clear(); accPerPoint.clear();
    RegionPoint seed; seed.x = sx; seed.y = sy;
    reg_angle = L.angles[(size_t)sy * L.w + sx];
    seed.angle = reg_angle; seed.modgrad = L.modgrad[(size_t)sy * L.w + sx];
    reg.push_back(seed);
    float sumdx = float(std::cos(reg_angle)), sumdy = float(std::sin(reg_angle));
    L.used[(size_t)sy * L.w + sx] = 1;
    for (size_t i = 0; i < reg.size(); ++i) {
        const int px = reg[i].x, py = reg[i].y;
        int xx_min = std::max(px - 1, 0), xx_max = std::min(px + 1, L.w - 1);
        int yy_min = std::max(py - 1, 0), yy_max = std::min(py + 1, L.h - 1);
        int acc = 0;
        for (int yy = yy_min; yy <= yy_max; ++yy)
            for (int xx = xx_min; xx <= xx_max; ++xx) {
                uint8_t& is_used = L.used[(size_t)yy * L.w + xx];
                if (is_used != 1 && L.isAligned(xx, yy, reg_angle, prec)) {
                    const double angle = L.angles[(size_t)yy * L.w + xx];
                    is_used = 1;
                    RegionPoint rp; rp.x = xx; rp.y = yy; rp.modgrad = L.modgrad[(size_t)yy * L.w + xx]; rp.angle = angle;
                    reg.push_back(rp);
                    sumdx += cr_cosf(float(angle)); sumdy += cr_sinf(float(angle));
                    reg_angle = fast_atan2(sumdy, sumdx) * DEG_TO_RADS;
                    ++acc;
                }
            }
        accPerPoint.push_back(acc);
        T.accHist[std::min(acc, 8)]++;
    }
    // staging statistics: a staging of width q covers q consecutive queue points (the queue may grow while it is processed)
    for (int k = 0; k < 4; ++k) {
        const int q = 1 << k;
        for (size_t i = 0; i < accPerPoint.size(); i += q) {
            int a = 0;
            for (size_t j = i; j < std::min(accPerPoint.size(), i + q); ++j) a += accPerPoint[j];
            T.stagings[k]++; T.ticks[k] += std::max(1, a);
        }
    }
    T.accepted += (long long)reg.size() - 1;
}
}  // namespace

extern "C" int lsd_trace_frame(const uint8_t* gray, int w, int h, long long* out, int nout) {
    using namespace orc;
    Img8 image(w, h); std::memcpy(image.d.data(), gray, (size_t)w * h);
    Lsd L;
    const double prec = M_PI * L.ANG_TH / 180, p = L.ANG_TH / 180;
    const double rho = L.QUANT / std::sin(prec);
    const double sigma = L.SIGMA_SCALE / L.SCALE;
    const unsigned hk = (unsigned)(std::ceil(sigma * std::sqrt(2 * 3.0 * std::log(10.0))));
    Img8 g = gaussian_blur_8u(image, 1 + 2 * hk, sigma);
    L.scaled = resize_linear_exact_8u(g, L.SCALE, L.SCALE);
    L.w = L.scaled.w; L.h = L.scaled.h;
    L.ll_angle(rho);
    L.LOG_NT = 5 * (std::log10(double(L.w)) + std::log10(double(L.h))) / 2 + std::log10(11.0);
    const size_t min_reg_size = size_t(-L.LOG_NT / std::log10(p));
    L.used.assign((size_t)L.w * L.h, 0);
    Trace T;
    std::vector<RegionPoint> reg; std::vector<int> app;
    for (size_t i = 0; i < L.order.size(); ++i) {
        const int idx = L.order[i], px = idx % L.w, py = idx / L.w;
        if (L.angles[idx] != NOTDEF) T.nDefined++;
        if (L.used[idx] == 0 && L.angles[idx] != NOTDEF) {
            double reg_angle;
            T.seeds++;
            grow_traced(L, px, py, reg, reg_angle, prec, T, app);
            int b = 0; for (size_t s = reg.size(); s > 1; s = (s + 1) >> 1) ++b;
            T.sizeHist[std::min(b, 11)]++;
            if (reg.size() < min_reg_size) continue;
            T.regionsGE++;
            Rect rec;
            L.region2rect(reg, reg_angle, prec, p, rec); T.rectCalls++; T.rectPoints += reg.size();
            double density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
            if (density < L.DENSITY_TH) {
                T.refines++;
                // refine(): un-mark, statistics, re-grow with tau (restated from Lsd::refine so that the re-growth can be traced)
                double xc = double(reg[0].x), yc = double(reg[0].y);
                const double ang_c = reg[0].angle;
                double sum = 0, s_sum = 0; int n = 0;
                for (size_t k = 0; k < reg.size(); ++k) {
                    L.used[(size_t)reg[k].y * L.w + reg[k].x] = 0;
                    if (dist(xc, yc, reg[k].x, reg[k].y) < rec.width) { double ang_d = angle_diff_signed(reg[k].angle, ang_c); sum += ang_d; s_sum += ang_d * ang_d; ++n; }
                }
                double mean_angle = sum / double(n);
                double tau = 2.0 * std::sqrt((s_sum - 2.0 * mean_angle * sum) / double(n) + mean_angle * mean_angle);
                const long long before = T.accepted;
                grow_traced(L, reg[0].x, reg[0].y, reg, reg_angle, tau, T, app);
                T.regrowAccepted += T.accepted - before;
                if (reg.size() < 2) continue;
                L.region2rect(reg, reg_angle, prec, p, rec); T.rectCalls++; T.rectPoints += reg.size();
                density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
                if (density < L.DENSITY_TH) {
                    // reduce_region_radius with iteration counting
                    double radSq1 = distSq(xc, yc, rec.x1, rec.y1), radSq2 = distSq(xc, yc, rec.x2, rec.y2);
                    double radSq = radSq1 > radSq2 ? radSq1 : radSq2;
                    bool good = true;
                    while (density < L.DENSITY_TH) {
                        T.reduceIters++;
                        radSq *= 0.75 * 0.75;
                        for (size_t k = 0; k < reg.size(); ++k)
                            if (distSq(xc, yc, double(reg[k].x), double(reg[k].y)) > radSq) {
                                L.used[(size_t)reg[k].y * L.w + reg[k].x] = 0;
                                std::swap(reg[k], reg[reg.size() - 1]); reg.pop_back(); --k;
                            }
                        if (reg.size() < 2) { good = false; break; }
                        L.region2rect(reg, reg_angle, prec, p, rec); T.rectCalls++; T.rectPoints += reg.size();
                        density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
                    }
                    if (!good) continue;
                }
            }
        }
    }
    long long v[64] = {0}; int k = 0;
    v[k++] = T.nDefined; v[k++] = T.seeds; v[k++] = T.accepted; v[k++] = T.regionsGE; v[k++] = T.refines; v[k++] = T.reduceIters;
    v[k++] = T.rectCalls; v[k++] = T.rectPoints; v[k++] = T.regrowAccepted;
    for (int i = 0; i < 4; ++i) v[k++] = T.stagings[i];
    for (int i = 0; i < 4; ++i) v[k++] = T.ticks[i];
    for (int i = 0; i < 12; ++i) v[k++] = T.sizeHist[i];
    for (int i = 0; i < 9; ++i) v[k++] = T.accHist[i];
    for (int i = 0; i < std::min(k, nout); ++i) out[i] = v[i];
    return k;
}
