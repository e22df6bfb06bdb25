// TEST INFRASTRUCTURE (not product): the protocol of the multi-wave LSD core (structure-slam-pointline_amd/csrc/lsd_regions.h)
// with real threads and real races on the CPU, on top of the oracle's LSD (oracle/lsd_oracle.cpp).  One MAIN thread replays flsd()'s seed
// loop in order and is the only writer of the `used` map; H HELPER threads claim chunks of 64 order positions and run the per-seed body
// (region_grow, region2rect, refine with re-growth and reduce_region_radius) for the unused seeds of their chunk, each seed on its own, on a
// read-only view of the map plus private marks.  The main thread TAKES a helper's result when
//   (b) every point the helper accepted (lists A and B) is unused now, and
//   (c) no refine() of the main thread released pixels near the result (bounding boxes, sequence numbers) since the helper sampled the counter,
// and runs the body itself otherwise.  Compared with the sequential run: every seed that ran (order position, region size, emitted or not,
// the rectangle's twelve doubles) and the final `used` map -- bit for bit.  `chaos` makes threads yield at random pixel reads so that views go
// stale at every possible point; with check (b) or (c) switched off the comparison must (and does) fail, which shows the test can see a broken
// protocol.  Built and run by tests/test_mw_proto_cpu.py.
//
// CLUSTER protocol (csrc/lsd_cluster.h; mode 1): the helpers sit on other compute units, so their view of the map can be
// ARBITRARILY STALE -- modelled here by a per-helper cache of 64-pixel lines that is refreshed from the map only once in `stale` accesses
// (lines of different age side by side, never going back in time).  The main thread therefore never releases a pixel in the map: a seed it
// runs itself is grown on private marks (the very body the helpers run) and only what ends up USED is committed; a taken result commits
// its last list.  With the map monotonic, check (b) alone is the validation -- and the negative controls show both halves: without (b) the
// runs differ, and the multi-wave protocol (releases in the map, checks (b) + (c)) on stale views differs as well, which is why the
// cluster form needs the monotonic map.
//   usage: mw_proto <frame.raw> <w> <h> <helpers> <repeats> <chaos: yield once in N reads, 0 = never> [checks: 3 = both (default), 1 = only (b), 2 = only (c), 0 = none] [percentage of seeds the main thread does itself regardless] [mode: 0 multi-wave, 1 cluster] [stale: a cached line is refreshed once in N reads, 0 = views are always fresh]
#include "../../oracle/lsd_oracle.cpp"
namespace orc { int g_gaussVariant = 0; }      // (defined in orb_oracle.cpp, which this single-file build does not link)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
using namespace orc;

struct SeedLog { int pos, size, emit; Rect rec; };
static bool same_log(const std::vector<SeedLog>& a, const std::vector<SeedLog>& b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i) {
        if (a[i].pos != b[i].pos || a[i].size != b[i].size || a[i].emit != b[i].emit) return false;
        if (a[i].emit && std::memcmp(&a[i].rec, &b[i].rec, sizeof(Rect)) != 0) return false;
    }
    return true;
}

struct Result { int lane = -1, startSeq = 0, emit = 0; std::vector<int> A, B, F; Rect rec; int x0, y0, x1, y1; };
struct Chunk {
    std::atomic<int> pos{-1}, done{0};      // done: lanes below it are settled
    std::mutex mu; std::vector<Result> res;
};

struct Mw : Lsd {
    double prec = 0, p = 0; size_t min_reg_size = 0;
    int mode = 0, stale = 0;                // mode 1: cluster protocol (monotonic map, private growth of the main thread, check (b) only); stale: see above
    int H = 4, chaos = 0, checks = 3, ownPct = 0;      // ownPct: the main thread ignores the helper's result for this share of the seeds (more refines of its own -> more releases)
    std::vector<Chunk> chunks;              // one per chunk of the order list (the kernel recycles a few slots; storage is not what is tested)
    std::atomic<int> cursor{0}, mainPos{0}, finished{0}, unmarkSeq{0};
    struct Ev { int x0, y0, x1, y1; }; std::vector<Ev> events; std::mutex evMu;
    long nTaken = 0, nOwn = 0, nRefused = 0; std::atomic<long> nRepublished{0}, nLate{0};

    void prepare(const Img8& image) {       // Lsd::detect up to the seed loop
        prec = M_PI * ANG_TH / 180; p = ANG_TH / 180;
        const double rho = QUANT / std::sin(prec), sigma = SIGMA_SCALE / SCALE, sprec = 3;
        const unsigned hk = (unsigned)(std::ceil(sigma * std::sqrt(2 * sprec * std::log(10.0))));
        Img8 g = gaussian_blur_8u(image, 1 + 2 * hk, sigma);
        scaled = resize_linear_exact_8u(g, SCALE, SCALE);
        w = scaled.w; h = scaled.h;
        ll_angle(rho);
        LOG_NT = 5 * (std::log10(double(w)) + std::log10(double(h))) / 2 + std::log10(11.0);
        min_reg_size = size_t(-LOG_NT / std::log10(p));
    }
    // ---------------------------------------------------------------- sequential reference with the same log
    void run_sequential(std::vector<SeedLog>& log) {
        used.assign((size_t)w * h, 0);
        std::vector<RegionPoint> reg;
        for (size_t i = 0; i < order.size(); ++i) {
            const int idx = order[i];
            if (used[idx] == 0 && angles[idx] != NOTDEF) {
                double reg_angle; SeedLog s; s.pos = (int)i; s.emit = 0;
                region_grow(idx % w, idx / w, reg, reg_angle, prec);
                s.size = (int)reg.size();
                if (reg.size() >= min_reg_size) {
                    region2rect(reg, reg_angle, prec, p, s.rec);
                    s.emit = refine(reg, reg_angle, prec, p, s.rec, DENSITY_TH) ? 1 : 0;
                    s.size = (int)reg.size();
                }
                log.push_back(s);
            }
        }
    }
    // ---------------------------------------------------------------- helper side: the body on a read-only view + private marks
    struct HCtx { std::vector<uint8_t> mine; unsigned long long rng; int stale = 0; std::vector<uint8_t> cache, have; };
    void jitter(unsigned long long& rng) { if (chaos) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; if (rng % (unsigned)chaos == 0) std::this_thread::yield(); } }
    bool view_unused(HCtx& c, int idx) {
        jitter(c.rng);
        if (c.stale) {                          // a cache of 64-pixel lines: a line is fetched on first use and refreshed only now and then
            const size_t line = (size_t)idx / 64, n = used.size();
            c.rng ^= c.rng << 13; c.rng ^= c.rng >> 7; c.rng ^= c.rng << 17;
            if (!c.have[line] || c.rng % (unsigned)c.stale == 0) {
                for (size_t i = line * 64; i < std::min(n, line * 64 + 64); ++i) c.cache[i] = reinterpret_cast<volatile uint8_t*>(used.data())[i];
                c.have[line] = 1;
            }
            return c.cache[idx] == 0 && !c.mine[idx];
        }
        return reinterpret_cast<volatile uint8_t*>(used.data())[idx] == 0 && !c.mine[idx];
    }
    void grow_private(HCtx& c, int sx, int sy, std::vector<RegionPoint>& reg, double& reg_angle, double prc) {
        reg.clear();
        RegionPoint seed; seed.x = sx; seed.y = sy;
        reg_angle = angles[(size_t)sy * w + sx];
        seed.angle = reg_angle; seed.modgrad = modgrad[(size_t)sy * w + sx];
        reg.push_back(seed);
        float sumdx = float(std::cos(reg_angle)), sumdy = float(std::sin(reg_angle));
        c.mine[(size_t)sy * w + sx] = 1;
        for (size_t i = 0; i < reg.size(); ++i) {
            const int px = reg[i].x, py = reg[i].y;
            int xx_min = std::max(px - 1, 0), xx_max = std::min(px + 1, w - 1), yy_min = std::max(py - 1, 0), yy_max = std::min(py + 1, h - 1);
            for (int yy = yy_min; yy <= yy_max; ++yy)
                for (int xx = xx_min; xx <= xx_max; ++xx) {
                    const int id = yy * w + xx;
                    if (view_unused(c, id) && isAligned(xx, yy, reg_angle, prc)) {
                        const double angle = angles[id];
                        c.mine[id] = 1;
                        RegionPoint rp; rp.x = xx; rp.y = yy; rp.modgrad = modgrad[id]; rp.angle = angle;
                        reg.push_back(rp);
                        sumdx += cr_cosf(float(angle)); sumdy += cr_sinf(float(angle));
                        reg_angle = fast_atan2(sumdy, sumdx) * DEG_TO_RADS;
                    }
                }
        }
    }
    static void to_idx(const std::vector<RegionPoint>& reg, int w, std::vector<int>& out) { out.clear(); for (auto& r : reg) out.push_back(r.y * w + r.x); }
    void helper_body(HCtx& c, int sx, int sy, Result& R) {
        R.startSeq = unmarkSeq.load(std::memory_order_acquire);          // BEFORE the first pixel of this region is read
        std::vector<RegionPoint> reg; double reg_angle;
        grow_private(c, sx, sy, reg, reg_angle, prec);
        to_idx(reg, w, R.A); R.B.clear(); R.emit = 0;
        if (reg.size() >= min_reg_size) {
            region2rect(reg, reg_angle, prec, p, R.rec);
            // Lsd::refine on the private marks
            double density = double(reg.size()) / (dist(R.rec.x1, R.rec.y1, R.rec.x2, R.rec.y2) * R.rec.width);
            bool ok = true;
            if (density < DENSITY_TH) {
                double xc = double(reg[0].x), yc = double(reg[0].y); const double ang_c = reg[0].angle;
                double sum = 0, s_sum = 0; int n = 0;
                for (size_t i = 0; i < reg.size(); ++i) {
                    c.mine[(size_t)reg[i].y * w + reg[i].x] = 0;
                    if (dist(xc, yc, reg[i].x, reg[i].y) < R.rec.width) { double ang_d = angle_diff_signed(reg[i].angle, ang_c); sum += ang_d; s_sum += ang_d * ang_d; ++n; }
                }
                double mean_angle = sum / double(n);
                double tau = 2.0 * std::sqrt((s_sum - 2.0 * mean_angle * sum) / double(n) + mean_angle * mean_angle);
                grow_private(c, reg[0].x, reg[0].y, reg, reg_angle, tau);
                to_idx(reg, w, R.B);
                if (reg.size() < 2) ok = false;
                else {
                    region2rect(reg, reg_angle, prec, p, R.rec);
                    density = double(reg.size()) / (dist(R.rec.x1, R.rec.y1, R.rec.x2, R.rec.y2) * R.rec.width);
                    if (density < DENSITY_TH) {                          // Lsd::reduce_region_radius on the private marks
                        double radSq1 = distSq(xc, yc, R.rec.x1, R.rec.y1), radSq2 = distSq(xc, yc, R.rec.x2, R.rec.y2);
                        double radSq = radSq1 > radSq2 ? radSq1 : radSq2;
                        while (density < DENSITY_TH) {
                            radSq *= 0.75 * 0.75;
                            for (size_t i = 0; i < reg.size(); ++i)
                                if (distSq(xc, yc, double(reg[i].x), double(reg[i].y)) > radSq) {
                                    c.mine[(size_t)reg[i].y * w + reg[i].x] = 0;
                                    std::swap(reg[i], reg[reg.size() - 1]); reg.pop_back(); --i;
                                }
                            if (reg.size() < 2) { ok = false; break; }
                            region2rect(reg, reg_angle, prec, p, R.rec);
                            density = double(reg.size()) / (dist(R.rec.x1, R.rec.y1, R.rec.x2, R.rec.y2) * R.rec.width);
                        }
                    }
                }
            }
            R.emit = ok ? 1 : 0;
        }
        to_idx(reg, w, R.F);
        for (int id : R.F) c.mine[id] = 0;                                // the next region is grown on its own
        R.x0 = R.y0 = 1 << 30; R.x1 = R.y1 = -1;
        for (const std::vector<int>* l : {&R.A, &R.B}) for (int id : *l) { R.x0 = std::min(R.x0, id % w); R.x1 = std::max(R.x1, id % w); R.y0 = std::min(R.y0, id / w); R.y1 = std::max(R.y1, id / w); }
    }
    void helper_thread(int hid) {
        HCtx c; c.mine.assign((size_t)w * h, 0); c.rng = 88172645463325252ull + 7919ull * hid;
        c.stale = stale; if (stale) { c.cache.assign((size_t)w * h, 0); c.have.assign(((size_t)w * h + 63) / 64, 0); }
        const int nOrd = (int)order.size();
        for (;;) {
            if (finished.load()) return;
            int cpos = cursor.load();
            if (cpos >= nOrd) return;
            const int mp = mainPos.load();
            if (cpos < mp) { cursor.compare_exchange_strong(cpos, mp); continue; }
            if (!cursor.compare_exchange_strong(cpos, cpos + 64)) continue;
            Chunk& K = chunks[cpos / 64];
            K.pos.store(cpos, std::memory_order_release);
            for (int lane = 0; lane < 64 && cpos + lane < nOrd; ++lane) {
                if (mainPos.load() > cpos || finished.load()) break;
                const int idx = order[cpos + lane];
                if (angles[idx] == NOTDEF || !view_unused(c, idx)) { K.done.store(lane + 1, std::memory_order_release); continue; }
                Result R; R.lane = lane;
                helper_body(c, idx % w, idx / w, R);
                { std::lock_guard<std::mutex> g(K.mu); K.res.push_back(std::move(R)); }
                K.done.store(lane + 1, std::memory_order_release);
            }
            K.done.store(64, std::memory_order_release);
            // cluster protocol: while the main thread is still in front of the chunk the helper checks what it published the way the main
            // thread will (every accepted point unused -- in ITS view) and publishes a seed whose result died again, as a NEW result (results
            // are never rewritten; the later one of a seed counts)
            for (int round = 0; mode == 1 && round < 3 && mainPos.load() < cpos && !finished.load(); ++round) {
                size_t nres; { std::lock_guard<std::mutex> g(K.mu); nres = K.res.size(); }
                for (size_t ri = 0; ri < nres && mainPos.load() < cpos; ++ri) {
                    Result old; { std::lock_guard<std::mutex> g(K.mu); old = K.res[ri]; }
                    bool superseded = false;
                    { std::lock_guard<std::mutex> g(K.mu); for (size_t rj = ri + 1; rj < K.res.size(); ++rj) superseded = superseded || K.res[rj].lane == old.lane; }
                    if (superseded) continue;
                    bool dead = false;
                    for (const std::vector<int>* l : {&old.A, &old.B}) for (int id : *l) dead = dead || !view_unused(c, id);
                    const int idx = order[cpos + old.lane];
                    if (!dead || !view_unused(c, idx)) continue;
                    Result R; R.lane = old.lane;
                    helper_body(c, idx % w, idx / w, R);
                    { std::lock_guard<std::mutex> g(K.mu); K.res.push_back(std::move(R)); }
                    ++nRepublished;
                }
                std::this_thread::yield();
            }
        }
    }
    // ---------------------------------------------------------------- main side
    void run_multi(std::vector<SeedLog>& log) {
        used.assign((size_t)w * h, 0);
        const int nOrd = (int)order.size();
        chunks = std::vector<Chunk>((nOrd + 63) / 64 + 1);
        cursor = 0; mainPos = 0; finished = 0; unmarkSeq = 0; events.clear(); nTaken = nOwn = nRefused = 0; nRepublished = 0; nLate = 0;
        std::vector<std::thread> th;
        for (int i = 0; i < H; ++i) th.emplace_back([this, i] { helper_thread(i); });
        unsigned long long rng = 1234567;
        std::vector<RegionPoint> reg;
        HCtx own; own.mine.assign((size_t)w * h, 0); own.rng = 99;          // cluster protocol: the main thread's private marks (its view is the map itself)
        for (int pos0 = 0; pos0 < nOrd; pos0 += 64) {
            mainPos.store(pos0, std::memory_order_release);
            // whose chunk: below the cursor a helper claimed it, otherwise the main thread takes it
            bool mine = false;
            for (;;) { int c = cursor.load(); if (c > pos0) break; if (cursor.compare_exchange_strong(c, pos0 + 64)) { mine = true; break; } }
            Chunk& K = chunks[pos0 / 64];
            for (int lane = 0; lane < 64 && pos0 + lane < nOrd; ++lane) {
                const int idx = order[pos0 + lane];
                if (!(used[idx] == 0 && angles[idx] != NOTDEF)) continue;
                jitter(rng);
                SeedLog s; s.pos = pos0 + lane; s.emit = 0;
                bool took = false;
                rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
                if (!mine && (int)(rng % 100) >= ownPct) {
                    while (K.pos.load(std::memory_order_acquire) != pos0) std::this_thread::yield();
                    while (K.done.load(std::memory_order_acquire) <= lane) std::this_thread::yield();
                    Result Rcopy; Result* R = nullptr;      // (a copy: the helper may append to the list, and move it, while the main thread looks at this one)
                    int tried = -1;
                    { std::lock_guard<std::mutex> g(K.mu); for (size_t ri = 0; ri < K.res.size(); ++ri) if (K.res[ri].lane == lane) { Rcopy = K.res[ri]; R = &Rcopy; tried = (int)ri; } }
                    for (int attempt = 0; attempt < 2 && R && !took; ++attempt) {
                        if (attempt == 1) {         // cluster protocol: after a refusal, one more look for a result the helper published since
                            R = nullptr;
                            if (mode != 1) break;
                            std::lock_guard<std::mutex> g(K.mu);
                            for (size_t ri = tried + 1; ri < K.res.size(); ++ri) if (K.res[ri].lane == lane) { Rcopy = K.res[ri]; R = &Rcopy; }
                            if (!R) break;
                            ++nLate;
                        }
                        bool ok = true;
                        if ((checks & 2) && mode == 0) {                    // (c) no release near the result since its sample (the cluster protocol has no releases)
                            std::lock_guard<std::mutex> g(evMu);
                            for (int sq = R->startSeq; ok && sq < (int)events.size(); ++sq) {
                                const Ev& e = events[sq];
                                ok = R->x1 + 1 < e.x0 || e.x1 < R->x0 - 1 || R->y1 + 1 < e.y0 || e.y1 < R->y0 - 1;
                            }
                        }
                        if (checks & 1) for (const std::vector<int>* l : {&R->A, &R->B}) for (int id : *l) ok = ok && used[id] == 0;      // (b)
                        if (ok) {
                            for (int id : R->F) used[id] = 1;
                            s.size = (int)R->F.size(); s.emit = R->emit; s.rec = R->rec; took = true; ++nTaken;
                        } else ++nRefused;
                    }
                }
                if (!took && mode == 1) {                                   // cluster protocol: the body on private marks, then ONE commit -- the map never loses a pixel
                    Result R; ++nOwn;
                    helper_body(own, idx % w, idx / w, R);
                    for (int id : R.F) used[id] = 1;
                    s.size = (int)R.F.size(); s.emit = R.emit; s.rec = R.rec;
                } else if (!took) {                                         // the body by the main thread itself, on the map
                    double reg_angle; ++nOwn;
                    region_grow(idx % w, idx / w, reg, reg_angle, prec);
                    s.size = (int)reg.size();
                    if (reg.size() >= min_reg_size) {
                        region2rect(reg, reg_angle, prec, p, s.rec);
                        // Lsd::refine with the box of everything it touches (the first region and the re-grown one) logged AFTER its last store
                        Ev e{1 << 30, 1 << 30, -1, -1};
                        auto grow_box = [&](const std::vector<RegionPoint>& r) { for (auto& q : r) { e.x0 = std::min(e.x0, q.x); e.x1 = std::max(e.x1, q.x); e.y0 = std::min(e.y0, q.y); e.y1 = std::max(e.y1, q.y); } };
                        double density = double(reg.size()) / (dist(s.rec.x1, s.rec.y1, s.rec.x2, s.rec.y2) * s.rec.width);
                        bool ok = true;
                        if (density < DENSITY_TH) {
                            grow_box(reg);
                            // (stale views: an adversarial schedule -- the region stays marked for a while before refine() releases it, long enough
                            // for helpers to cache lines that show it USED)
                            if (stale) std::this_thread::sleep_for(std::chrono::microseconds(400));
                            double xc = double(reg[0].x), yc = double(reg[0].y); const double ang_c = reg[0].angle;
                            double sum = 0, s_sum = 0; int n = 0;
                            for (size_t i = 0; i < reg.size(); ++i) {
                                used[(size_t)reg[i].y * w + reg[i].x] = 0;
                                if (dist(xc, yc, reg[i].x, reg[i].y) < s.rec.width) { double ang_d = angle_diff_signed(reg[i].angle, ang_c); sum += ang_d; s_sum += ang_d * ang_d; ++n; }
                            }
                            double mean_angle = sum / double(n);
                            double tau = 2.0 * std::sqrt((s_sum - 2.0 * mean_angle * sum) / double(n) + mean_angle * mean_angle);
                            region_grow(reg[0].x, reg[0].y, reg, reg_angle, tau);
                            grow_box(reg);
                            if (reg.size() < 2) ok = false;
                            else {
                                region2rect(reg, reg_angle, prec, p, s.rec);
                                density = double(reg.size()) / (dist(s.rec.x1, s.rec.y1, s.rec.x2, s.rec.y2) * s.rec.width);
                                if (density < DENSITY_TH) ok = reduce_region_radius(reg, reg_angle, prec, p, s.rec, density, DENSITY_TH);
                            }
                            std::lock_guard<std::mutex> g(evMu);
                            events.push_back(e);
                            unmarkSeq.store((int)events.size(), std::memory_order_release);
                        }
                        s.emit = ok ? 1 : 0;
                        s.size = (int)reg.size();
                    }
                }
                log.push_back(s);
            }
        }
        finished.store(1);
        for (auto& t : th) t.join();
    }
};

int main(int argc, char** argv) {
    if (argc < 7) { std::fprintf(stderr, "usage: mw_proto frame.raw w h helpers repeats chaos [checks]\n"); return 2; }
    const int W = std::atoi(argv[2]), Hh = std::atoi(argv[3]);
    Img8 img(W, Hh);
    FILE* f = std::fopen(argv[1], "rb"); if (!f || std::fread(img.d.data(), 1, img.d.size(), f) != img.d.size()) { std::fprintf(stderr, "cannot read frame\n"); return 2; }
    std::fclose(f);
    Mw M; M.H = std::atoi(argv[4]); const int reps = std::atoi(argv[5]); M.chaos = std::atoi(argv[6]); M.checks = argc > 7 ? std::atoi(argv[7]) : 3; M.ownPct = argc > 8 ? std::atoi(argv[8]) : 0;
    M.mode = argc > 9 ? std::atoi(argv[9]) : 0; M.stale = argc > 10 ? std::atoi(argv[10]) : 0;
    M.prepare(img);
    std::vector<SeedLog> ref; M.run_sequential(ref);
    const std::vector<uint8_t> usedRef = M.used;
    int bad = 0;
    for (int r = 0; r < reps; ++r) {
        std::vector<SeedLog> log; M.run_multi(log);
        const bool same = same_log(ref, log) && usedRef == M.used;
        bad += !same;
        std::printf("run %d: %s  seeds %zu  taken %ld  own %ld  refused %ld  refine events %zu  published again %ld  taken from a later result %ld\n", r, same ? "identical" : "DIFFERENT", log.size(), M.nTaken, M.nOwn, M.nRefused, M.events.size(), M.nRepublished.load(), M.nLate.load());
    }
    std::printf("different runs: %d of %d\n", bad, reps);
    return bad ? 1 : 0;
}
