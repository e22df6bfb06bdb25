// CPU prototype (real threads, real races) of the multi-wave speculation planned for the single-frame LSD core.
// W worker threads stand for W waves of one workgroup: each takes the next seed in order, grows / refines its region speculatively on a
// shared claim map, and commits in seed order; a region whose read tiles were written by a region that committed after it started is
// redone at its turn.  The output (rectangles -> rect_improve -> segments) must equal the oracle's sequential flsd() bit for bit.
//   pixel state (atomic int): NOTDEF = INT_MIN, unused = 0, claimed by the seed at order position s = s - 2^30 (older seed = smaller value)
//   claim = atomic min, un-claim = CAS(own tag -> 0); a reader treats a tag above its own (a younger seed's claim) as unused.
// build: g++ -O2 -std=c++17 -ffp-contract=off -pthread -Ioracle tools/lsd_spec_proto.cpp -o /tmp/lsd_spec_proto
// run:   /tmp/lsd_spec_proto <frame.raw 640x480> <threads> <repeats> [chaos: yield once in N pixel reads] [0 = no stamp validation] [1 = FIFO hand-out as in the draft kernel]
#include "../oracle/lsd_oracle.cpp"
namespace orc { int g_gaussVariant = 0; }      // (defined in orb_oracle.cpp, which this single-file build does not link)
#include <atomic>
#include <climits>
#include <cstdio>
#include <mutex>
#include <thread>
using namespace orc;

static const int NOTDEF_T = INT_MIN, TAG0 = -(1 << 30), TS = 8;
static inline bool is_tag(int x) { return x < 0 && x != NOTDEF_T; }

struct Spec : Lsd {
    std::vector<std::atomic<int>> tag;
    std::vector<std::atomic<int>> tileStamp;
    int tw = 0, ntiles = 0;
    std::atomic<int> commitSeq{0}, commitTicket{0};
    std::mutex pickMu; size_t pickPos = 0; int tickets = 0;
    static const int RING = 256; int ringPos[RING];
    std::vector<Rect> emitted;
    struct SeedLog { int pos, emit, size; };
    std::vector<SeedLog> seedLog;            // every seed that ran, in commit order: the strong comparison (segments alone hide most conflicts)
    size_t min_reg_size = 0; double prec = 0, p = 0;
    long nSpecOk = 0, nRedo = 0, nWait = 0;          // written under the commit turn

    struct Ctx { int T = 0; std::vector<RegionPoint> reg; std::vector<char> rd, wr; std::vector<int> rdl, wrl; unsigned long long rng = 88172645463325252ull; };
    int chaos = 0;            // > 0: a thread yields at a random one in `chaos` pixel reads, so that regions really overlap in time
    bool validate = true;     // false: skip the stamp test (the comparison must then FAIL under chaos: shows that the test can see a broken protocol)
    void jitter(Ctx& c) { if (chaos) { c.rng ^= c.rng << 13; c.rng ^= c.rng >> 7; c.rng ^= c.rng << 17; if (c.rng % (unsigned)chaos == 0) std::this_thread::yield(); } }
    void touch_r(Ctx& c, int x, int y) { int t = (y / TS) * tw + x / TS; if (!c.rd[t]) { c.rd[t] = 1; c.rdl.push_back(t); } }
    void touch_w(Ctx& c, int x, int y) { int t = (y / TS) * tw + x / TS; if (!c.wr[t]) { c.wr[t] = 1; c.wrl.push_back(t); } touch_r(c, x, y); }
    bool unused_for(Ctx& c, int x, int y) {
        touch_r(c, x, y); jitter(c);
        const int v = tag[(size_t)y * w + x].load(std::memory_order_acquire);
        return v >= 0 || (is_tag(v) && v > c.T);
    }
    void claim(Ctx& c, int x, int y) {
        touch_w(c, x, y);
        std::atomic<int>& a = tag[(size_t)y * w + x];
        int cur = a.load(std::memory_order_acquire);
        while (cur > c.T && cur != NOTDEF_T) { if (a.compare_exchange_weak(cur, c.T, std::memory_order_acq_rel)) break; }
    }
    void unclaim(Ctx& c, int x, int y) {
        touch_w(c, x, y);
        int exp = c.T; tag[(size_t)y * w + x].compare_exchange_strong(exp, 0, std::memory_order_acq_rel);
    }
    // Lsd::region_grow with the claim map
    void grow(Ctx& c, int sx, int sy, double& reg_angle, double prc) {
        auto& reg = c.reg; reg.clear();
        RegionPoint seed; seed.x = sx; seed.y = sy;
        reg_angle = angles[(size_t)sy * w + sx];
        seed.angle = reg_angle; seed.modgrad = modgrad[(size_t)sy * w + sx];
        reg.push_back(seed);
        float sumdx = float(std::cos(reg_angle)), sumdy = float(std::sin(reg_angle));
        claim(c, sx, sy);
        for (size_t i = 0; i < reg.size(); ++i) {
            const int px = reg[i].x, py = reg[i].y;
            int xx_min = std::max(px - 1, 0), xx_max = std::min(px + 1, w - 1), yy_min = std::max(py - 1, 0), yy_max = std::min(py + 1, h - 1);
            for (int yy = yy_min; yy <= yy_max; ++yy)
                for (int xx = xx_min; xx <= xx_max; ++xx) {
                    if (unused_for(c, xx, yy) && isAligned(xx, yy, reg_angle, prc)) {
                        const double angle = angles[(size_t)yy * w + xx];
                        claim(c, xx, yy);
                        RegionPoint rp; rp.x = xx; rp.y = yy; rp.modgrad = modgrad[(size_t)yy * w + xx]; rp.angle = angle;
                        reg.push_back(rp);
                        sumdx += cr_cosf(float(angle)); sumdy += cr_sinf(float(angle));
                        reg_angle = fast_atan2(sumdy, sumdx) * DEG_TO_RADS;
                    }
                }
        }
    }
    bool reduce(Ctx& c, double reg_angle, Rect& rec, double density) {
        auto& reg = c.reg;
        double xc = double(reg[0].x), yc = double(reg[0].y);
        double radSq1 = distSq(xc, yc, rec.x1, rec.y1), radSq2 = distSq(xc, yc, rec.x2, rec.y2);
        double radSq = radSq1 > radSq2 ? radSq1 : radSq2;
        while (density < DENSITY_TH) {
            radSq *= 0.75 * 0.75;
            for (size_t i = 0; i < reg.size(); ++i) {
                if (distSq(xc, yc, double(reg[i].x), double(reg[i].y)) > radSq) {
                    unclaim(c, reg[i].x, reg[i].y);
                    std::swap(reg[i], reg[reg.size() - 1]); reg.pop_back(); --i;
                }
            }
            if (reg.size() < 2) return false;
            region2rect(reg, reg_angle, prec, p, rec);
            density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        }
        return true;
    }
    bool refine_t(Ctx& c, double& reg_angle, Rect& rec) {
        auto& reg = c.reg;
        double density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        if (density >= DENSITY_TH) return true;
        double xc = double(reg[0].x), yc = double(reg[0].y);
        const double ang_c = reg[0].angle;
        double sum = 0, s_sum = 0; int n = 0;
        for (size_t i = 0; i < reg.size(); ++i) {
            unclaim(c, reg[i].x, reg[i].y);
            if (dist(xc, yc, reg[i].x, reg[i].y) < rec.width) { double d = angle_diff_signed(reg[i].angle, ang_c); sum += d; s_sum += d * d; ++n; }
        }
        double mean_angle = sum / double(n);
        double tau = 2.0 * std::sqrt((s_sum - 2.0 * mean_angle * sum) / double(n) + mean_angle * mean_angle);
        int sx = reg[0].x, sy = reg[0].y;
        grow(c, sx, sy, reg_angle, tau);
        if (reg.size() < 2) return false;
        region2rect(reg, reg_angle, prec, p, rec);
        density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        if (density < DENSITY_TH) return reduce(c, reg_angle, rec, density);
        return true;
    }
    bool body(Ctx& c, int idx, Rect& rec) {                 // one seed of flsd()'s loop, up to the hand-over to rect_improve
        double reg_angle;
        grow(c, idx % w, idx / w, reg_angle, prec);
        if (c.reg.size() < min_reg_size) return false;
        region2rect(c.reg, reg_angle, prec, p, rec);
        return refine_t(c, reg_angle, rec);
    }
    // ---- hand-out exactly as in the draft HIP kernel (branch wip/lsd-spec): a FIFO of seeds refilled under a lock by whoever finds it empty,
    // 64 order entries per scan step, the commit counter sampled once per refill, the "claimed by a region in flight" flag taken at scan time
    bool fifoMode = false;
    static const int FIFO = 32, PUSH_CAP = 16;
    std::atomic<int> fifoHead{0}, fifoTail{0}, exhausted{0};
    int fifoPos[FIFO], fifoSeq[FIFO], fifoWait[FIFO];
    bool take_fifo(int& ticket, int& pos, int& startSeq, bool& wait) {
        while (true) {
            const int head = fifoHead.load(std::memory_order_acquire), tail = fifoTail.load(std::memory_order_acquire);
            if (head < tail) {
                const int sl = head % FIFO;
                pos = fifoPos[sl]; startSeq = fifoSeq[sl]; wait = fifoWait[sl] != 0;
                int expect = head;
                if (fifoHead.compare_exchange_strong(expect, head + 1, std::memory_order_acq_rel)) { ticket = head; return true; }
                continue;
            }
            if (exhausted.load(std::memory_order_acquire)) return false;
            if (!pickMu.try_lock()) { std::this_thread::yield(); continue; }
            const int seq0 = commitSeq.load(std::memory_order_acquire);
            const int c0 = commitTicket.load(std::memory_order_acquire);
            int tl = fifoTail.load(std::memory_order_acquire);
            const int oldestTag = c0 == tl ? INT_MAX : TAG0 + ringPos[c0 % RING];
            size_t p0 = pickPos; int pushed = 0;
            while (p0 < order.size() && pushed < PUSH_CAP) {
                const int room = FIFO - (tl - fifoHead.load(std::memory_order_acquire));
                if (room <= 0) break;
                int have = 0, took = 0; size_t resume = p0 + 64;
                for (size_t q = p0; q < std::min(p0 + 64, order.size()); ++q) {
                    const int v = tag[order[q]].load(std::memory_order_acquire);
                    const bool cand = v >= 0 || (is_tag(v) && v >= oldestTag);
                    if (!cand) continue;
                    if (have < room) { const int t = tl + have; fifoPos[t % FIFO] = (int)q; fifoSeq[t % FIFO] = seq0; fifoWait[t % FIFO] = v < 0; ringPos[t % RING] = (int)q; ++took; }
                    else if (have == room) resume = q;                     // first candidate that did not fit
                    ++have;
                }
                tl += took; pushed += took;
                if (took < have) { p0 = resume; break; }
                p0 += 64;
            }
            pickPos = std::min(p0, order.size());
            if (p0 >= order.size() && pushed == 0) exhausted.store(1, std::memory_order_release);
            fifoTail.store(tl, std::memory_order_release);
            pickMu.unlock();
        }
    }
    void worker() {
        Ctx c; c.rd.assign(ntiles, 0); c.wr.assign(ntiles, 0); c.rng += (unsigned long long)(size_t)&c;
        while (true) {
            int ticket, pos = -1, startSeq; bool wait = false;
            if (fifoMode) {
                if (!take_fifo(ticket, pos, startSeq, wait)) return;
                c.T = TAG0 + pos;
                const int v0 = tag[order[pos]].load(std::memory_order_acquire);
                if (!wait && !(v0 >= 0 || (is_tag(v0) && v0 > c.T))) wait = true;      // the kernel re-reads the seed before it speculates
            } else {
                std::lock_guard<std::mutex> lk(pickMu);
                // BEFORE the scan: the scan's "this pixel is unused" is the first read of the speculative run, and a region that commits
                // between the scan and a later read of commitSeq would escape the stamp test (found by this prototype: a seed grown from a
                // pixel that an older region had claimed and committed in that window)
                startSeq = commitSeq.load(std::memory_order_acquire);
                const int c0 = commitTicket.load(std::memory_order_acquire);
                const int oldestTag = c0 == tickets ? INT_MAX : TAG0 + ringPos[c0 % RING];      // claims below it are committed
                size_t q = pickPos;
                for (; q < order.size(); ++q) {
                    const int v = tag[order[q]].load(std::memory_order_acquire);
                    if (v >= 0) break;                                          // unused
                    if (is_tag(v) && v >= oldestTag) { wait = true; break; }    // claimed by a region still in flight: may come back
                }
                if (q >= order.size()) { pickPos = q; return; }
                pos = (int)q; pickPos = q + 1; ticket = tickets++; ringPos[ticket % RING] = pos;
            }
            c.T = TAG0 + pos;
            const int idx = order[pos];
            touch_r(c, idx % w, idx / w);
            Rect rec; bool emit = false;
            if (!wait) emit = body(c, idx, rec);
            while (commitTicket.load(std::memory_order_acquire) != ticket) std::this_thread::yield();
            bool redo = wait, ran = !wait;
            if (!wait && validate) for (int t : c.rdl) if (tileStamp[t].load(std::memory_order_acquire) > startSeq) { redo = true; break; }
            if (redo) {
                for (auto& rp : c.reg) unclaim(c, rp.x, rp.y);            // roll the speculative claims back (their tiles stay in the write set)
                c.reg.clear(); emit = false; ran = false;
                const int v = tag[idx].load(std::memory_order_acquire);
                if (v >= 0 || (is_tag(v) && v > c.T)) { emit = body(c, idx, rec); ran = true; }   // oldest in flight now: this run is the sequential one
                if (wait) ++nWait; else ++nRedo;
            } else ++nSpecOk;
            const int seq = commitSeq.load(std::memory_order_relaxed) + 1;
            for (int t : c.wrl) tileStamp[t].store(seq, std::memory_order_release);
            commitSeq.store(seq, std::memory_order_release);
            if (emit) emitted.push_back(rec);
            if (ran) seedLog.push_back({pos, (int)emit, (int)c.reg.size()});
            for (int t : c.rdl) c.rd[t] = 0; for (int t : c.wrl) c.wr[t] = 0; c.rdl.clear(); c.wrl.clear(); c.reg.clear();
            commitTicket.store(ticket + 1, std::memory_order_release);
        }
    }
    void run(const Img8& image, int nthreads, std::vector<Seg4f>& lines) {
        prec = M_PI * ANG_TH / 180; p = ANG_TH / 180;
        const double rho = QUANT / std::sin(prec), sigma = SIGMA_SCALE / SCALE;
        const unsigned hk = (unsigned)(std::ceil(sigma * std::sqrt(2 * 3.0 * std::log(10.0))));
        scaled = resize_linear_exact_8u(gaussian_blur_8u(image, 1 + 2 * hk, sigma), SCALE, SCALE);
        w = scaled.w; h = scaled.h;
        ll_angle(rho);
        LOG_NT = 5 * (std::log10(double(w)) + std::log10(double(h))) / 2 + std::log10(11.0);
        min_reg_size = size_t(-LOG_NT / std::log10(p));
        tw = (w + TS - 1) / TS; ntiles = tw * ((h + TS - 1) / TS);
        tag = std::vector<std::atomic<int>>((size_t)w * h); tileStamp = std::vector<std::atomic<int>>(ntiles);
        for (size_t i = 0; i < tag.size(); ++i) tag[i].store(angles[i] == NOTDEF ? NOTDEF_T : 0);
        for (auto& t : tileStamp) t.store(0);
        commitSeq = 0; commitTicket = 0; pickPos = 0; tickets = 0; emitted.clear(); seedLog.clear(); nSpecOk = nRedo = nWait = 0; fifoHead = 0; fifoTail = 0; exhausted = 0;
        std::vector<std::thread> th;
        for (int i = 0; i < nthreads; ++i) th.emplace_back([this] { worker(); });
        for (auto& t : th) t.join();
        lines.clear();
        for (Rect rec : emitted) {                                  // the rest of flsd(): not part of the sequential dependency
            double log_nfa = rect_improve(rec);
            if (log_nfa <= LOG_EPS) continue;
            rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
            rec.x1 /= SCALE; rec.y1 /= SCALE; rec.x2 /= SCALE; rec.y2 /= SCALE; rec.width /= SCALE;
            lines.push_back({float(rec.x1), float(rec.y1), float(rec.x2), float(rec.y2)});
        }
    }
};

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: lsd_spec_proto frame.raw [threads] [repeats]\n"); return 2; }
    const int nth = argc > 2 ? atoi(argv[2]) : 8, reps = argc > 3 ? atoi(argv[3]) : 5, chaos = argc > 4 ? atoi(argv[4]) : 0;
    const bool validate = !(argc > 5 && atoi(argv[5]) == 0);
    const bool fifo = argc > 6 && atoi(argv[6]) != 0;
    FILE* f = fopen(argv[1], "rb"); if (!f) return 2;
    Img8 im(640, 480); if (fread(im.d.data(), 1, im.d.size(), f) != im.d.size()) return 2; fclose(f);
    Lsd ref; std::vector<Seg4f> want; ref.detect(im, want);
    // one thread = the sequential algorithm in this code: it must reproduce the oracle's segments, and it is the reference for the strong
    // comparison of the threaded runs (every rectangle handed to rect_improve, every seed's final region size, the final claim map)
    Spec one; std::vector<Seg4f> seq; one.run(im, 1, seq);
    bool okSeq = seq.size() == want.size();
    for (size_t i = 0; okSeq && i < seq.size(); ++i) okSeq = memcmp(&seq[i], &want[i], sizeof(Seg4f)) == 0;
    printf("1 thread: %zu segments (%zu expected) %s, %zu rectangles, %zu seeds ran\n", seq.size(), want.size(), okSeq ? "EQUAL" : "MISMATCH", one.emitted.size(), one.seedLog.size());
    int bad = !okSeq;
    for (int r = 0; r < reps; ++r) {
        Spec s; s.chaos = chaos; s.validate = validate; s.fifoMode = fifo; std::vector<Seg4f> got; s.run(im, nth, got);
        bool same = got.size() == want.size();
        for (size_t i = 0; same && i < got.size(); ++i) same = memcmp(&got[i], &want[i], sizeof(Seg4f)) == 0;
        bool rects = s.emitted.size() == one.emitted.size();
        for (size_t i = 0; rects && i < s.emitted.size(); ++i) rects = memcmp(&s.emitted[i], &one.emitted[i], sizeof(Rect)) == 0;
        bool seeds = s.seedLog.size() == one.seedLog.size();
        for (size_t i = 0; seeds && i < s.seedLog.size(); ++i) seeds = memcmp(&s.seedLog[i], &one.seedLog[i], sizeof(Spec::SeedLog)) == 0;
        bool map = true;
        for (size_t i = 0; map && i < s.tag.size(); ++i) map = (s.tag[i].load() < 0) == (one.tag[i].load() < 0);      // used / notdef pattern
        const bool all = same && rects && seeds && map;
        printf("rep %d: segments %s rectangles %s seeds %s claim-map %s | %ld speculative ok, %ld redone, %ld waited\n", r, same ? "EQUAL" : "MISMATCH",
               rects ? "EQUAL" : "MISMATCH", seeds ? "EQUAL" : "MISMATCH", map ? "EQUAL" : "MISMATCH", s.nSpecOk, s.nRedo, s.nWait);
        bad += !all;
    }
    return bad ? 1 : 0;
}
