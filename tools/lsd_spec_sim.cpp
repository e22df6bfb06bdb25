// What-if for single-frame LSD latency: W waves grow the next seeds speculatively, results commit in seed order, a region whose
// 3x3-dilated pixel set shares a TSxTS tile with a region committed after it started is redone.  CPU only: runs the oracle's
// flsd() with instrumentation, then replays the seed list through an event model (cost unit = one accepted pixel).
// build: g++ -O2 -std=c++17 -ffp-contract=off -Ioracle tools/lsd_spec_sim.cpp -o /tmp/lsd_spec_sim
// run:   python -c "import sys; sys.path.insert(0,'tests'); from synth import synth_frame; synth_frame(2000).tofile('/tmp/frame.raw')"
//        /tmp/lsd_spec_sim <waves> <tile> 0 <commit cost> /tmp/frame.raw
#include "../oracle/lsd_oracle.cpp"
namespace orc { int g_gaussVariant = 0; }      // (defined in orb_oracle.cpp, which this single-file build does not link)
#include <cstdio>
#include <set>
#include <random>
using namespace orc;
struct Item { bool active; int consumer; int cost; std::vector<int> tiles; int tile0; };
int TS = 8;
struct Sim : Lsd {
    std::vector<Item> items; std::vector<int> owner; int tw;
    void run(const Img8& image) {
        const double prec = M_PI * ANG_TH / 180, p = ANG_TH / 180;
        const double rho = QUANT / std::sin(prec);
        const double sigma = SIGMA_SCALE / SCALE;
        const unsigned hk = (unsigned)(std::ceil(sigma * std::sqrt(2 * 3.0 * std::log(10.0))));
        Img8 g = gaussian_blur_8u(image, 1 + 2 * hk, sigma);
        scaled = resize_linear_exact_8u(g, SCALE, SCALE);
        w = scaled.w; h = scaled.h; tw = (w + TS - 1) / TS;
        ll_angle(rho);
        LOG_NT = 5 * (std::log10(double(w)) + std::log10(double(h))) / 2 + std::log10(11.0);
        const size_t min_reg_size = size_t(-LOG_NT / std::log10(p));
        used.assign((size_t)w * h, 0); owner.assign((size_t)w * h, -1);
        std::vector<RegionPoint> reg;
        std::vector<uint8_t> before;
        for (size_t i = 0; i < order.size(); ++i) {
            const int idx = order[i], px = idx % w, py = idx / w;
            if (!(used[idx] == 0 && angles[idx] != NOTDEF)) { Item it; it.active = false; it.consumer = owner[idx]; it.cost = 0; it.tile0 = (py / TS) * tw + px / TS; items.push_back(it); continue; }
            Item it; it.active = true; it.consumer = -1; it.tile0 = (py / TS) * tw + px / TS;
            const int me = (int)items.size();
            std::set<int> tiles;
            auto addreg = [&](const std::vector<RegionPoint>& r) { for (auto& q : r) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) { int x = q.x + dx, y = q.y + dy; if (x < 0 || y < 0 || x >= w || y >= h) continue; tiles.insert((y / TS) * tw + x / TS); } };
            double reg_angle;
            region_grow(px, py, reg, reg_angle, prec);
            std::vector<RegionPoint> reg1 = reg;
            addreg(reg); int cost = (int)reg.size() + 10;
            if (reg.size() >= min_reg_size) {
                Rect rec; region2rect(reg, reg_angle, prec, p, rec);
                cost += (int)reg.size() / 8;
                bool ok = refine(reg, reg_angle, prec, p, rec, DENSITY_TH);
                if (reg.size() != reg1.size()) cost += (int)reg.size() + (int)reg1.size() / 8;
                addreg(reg);
            }
            // owner update: pixels of reg1 and reg that are now used
            for (auto& q : reg1) if (used[(size_t)q.y * w + q.x]) owner[(size_t)q.y * w + q.x] = me;
            for (auto& q : reg) if (used[(size_t)q.y * w + q.x]) owner[(size_t)q.y * w + q.x] = me;
            it.cost = cost; it.tiles.assign(tiles.begin(), tiles.end());
            items.push_back(it);
        }
    }
};
int main(int argc, char** argv) {
    int W = argc > 1 ? atoi(argv[1]) : 16; TS = argc > 2 ? atoi(argv[2]) : 8; int skipNear = argc > 3 ? atoi(argv[3]) : 0; int ovh = argc > 4 ? atoi(argv[4]) : 30;
    // synthetic frame comes from a raw file written by python
    FILE* f = fopen(argc > 5 ? argv[5] : "/tmp/frame.raw", "rb"); if (!f) { fprintf(stderr, "no frame file\n"); return 1; } Img8 im; im.w = 640; im.h = 480; im.d.resize(640 * 480); fread(im.d.data(), 1, im.d.size(), f); fclose(f);
    Sim s; s.run(im);
    long total = 0; int nact = 0; for (auto& it : s.items) if (it.active) { total += it.cost; ++nact; }
    {
        std::vector<double> commitT(s.items.size(), 0.0);
        std::vector<double> tileT(s.tw * ((s.h + TS - 1) / TS) + 1, 0.0);
        std::multiset<double> freeT; for (int k = 0; k < W; ++k) freeT.insert(0.0);
        double lastCommit = 0, lastPick = 0; long redo = 0, phantomBusy = 0; double cc = ovh;
        for (size_t j = 0; j < s.items.size(); ++j) {
            const Item& it = s.items[j];
            if (!it.active && it.consumer < 0) continue;
            double sj = std::max(*freeT.begin(), lastPick);
            if (!it.active) {
                if (commitT[it.consumer] <= sj) { commitT[j] = std::max(lastCommit, commitT[it.consumer]); continue; }
                // wave wasted until consumer commits
                freeT.erase(freeT.begin()); lastPick = sj;
                double fr = std::max(sj, std::min(sj + s.items[it.consumer].cost, commitT[it.consumer]));
                freeT.insert(fr); ++phantomBusy;
                commitT[j] = std::max(lastCommit, commitT[it.consumer]); lastCommit = commitT[j];
                continue;
            }
            freeT.erase(freeT.begin()); lastPick = sj;
            double st = sj; for (int tl : it.tiles) if (tileT[tl] > st) st = tileT[tl];
            if (st > sj) ++redo;
            double f = st + it.cost;
            double cm = std::max(f, lastCommit) + cc;
            commitT[j] = cm; lastCommit = cm;
            for (int tl : it.tiles) tileT[tl] = cm;
            freeT.insert(cm);
        }
        printf("ASYNC W=%d TS=%d commit=%d: seqcost=%ld time=%.0f speedup=%.2f redo=%ld phantomBusy=%ld active=%d\n", W, TS, ovh, total, lastCommit, total / lastCommit, redo, phantomBusy, nact);
    }
    // size distribution
    std::vector<int> cs; for (auto& it : s.items) if (it.active) cs.push_back(it.cost); std::sort(cs.begin(), cs.end());
    long acc = 0; printf("cost pct: "); for (double q : {0.5, 0.9, 0.99, 1.0}) printf("q%.2f=%d ", q, cs[std::min(cs.size() - 1, (size_t)(q * cs.size()))]); printf("\n");
    return 0;
}
