// CPU check of csrc/lsd_align_win.h (the header the HIP rectangle counter uses): for random (theta, prec) the integer windows must
// reproduce the reference predicate (oracle/lsd_oracle.cpp:68-76 = OpenCV isAligned) on
//   * every angle value the gradient table can hold (fastAtan2 of integer gradients, cvleaf.h),
//   * the neighbours (+-3 bit patterns) of every window end point,
//   * random bit patterns in [0, bits(360)].
// usage: align_win_test <cases> <seed>      prints "mismatches: N of M tests, max windows W"
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include "cvleaf.h"
#include "../../structure-slam-pointline_amd/csrc/lsd_align_win.h"

static uint64_t rs;
static double rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)(rs >> 11) * 0x1p-53; }

int main(int argc, char** argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 2000;
    rs = argc > 2 ? strtoull(argv[2], 0, 10) * 0x9E3779B97F4A7C15ull + 1 : 88172645463325252ull;
    // distinct table angles (as bit patterns), gradient components in [-510, 510]
    std::vector<int> tab;
    for (int gy = -510; gy <= 510; ++gy) for (int gx = -510; gx <= 510; ++gx) {
        if (gx == 0 && gy == 0) continue;
        const float a = orc::fast_atan2((float)gx, (float)(-gy));
        tab.push_back(alnwin::float_to_bits(a));
    }
    std::sort(tab.begin(), tab.end()); tab.erase(std::unique(tab.begin(), tab.end()), tab.end());
    long bad = 0, tests = 0; int maxw = 0, fails = 0;
    const double PI = alnwin::A_PI;
    for (long c = 0; c < cases; ++c) {
        // theta: region2rect yields [0, 3pi); also negative values and values glued to the window edges
        double theta;
        const int mode = (int)(rnd() * 8);
        if (mode == 0) theta = (rnd() - 0.5) * 0.9;                    // around 0
        else if (mode == 1) theta = 2 * PI + (rnd() - 0.5) * 0.9;      // around 2pi
        else if (mode == 2) theta = (double)alnwin::bits_to_float(tab[(size_t)(rnd() * tab.size())]) * alnwin::A_D2R;      // exactly a table angle
        else if (mode == 3) theta = -PI + rnd() * 0.5;
        else theta = -PI + rnd() * 4 * PI;
        double prec = PI * (22.5 / 180.0);
        const int h = (int)(rnd() * 7);
        for (int i = 0; i < h; ++i) prec /= 2;
        if (mode == 5) prec = rnd() * 1.5;                              // any tolerance below pi/2
        if (mode == 6) theta = prec * (rnd() < 0.5 ? 1 : -1) + (rnd() - 0.5) * 1e-9 + (rnd() < 0.5 ? 0 : 2 * PI);      // pruning edges
        int n, lo[2], hi[2];
        if (!alnwin::windows(theta, prec, n, lo, hi)) { ++fails; continue; }
        maxw = std::max(maxw, n);
        auto in_win = [&](int b) { return (n > 0 && b >= lo[0] && b <= hi[0]) || (n > 1 && b >= lo[1] && b <= hi[1]); };
        auto check = [&](int b) {
            if (b < 0 || b > alnwin::BMAX) return;
            ++tests;
            if (alnwin::aligned_ref(alnwin::bits_to_float(b), theta, prec) != in_win(b)) {
                if (++bad <= 10) printf("MISMATCH theta=%.17g prec=%.17g b=0x%08x a=%.9g ref=%d win=%d\n", theta, prec, b, alnwin::bits_to_float(b),
                                        (int)alnwin::aligned_ref(alnwin::bits_to_float(b), theta, prec), (int)in_win(b));
            }
        };
        for (int i = 0; i < n; ++i) for (int d = -3; d <= 3; ++d) { check(lo[i] + d); check(hi[i] + d); }
        check(0); check(1); check(alnwin::BMAX); check(alnwin::BMAX - 1);
        for (int i = 0; i < 64; ++i) check((int)(rnd() * (alnwin::BMAX + 1.0)));
        if (c % 16 == 0) for (int b : tab) check(b);                   // the whole table every 16th case
        else for (int i = 0; i < 2000; ++i) check(tab[(size_t)(rnd() * tab.size())]);
    }
    printf("table angles: %zu distinct\n", tab.size());
    printf("mismatches: %ld of %ld tests, max windows %d, three-window cases %d\n", bad, tests, maxw, fails);
    return bad || fails ? 1 : 0;
}
