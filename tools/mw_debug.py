"""Multi-wave LSD core (lsd_regions.h): runs fuzz cases through the line extractor, compares the raw LSD segments with the oracle and
prints the main wave's statistics (regions taken from helpers / grown itself / chunks abandoned).  usage: mw_debug.py [ncases] [seed] [only_it]"""
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import numpy as np, ctypes as C, pkg, oracle_lib
from fuzz_parity import cases
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
fe = pkg.frontend(); ctx = fe.Context(0); orc = oracle_lib.Oracle()
nbad = 0
for it, img, nfeat, nlev, sf, ini, mn, cap in cases(n, rng):
    if only >= 0 and it != only: continue
    lx = fe.LineExtractor(ctx, cap)
    reps = 5 if only >= 0 else 1
    for rep in range(reps):
        kl, ld, fn = lx(img); raw = lx.debug_segments(0)
        out = (C.c_longlong * 8)(); fe.lib().sslam_lines_debug_cycles(lx.h, 0, out)
        okl, old, ofn, oraw = orc.lines_extract(img, cap)
        same = raw.shape == oraw.shape and np.array_equal(raw, oraw)
        nbad += not same
        if os.environ.get("MW_STATS"):
            names = ["taken", "own (no region or not valid)", "", "", "", "", "", "own >= 100 points"]
            why = [(out[3 + c // 4] >> (16 * (c % 4))) & 0xFFFF for c in range(8)]
            print("   helper skips: in map at scan %d, used at turn %d, in map at turn %d, results full %d, arena full %d, growth gave up %d, refine gave up %d, main passed %d" % tuple(why))
            print("   " + "  ".join("%s %d/%dpx" % (names[c], out[c] & 0xFFFFFFFF, out[c] >> 32) for c in (0, 1, 7)) + "   main %.2f Mcyc, helpers busy %.2f idle %.2f Mcyc" % (out[2] / 1e6, out[5] / 1e6, out[6] / 1e6))
        print("it %d %s rep %d: %s  segs %d/%d  taken %d (%d px) own %d (%d px) badChunks %d" % (it, img.shape, rep, "ok" if same else "DIFF", len(raw), len(oraw), out[5] & 0xFFFFFFFF, out[5] >> 32, out[6] & 0xFFFFFFFF, out[6] >> 32, out[7]))
        if not same and raw.shape == oraw.shape:
            d = np.nonzero((raw != oraw).any(axis=1))[0]; print("   first differing segments", d[:6], raw[d[:2]], oraw[d[:2]])
    lx.close()
print("bad", nbad)
