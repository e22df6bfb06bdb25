"""Where the helper waves of the multi-wave LSD core spend their cycles, summed over the helpers, per frame (bench frames).
Library built with -DSSLAM_MW_STATS -DSSLAM_MW_HCYC (tools/build_variant.sh)."""
import sys, ctypes as C; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, pkg, bench
fe = pkg.frontend(); ctx = fe.Context(0)
cur, prev = bench.synth_frames(640, 480, 64, 0)
ex = fe.LineExtractor(ctx, 200)
acc = np.zeros(8); pub = np.zeros(2)
for f in cur[:32]:
    ex(f); out = (C.c_longlong * 8)(); fe.lib().sslam_lines_debug_cycles(ex.h, 0, out); o = [int(x) for x in out]
    pub += (o[7] & 0xFFFFFFFF, o[7] >> 32); o[7] = 0; acc += np.array(o, float)
names = ["slot wait + reap", "claim + scan", "wait for ring space", "region_grow", "region2rect / refine", "publish (marks, box, map)", "insurance polling", "other"]
print("published per frame: %.0f regions, %.0f points (lists A + B)" % tuple(pub / 32))
print("helpers, Mcycles per frame (all helpers together): " + ", ".join("%s %.1f" % (n, a / 32 / 1e6) for n, a in zip(names, acc)) + "; total %.1f" % (acc.sum() / 32 / 1e6))
