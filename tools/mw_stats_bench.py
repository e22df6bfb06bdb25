"""Multi-wave LSD core statistics over the bench's 64 frames (library built with -DSSLAM_MW_STATS via tools/build_variant.sh)."""
import sys, ctypes as C; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, pkg, bench
fe = pkg.frontend(); ctx = fe.Context(0)
cur, prev = bench.synth_frames(640, 480, 64, 0)
ex = fe.LineExtractor(ctx, 200)
acc = np.zeros(8, dtype=np.int64); why = np.zeros(8, dtype=np.int64)
for f in cur:
    ex(f); out = (C.c_longlong * 8)(); fe.lib().sslam_lines_debug_cycles(ex.h, 0, out)
    o = [int(x) for x in out]
    acc += np.array([o[0] & 0xFFFFFFFF, o[0] >> 32, o[1] & 0xFFFFFFFF, o[1] >> 32, o[7] & 0xFFFF, (o[7] >> 16) & 0xFFFF, o[5], o[6]], dtype=np.int64)
    ev = (o[7] >> 32) & 0xFFFF; oth = (o[7] >> 48) & 0xFFFF
    extra = globals().setdefault("extra", np.zeros(2)); extra += (ev, oth)
    why += np.array([(o[3 + c // 4] >> (16 * (c % 4))) & 0xFFFF for c in range(8)], dtype=np.int64)
n = len(cur)
print("per frame: taken %.0f regions / %.0f px; grown by the main wave %.0f / %.0f px, of which >= 100 points: %.1f without a region, %.1f with a point used; helpers busy %.1f idle %.1f Mcyc" % (
    acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n, acc[6] / n / 1e6, acc[7] / n / 1e6))
print(">= 100 points: after a refine event %.1f, other %.1f" % tuple(extra / n))
print("helper skips per frame: in map at scan %.1f, used at turn %.1f, in map at turn %.1f, results full %.1f, arena full %.1f, growth gave up %.1f, refine gave up %.1f, main passed %.1f" % tuple(why / n))
