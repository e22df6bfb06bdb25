"""Stage clocks of the multi-wave LSD core's MAIN wave over the bench frames (library built with -DSSLAM_LSD_CYCLES: tools/build_variant.sh)."""
import sys, ctypes as C; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, pkg, bench
fe = pkg.frontend(); ctx = fe.Context(0)
cur, prev = bench.synth_frames(640, 480, 64, 0)
ex = fe.LineExtractor(ctx, 200)
acc = np.zeros(8)
for f in cur[:32]:
    ex(f); out = (C.c_longlong * 8)(); fe.lib().sslam_lines_debug_cycles(ex.h, 0, out); acc += np.array(list(out), float)
tot = acc[4]
print("main wave: total %.2f Mcycles/frame; grow-or-take %.1f%% (wait for helper %.1f%%, validate+take %.1f%%, own growth %.1f%%)  rect %.1f%%  refine %.1f%% (reduce %.1f%%)  rest %.1f%%" % (
    tot / 32 / 1e6, 100 * acc[0] / tot, 100 * acc[5] / tot, 100 * acc[6] / tot, 100 * acc[7] / tot, 100 * acc[1] / tot, 100 * acc[2] / tot, 100 * acc[3] / tot, 100 * (tot - acc[0] - acc[1] - acc[2]) / tot))
