"""Divergence of the NFA tail loop (build with SSLAM_EXTRA_FLAGS=-DSSLAM_LSD_STATS): per-lane iterations vs what a wave executes."""
import sys, ctypes as C; sys.path.insert(0, 'tests')
import pkg
from synth import synth_frame
fe = pkg.frontend(); ctx = fe.Context(0)
ex = fe.LineExtractor(ctx, 200)
for seed in (2000, 2001, 2002):
    img = synth_frame(seed)
    ex(img)
    out = (C.c_longlong * 8)()
    fe.lib().sslam_lines_debug_cycles(ex.h, 0, out)
    it, wmax, ev = out[5], out[6], out[7]  # useful, executed, evaluations
    print(f"seed {seed}: evals {ev}  tail iterations {it} (mean {it / max(ev, 1):.1f})  wave-executed lane-iterations {wmax}  efficiency {it / max(wmax, 1):.3f}")
