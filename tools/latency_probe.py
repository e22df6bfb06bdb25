"""Single-frame latency of the host entry points over the bench's 64 varied frames (what bench.py reports as `latency`), plus fuzz-style
parity of the same frames against the oracle for the line extractor.  Environment knobs pass through (e.g. SSLAM_LSD_NO_BITMAP=1)."""
import sys, os, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch, pkg, bench, oracle_lib
torch.cuda.set_device(0)
fe = pkg.frontend(); ctx = fe.Context(0)
cur, prev = bench.synth_frames(640, 480, 64, 0)
bench.NFEAT, bench.NLINES = 1000, 200
r = bench.latency_leg(fe, ctx, cur, True, nframes=128)
print({k: (round(v["p50"], 2), round(v["p90"], 2)) for k, v in r.items() if isinstance(v, dict)}, "fps", round(r["frames_per_s_one_at_a_time"], 1))
if "--check" in sys.argv:
    orc = oracle_lib.Oracle(); lx = fe.LineExtractor(ctx, 200); bad = 0
    for i, f in enumerate(cur[:24]):
        kl, ld, fn = lx(f); okl, old, ofn, oraw = orc.lines_extract(f, 200)
        bad += int(not (np.array_equal(lx.debug_segments(0), oraw) and np.array_equal(ld, old) and np.array_equal(fn, ofn)))
    print("frames differing from the oracle:", bad)
