"""Cluster form of the LSD core (lsd_cluster.h, SSLAM_LSD_CLUSTER=1): parity of the bench frames against the oracle, latency per frame and the main
wave's counters (taken / own growth / refused results / chunks it claimed itself / polls spent waiting / bounded waits that expired).
usage: cl_probe.py [nframes=24] [--nocheck]"""
import sys, os, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, ctypes as C, torch, pkg, bench, oracle_lib
torch.cuda.set_device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 24
fe = pkg.frontend(); ctx = fe.Context(0)
cur, prev = bench.synth_frames(640, 480, 64, 0)
lx = fe.LineExtractor(ctx, 200)
orc = None if "--nocheck" in sys.argv else oracle_lib.Oracle()
bad = 0; ts = []; acc = np.zeros(8); cyc = np.zeros(5); hst = np.zeros(8, np.int64); stg = np.zeros(4)
for i, f in enumerate(cur[:n]):
    lx(f)
    t0 = time.perf_counter(); kl, ld, fn = lx(f); ts.append((time.perf_counter() - t0) * 1e3)
    out = (C.c_longlong * 8)(); fe.lib().sslam_lines_debug_cycles(lx.h, 0, out); o = [int(x) for x in out]
    cyc += np.array(o[:5], float)
    if "--cycles" in sys.argv:
        o2 = (C.c_longlong * 8)(); fe.testing_lib().sslam_lines_debug_cluster(lx.h, 0, o2); hst += np.array([int(x) for x in o2], np.int64)
    stg += np.array([o[3] & 0xFFFFFFFF, o[3] >> 32, o[4] & 0xFFFFFFFF, o[4] >> 32], float)
    acc += np.array([o[5] & 0xFFFFFFFF, o[5] >> 32, o[6] & 0xFFFFFFFF, o[6] >> 32, o[2], o[1], o[0], o[7]], float)
    if orc is not None:
        okl, old, ofn, oraw = orc.lines_extract(f, 200)
        same = np.array_equal(lx.debug_segments(0), oraw) and np.array_equal(ld, old) and np.array_equal(fn, ofn)
        bad += int(not same)
        if not same: print("frame", i, "DIFFERS: segments", len(lx.debug_segments(0)), "oracle", len(oraw))
ts = np.array(ts)
if "--cycles" not in sys.argv:
    print("feeder: chunks staged %.0f, takes validated from LDS %.0f (gathered again after a nearby commit: %.0f); results taken from a helper's second publication %.1f per frame" % (stg[0] / n, stg[2] / n, stg[1] / n, stg[3] / n))
if "--cycles" in sys.argv:      # library built with -DSSLAM_CL_CYCLES: cyc[0..4] = wait, validate + take, own growth, own rect / refine / commit, total
    tot = cyc[4]
    print("main wave %.2f Mcycles/frame: wait %.1f%%  validate+take %.1f%%  own growth %.1f%%  own rect/refine/commit %.1f%%  seed scans and the rest %.1f%%" % (
        tot / n / 1e6, 100 * cyc[0] / tot, 100 * cyc[1] / tot, 100 * cyc[2] / tot, 100 * cyc[3] / tot, 100 * (tot - cyc[:4].sum()) / tot))
    lo = lambda v: (v & 0xFFFFFFFF) / n; hi = lambda v: (v >> 32) / n
    print("helpers per frame: seeds left when results / arena were full %.1f, growth gave up %.1f (%.0f px grown before), refine gave up %.1f, seeds left when the main wave passed %.1f;  main wave's own growth: after a refused result %.1f (%.0f px), all %.1f (%.0f px)" % (
        hst[0] / n, lo(hst[1]), hi(hst[1]), hst[2] / n, hst[3] / n, lo(hst[4]), hi(hst[4]), lo(hst[5]), hi(hst[5])))
print("form %s: %d frames, lines_extract host p50 %.2f ms p90 %.2f ms min %.2f; per frame: taken %.0f (%.0f px) own %.0f (%.0f px) refused %.1f own-chunks %.1f wait-polls %.0f expired %.1f; differing from the oracle: %s" % (
    os.environ.get("SSLAM_LSD_CLUSTER", "mw"), n, np.percentile(ts, 50), np.percentile(ts, 90), ts.min(), *(acc / n), bad if orc is not None else "not checked"))
