#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_lines_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -4
import sys, os, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch, pkg, bench, oracle_lib
torch.cuda.set_device(0)
fe = pkg.frontend(); ctx = fe.Context(0)
cur, prev = bench.synth_frames(1280, 960, 16, 0)
orc = oracle_lib.Oracle()
for flav in ("cl", "mw"):
    os.environ["SSLAM_LSD_FLAVOUR"] = flav
    lx = fe.LineExtractor(ctx, 400); ts = []; bad = 0
    for i, f in enumerate(cur):
        lx(f); t0 = time.perf_counter(); kl, ld, fn = lx(f); ts.append((time.perf_counter() - t0) * 1e3)
        if i < 4:
            okl, old, ofn, oraw = orc.lines_extract(f, 400)
            bad += int(not (np.array_equal(lx.debug_segments(0), oraw) and np.array_equal(ld, old)))
    print(flav, "1280x960 lines_extract host p50 %.2f ms p90 %.2f; frames differing from the oracle (4 checked): %d" % (np.percentile(ts, 50), np.percentile(ts, 90), bad))
    lx.close()
PY
