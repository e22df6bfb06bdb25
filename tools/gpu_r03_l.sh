#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03l; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_batch_gpu.py -x -q -m gpu 2>&1 | tail -5
SSLAM_LIB=$R/structure-slam-pointline_amd/lib/variants/mwstats.so timeout 200 python tools/mw_stats_bench.py 2>&1 | tail -4
SSLAM_LIB=$R/structure-slam-pointline_amd/lib/variants/lsdcyc.so timeout 200 python tools/mw_cycles.py 2>&1 | tail -2
timeout 600 python - <<'PY' 2>&1 | tail -3
import sys, json; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import pkg, bench, torch
torch.cuda.set_device(0)
fe = pkg.frontend(); ctx = fe.Context(0)
cur, prev = bench.synth_frames(640, 480, 64, 0)
bench.NFEAT, bench.NLINES = 1000, 200
print(json.dumps(bench.pcie_leg(fe, ctx, cur, True)))
PY
