#!/bin/bash
# round 4, call d: the sequential core at 7 / 8 waves per SIMD (72 / 64 VGPRs) against 6, at the bench's batch and at batches that fill the larger slot counts; group test re-run
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O
cd $R
timeout 420 python -m pytest tests/test_group_gpu.py -x -q -m gpu > $O/pytest_group.txt 2>&1; tail -3 $O/pytest_group.txt
V=$R/structure-slam-pointline_amd/lib/variants
for v in "w6:" "w7:SSLAM_LIB=$V/core7.so" "w8:SSLAM_LIB=$V/core8.so"; do
  n=${v%%:*}; e=${v#*:}
  env $e timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench_$n.json 2>/dev/null
  env $e timeout 300 python bench.py --no-cpu-baseline --no-extras --no-overlap > $O/bench_${n}_one.json 2>/dev/null
done
SSLAM_LIB=$V/core7.so timeout 300 python bench.py --no-cpu-baseline --no-extras --batch 14336 > $O/bench_w7_B14336.json 2>/dev/null
SSLAM_LIB=$V/core8.so timeout 300 python bench.py --no-cpu-baseline --no-extras --batch 16384 > $O/bench_w8_B16384.json 2>/dev/null
SSLAM_LIB=$V/core7.so timeout 300 python -m pytest tests/test_lines_gpu.py tests/test_batch_gpu.py -x -q -m gpu > $O/pytest_w7.txt 2>&1; tail -2 $O/pytest_w7.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04d/bench_*.json')):
    try:
        d=json.load(open(f)); k=d['roofline']['kernels_ms_per_step']
        print(f.split('/')[-1], d['config']['batch_per_gpu'], round(d['value']), round(d['ms_per_step'],1), 'core', round(k.get('k_lsd_regions',0),1))
    except Exception as e: print(f, 'failed', e)
PY
