#!/bin/bash
# Round 6, GPU call I: the spill-free instantiations of the sequential core (96 VGPRs) for batches that leave at most four waves per SIMD resident anyway: calls of 65 .. 4 096 frames
# (lone waves up to 1 023, the throughput form beyond), BASELINE configs[3] (3 072 frames of 1280x960); A/B through SSLAM_LSD_SPILLFREE.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06i; mkdir -p $O
for v in 1 0; do SSLAM_LSD_SPILLFREE=$v timeout 300 python tools/small_batch_probe.py 96 128 256 512 1024 2048 3072 4096 > $O/small_batches_spillfree$v.txt 2>&1; tail -8 $O/small_batches_spillfree$v.txt; done
for v in 1 0; do SSLAM_LSD_SPILLFREE=$v timeout 600 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_c4_spillfree$v.json 2> $O/bench_c4_$v.err; python - <<PY
import json
d = json.loads(open('gpurun_out/r06i/bench_c4_spillfree$v.json').read().strip().splitlines()[-1]); print('c4 spillfree $v', round(d['value']), d['ms_per_step'], d['roofline']['kernels_ms_per_step'].get('k_lsd_regions'))
PY
done
timeout 600 python -m pytest tests/test_lines_gpu.py tests/test_batch_gpu.py tests/test_configs_gpu.py tests/test_edge_gpu.py -q -m gpu > $O/pytest_subset.txt 2>&1; echo "rc=$?" >> $O/pytest_subset.txt; tail -4 $O/pytest_subset.txt
