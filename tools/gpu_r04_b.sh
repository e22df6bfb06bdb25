#!/bin/bash
# round 4, call b: the in-process RCCL stand-in (N > 1 group paths), the stress / fuzz tests of the driver-run suite, bench line with other_workloads
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_group_gpu.py tests/test_pin_gpu.py -x -q -m gpu > $O/pytest_group.txt 2>&1; tail -15 $O/pytest_group.txt
timeout 900 python -m pytest tests/test_stress_gpu.py -x -q -m gpu -s > $O/pytest_stress.txt 2>&1; tail -8 $O/pytest_stress.txt
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04b/bench.json'))
print(round(d['value']), d['ms_per_step'], d.get('latency',{}).get('lines_extract_hipEvent'), d.get('other_workloads'))
print(d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('parity_vs_gpu'), d.get('pcie_inclusive',{}).get('pinned_frames_per_s'))
PY
