#!/bin/bash
# round 4, call c: stand-in RCCL tests (fixed), stress / fuzz tests, NFA launch forms, batch SearchForInitialization in LDS, single-frame latency A/B
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O
cd $R
timeout 420 python -m pytest tests/test_group_gpu.py -x -q -m gpu > $O/pytest_group.txt 2>&1; tail -4 $O/pytest_group.txt
timeout 600 python -m pytest tests/test_lines_gpu.py tests/test_match_gpu.py tests/test_shim_gpu.py tests/test_batch_gpu.py -x -q -m gpu > $O/pytest_a.txt 2>&1; tail -4 $O/pytest_a.txt
timeout 400 python -m pytest tests/test_stress_gpu.py -x -q -m gpu -s > $O/pytest_stress.txt 2>&1; tail -4 $O/pytest_stress.txt
for v in "def:" "old:SSLAM_NFA_FUSED=0" "old256:SSLAM_NFA_FUSED=0 SSLAM_COUNT_WAVES=256 SSLAM_EVAL_WAVES=64" "old128:SSLAM_NFA_FUSED=0 SSLAM_COUNT_WAVES=128 SSLAM_EVAL_WAVES=32" "wg8:SSLAM_NFA_WAVES=8"; do
  n=${v%%:*}; e=${v#*:}
  env $e timeout 200 python tools/latency_probe.py > $O/lat_$n.txt 2>&1; tail -1 $O/lat_$n.txt
done
timeout 300 python bench.py --no-cpu-baseline --no-extras --no-overlap > $O/bench_one_stream.json 2>/dev/null
SSLAM_SFI_BATCH=global timeout 300 python bench.py --no-cpu-baseline --no-extras --no-overlap > $O/bench_one_stream_sfi_global.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench_two_streams.json 2>/dev/null
python - <<'PY'
import json
for n in ('bench_one_stream','bench_one_stream_sfi_global','bench_two_streams'):
    try:
        d=json.load(open('gpurun_out/r04c/%s.json'%n)); k=d['roofline']['kernels_ms_per_step']
        print(n, round(d['value']), round(d['ms_per_step'],1), {a: round(b,1) for a,b in k.items() if b>0.5})
    except Exception as e: print(n, 'failed', e)
PY
