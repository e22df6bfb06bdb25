#!/bin/bash
# round 4, first GPU call: fused NFA launch + k_keylines LDS fix -- parity (fused form forced on the small test batches), then A/B bench lines
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O
cd $R
SSLAM_NFA_FUSED=2 timeout 900 python -m pytest tests/test_lines_gpu.py tests/test_configs_gpu.py tests/test_batch_gpu.py tests/test_pin_gpu.py -x -q -m gpu > $O/pytest_fused.txt 2>&1; tail -5 $O/pytest_fused.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_fused.json 2> $O/bench_fused.err; tail -c 300 $O/bench_fused.err
SSLAM_NFA_FUSED=0 timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_unfused.json 2>/dev/null
SSLAM_LIB=$R/structure-slam-pointline_amd/lib/variants/nfa4.so timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_fused_mw4.json 2>/dev/null
timeout 600 python bench.py --no-overlap --no-cpu-baseline --no-extras > $O/bench_fused_one_stream.json 2>/dev/null
SSLAM_LIB=$R/structure-slam-pointline_amd/lib/variants/nfa4.so timeout 600 python bench.py --no-overlap --no-cpu-baseline --no-extras > $O/bench_fused_mw4_one_stream.json 2>/dev/null
python - <<'PY'
import json
for n in ('bench_fused','bench_unfused','bench_fused_mw4','bench_fused_one_stream','bench_fused_mw4_one_stream'):
    try:
        d=json.load(open('gpurun_out/r04a/%s.json'%n)); k=d['roofline']['kernels_ms_per_step']
        print(n, round(d['value']), round(d['ms_per_step'],1), {a: round(b,1) for a,b in k.items() if b>0.5})
    except Exception as e: print(n, 'failed', e)
PY
