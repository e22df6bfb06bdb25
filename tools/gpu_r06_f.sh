#!/bin/bash
# Round 6, GPU call F: LBD's blur + Sobel on the side stream BESIDE the NFA stage (behind the core) against behind it (SSLAM_LBD_SOBEL_MAIN=1); one- and two-stream; then the bench line.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06f; mkdir -p $O
run() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_$n.txt 2>&1; head -2 $O/step_$n.txt | cut -c1-420; tail -1 $O/step_$n.txt | cut -c1-110; }
run default
run sobel_main SSLAM_LBD_SOBEL_MAIN=1
run default_again
STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_one_stream.txt 2>&1; head -2 $O/step_one_stream.txt | cut -c1-420
SSLAM_LBD_SOBEL_MAIN=1 STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_one_stream_sobel_main.txt 2>&1; head -2 $O/step_one_stream_sobel_main.txt | cut -c1-420
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06f/bench.json').read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], d['roofline']['bound'], d['roofline']['frac'], d['latency']['lines_extract_hipEvent'])
print({k: v for k, v in d['pcie_inclusive'].items() if 'per_s' in k}); print({k: (v.get('value'), v.get('ms_per_step')) for k, v in d.get('other_workloads', {}).items()}); print(d['cpu_baseline']['value'], d['cpu_baseline']['parity_vs_gpu']); print(d.get('latency_nfa_behind_core'))
PY
