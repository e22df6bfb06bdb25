#!/bin/bash
# Round 6, GPU call H: clean A/B of the 16-byte loads in the five-candidate counter (one stream, Sobel behind the NFA stage, both D11 forms); stream priorities around the
# prologue / pyramid overlap; host batches after the two-round rule; the bench line's PCIe leg with both ring slots warmed.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06h; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants
one() { n=$1; shift; env "$@" SSLAM_LBD_SOBEL_MAIN=1 STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_$n.txt 2>&1; head -2 $O/one_$n.txt | cut -c1-420; tail -1 $O/one_$n.txt | cut -c1-110; }
one t4_v1
one dword_v1 LD_PRELOAD=$V/rect5dword.so
one t4_v0 STEP_NFA_VARIANT=0
one dword_v0 LD_PRELOAD=$V/rect5dword.so STEP_NFA_VARIANT=0
run() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_$n.txt 2>&1; head -2 $O/step_$n.txt | cut -c1-420; tail -1 $O/step_$n.txt | cut -c1-110; }
run default
run point_low STEP_POINT_PRIO=1
run line_high STEP_LINE_PRIO=-1
run line_high_point_low STEP_LINE_PRIO=-1 STEP_POINT_PRIO=1
HOST_BATCH_BENCH_FRAMES=1 timeout 300 python tools/bench_host_batch.py 24576 0 > $O/host_default.txt 2>&1; tail -4 $O/host_default.txt | cut -c1-200
HOST_BATCH_BENCH_FRAMES=1 timeout 300 python tools/bench_host_batch.py 18432 0 > $O/host_default_18432.txt 2>&1; tail -4 $O/host_default_18432.txt | cut -c1-200
timeout 900 python bench.py --no-other-workloads > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06h/bench.json').read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], d['roofline']['bound'], d['latency']['lines_extract_hipEvent'])
print({k: v for k, v in d['pcie_inclusive'].items() if 'per_s' in k})
PY
