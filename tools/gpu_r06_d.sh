#!/bin/bash
# Round 6, GPU call D: the guest form of the core as the default when a co-running branch is announced (k_lsd_regions<false, 4>, 16 persistent workgroups per CU, pyramid beside
# the prologue, LBD's blur + Sobel on the side stream under the core): A/B against the round-5 schedule on this box, grid sizes, then the whole GPU suite and a short bench line.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06d; mkdir -p $O
run() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_$n.txt 2>&1; head -2 $O/step_$n.txt | cut -c1-420; tail -1 $O/step_$n.txt | cut -c1-110; }
run default
run r05_schedule STEP_GATE=core SSLAM_LSD_GUEST=0
run pyr_noguest SSLAM_LSD_GUEST=0
run guest_sobel_main SSLAM_LBD_SOBEL_MAIN=1
run guest_p3584 SSLAM_LSD_PERSIST=3584
run guest_p4608 SSLAM_LSD_PERSIST=4608
run guest_p5120 SSLAM_LSD_PERSIST=5120
run guest_prio STEP_POINT_PRIO=-1
STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_one_stream.txt 2>&1; head -2 $O/step_one_stream.txt | cut -c1-420
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt; (cd $R && timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -- tools/step_check 12288 3 1 > $O/kt.log 2>&1; python tools/rocpd_timeline.py $O/kt $R/profiles/r05_final_kernel_trace_B12288_one_stream.txt $O/timeline_default.txt | head -40; rm -rf $O/kt)
cd $R
timeout 60 tools/gather_probe 16 6 > $O/gather_probe.json 2>&1; cat $O/gather_probe.json
timeout 1200 python -m pytest tests -q -m gpu -x --durations=8 > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt; tail -16 $O/pytest_gpu.txt
timeout 600 python bench.py --no-other-workloads --no-pcie > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06d/bench.json').read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], d['roofline']['bound'], d['roofline']['frac'], d['roofline']['bound_note'])
print(json.dumps(d['roofline']['issue'])[:600]); print(json.dumps(d['roofline']['random_sector'])[:900]); print(d['cpu_baseline']['value'], d['cpu_baseline']['ms_per_frame'], d['cpu_baseline']['parity_vs_gpu'])
PY
for m in pyr 1; do SSLAM_POINTS_AT_CORE=$m timeout 600 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_c4_$m.json 2> $O/bench_c4_$m.err; python - <<PY
import json
d = json.loads(open('gpurun_out/r06d/bench_c4_$m.json').read().strip().splitlines()[-1]); print('c4 gate $m', round(d['value']), d['ms_per_step'])
PY
done
