#!/bin/bash
# Round 6, GPU call E: the guest form as shipped (LBD's Sobel back in the tail), counting-sort tile sizes (the per-tile histograms are most of the sort's traffic),
# the pageable bounce ring of sslam_frontend_batch, the whole GPU suite, c4.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06e; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants
run() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_$n.txt 2>&1; head -2 $O/step_$n.txt | cut -c1-420; tail -1 $O/step_$n.txt | cut -c1-110; }
run default
run tile16k LD_PRELOAD=$V/tile16k.so
run tile32k LD_PRELOAD=$V/tile32k.so
for v in default tile16k tile32k; do p=""; [ $v != default ] && p="LD_PRELOAD=$V/$v.so"; env $p STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_${v}_one_stream.txt 2>&1; head -2 $O/step_${v}_one_stream.txt | cut -c1-420; tail -1 $O/step_${v}_one_stream.txt | cut -c1-110; done
for t in 8 16 24 48; do SSLAM_BATCH_THREADS=$t timeout 300 python tools/bench_host_batch.py 18432 > $O/host_batch_t$t.txt 2>&1; tail -4 $O/host_batch_t$t.txt | cut -c1-200; done
timeout 1200 python -m pytest tests -q -m gpu --durations=5 > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt; tail -12 $O/pytest_gpu.txt
timeout 600 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_c4.json 2> $O/bench_c4.err; python - <<PY
import json
d = json.loads(open('gpurun_out/r06e/bench_c4.json').read().strip().splitlines()[-1]); print('c4', round(d['value']), d['ms_per_step'])
PY
