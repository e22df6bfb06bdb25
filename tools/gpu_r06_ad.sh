#!/bin/bash
# Round 6, GPU call AD: LBD's blur + Sobel on the POINT stream behind k_describe (sslam_lines_set_lbd_deferred / sslam_lines_finish_lbd_dev: the line call stops behind the KeyLine
# stage, the descriptor kernel follows the Sobel) instead of behind the NFA stage on the line stream; A/B in tools/step_check, the timeline, the bench line
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06ad; mkdir -p $O
timeout 900 python -m pytest tests/test_lines_gpu.py tests/test_batch_gpu.py tests/test_configs_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; head -2 $O/two_$n.txt | cut -c1-420; tail -1 $O/two_$n.txt; }
two defer X=1
two nodefer STEP_LBD_DEFER=0
two defer_b X=1
two nodefer_b STEP_LBD_DEFER=0
two defer_c X=1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt; (cd $R && timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -- tools/step_check 12288 5 2 > $O/kt.log 2>&1; python tools/rocpd_timeline.py $O/kt $R/profiles/r06_final_kernel_trace_one_stream.txt $O/timeline_defer.txt; rm -rf $O/kt)
cd $R
timeout 900 python bench.py --no-cpu-baseline --no-other-workloads > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], d['config'].get('gather_check'), d.get('cpu_baseline'))"
SSLAM_LBD_DEFER=0 timeout 900 python bench.py --no-cpu-baseline --no-other-workloads --no-extras > $O/bench_nodefer.json 2> $O/bench_nodefer.err; python -c "
import json; d=json.loads(open('$O/bench_nodefer.json').read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'])"
