#!/bin/bash
# Round 6, GPU call AC: LBD's blur + Sobel (depends on the source image alone) on a third stream from the START of the line call, i.e. beside the line prologue and the pyramid,
# instead of behind the NFA stage in the tail (7 ms of the critical path); the timeline of that schedule
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06ac; mkdir -p $O
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; head -2 $O/two_$n.txt | cut -c1-420; tail -1 $O/two_$n.txt; }
two default X=1
two side_early SSLAM_LBD_SOBEL=side_early
two default_b X=1
two side_early_b SSLAM_LBD_SOBEL=side_early
two early SSLAM_LBD_SOBEL=early
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt; (cd $R && SSLAM_LBD_SOBEL=side_early timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -- tools/step_check 12288 5 2 > $O/kt.log 2>&1; python tools/rocpd_timeline.py $O/kt $R/profiles/r06_final_kernel_trace_one_stream.txt $O/timeline_side_early.txt; rm -rf $O/kt)
