#!/bin/bash
# Round 6, GPU call U: does a kernel of another stream start only when the running kernel has issued its LAST workgroup?  The final timeline shows the pyramid's first k_resize
# running alone for 4.3 ms before the line branch starts, and its second one starting 4.3 ms after the gradient kernel did.  Grids the chip holds at once (workgroups walking
# the frames) for the fused gradient kernel and for k_resize: SSLAM_GRAD_GRID_WGS / SSLAM_RESIZE_GRID_WGS.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06u; mkdir -p $O
run() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_$n.txt 2>&1; head -2 $O/step_$n.txt | cut -c1-420; tail -1 $O/step_$n.txt | cut -c1-110; }
run default
run grad1024 SSLAM_GRAD_GRID_WGS=1024
run grad2048 SSLAM_GRAD_GRID_WGS=2048
run resize4096 SSLAM_RESIZE_GRID_WGS=4096
run both SSLAM_GRAD_GRID_WGS=1024 SSLAM_RESIZE_GRID_WGS=4096
run both_b SSLAM_GRAD_GRID_WGS=2048 SSLAM_RESIZE_GRID_WGS=8192
for v in "" "SSLAM_GRAD_GRID_WGS=1024" "SSLAM_RESIZE_GRID_WGS=4096"; do env $v STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 2>&1 | head -2 | tail -1 | grep -o "k_lsd_grad [0-9.]*\|k_resize [0-9.]*" | tr '\n' ' '; echo " ($v)"; done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt; (cd $R && SSLAM_GRAD_GRID_WGS=1024 SSLAM_RESIZE_GRID_WGS=4096 timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -- tools/step_check 12288 3 1 > $O/kt.log 2>&1; python tools/rocpd_timeline.py $O/kt $R/profiles/r06_final_kernel_trace_one_stream.txt $O/timeline_both.txt | head -34; rm -rf $O/kt)
