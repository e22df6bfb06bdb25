#!/bin/bash
# Round 6, GPU call B: the point branch as GUESTS under the sequential core.  Call A's timeline showed why leaving room alone gains little (persistent core grids of 5 120 / 4 096
# workgroups: 165.7 / 162.8 ms against 167.8): the first guest is k_resize, a latency-bound kernel that takes 59 ms at two waves per SIMD (9 ms alone), so FAST -- the kernel that
# could use the core's idle issue slots -- still starts 60 ms into the core.  Here the pyramid is built BESIDE the line prologue (STEP_GATE=pyr) and FAST .. matching start with the core.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06b; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants
run() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_$n.txt 2>&1; head -2 $O/step_$n.txt | cut -c1-400; tail -1 $O/step_$n.txt; }
run default
run pyr STEP_GATE=pyr
run pyr_p5120 STEP_GATE=pyr SSLAM_LSD_PERSIST=5120
run pyr_p4608 STEP_GATE=pyr SSLAM_LSD_PERSIST=4608
run pyr_p4096 STEP_GATE=pyr SSLAM_LSD_PERSIST=4096
run pyr_p3584 STEP_GATE=pyr SSLAM_LSD_PERSIST=3584
run pyr_p4096_prio STEP_GATE=pyr SSLAM_LSD_PERSIST=4096 STEP_POINT_PRIO=-1
run pyr_p5120_prio STEP_GATE=pyr SSLAM_LSD_PERSIST=5120 STEP_POINT_PRIO=-1
run pyr_mw4_p4096 LD_PRELOAD=$V/mw4.so STEP_GATE=pyr SSLAM_LSD_PERSIST=4096
run none_p4096 STEP_GATE=none SSLAM_LSD_PERSIST=4096
run pyr_p4096_sobel_early STEP_GATE=pyr SSLAM_LSD_PERSIST=4096 SSLAM_LBD_SOBEL=early
cd /tmp && export TMPDIR=/tmp
for v in "pyr_p4096 4096" "pyr_p5120 5120"; do set -- $v
rm -rf $O/kt; (cd $R && STEP_GATE=pyr SSLAM_LSD_PERSIST=$2 timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -- tools/step_check 12288 3 1 > $O/kt_$1.log 2>&1; python tools/rocpd_timeline.py $O/kt $R/profiles/r05_final_kernel_trace_B12288_one_stream.txt $O/timeline_$1.txt | head -40; rm -rf $O/kt)
done
