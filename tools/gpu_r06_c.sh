#!/bin/bash
# Round 6, GPU call C: issue PRIORITY for the sequential core (s_setprio inside k_lsd_regions<false>, SSLAM_LSD_PRIO).  Call B: with the point branch's kernels co-resident
# the core takes 104 ms instead of 78 and FAST 70 instead of 23 -- a zero-sum.  The core is one dependent chain per wave: each of its instructions that waits behind a guest's
# instruction lengthens it; the guests only want the slots the chain leaves empty.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06c; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants
run() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_$n.txt 2>&1; head -2 $O/step_$n.txt | cut -c1-420; tail -1 $O/step_$n.txt; }
run default
run prio3 SSLAM_LSD_PRIO=3
run prio1 SSLAM_LSD_PRIO=1
run prio3_p5120 SSLAM_LSD_PRIO=3 SSLAM_LSD_PERSIST=5120
run prio3_p4096 SSLAM_LSD_PRIO=3 SSLAM_LSD_PERSIST=4096
run prio3_pyr STEP_GATE=pyr SSLAM_LSD_PRIO=3
run prio3_pyr_p5120 STEP_GATE=pyr SSLAM_LSD_PRIO=3 SSLAM_LSD_PERSIST=5120
run prio3_pyr_p4096 STEP_GATE=pyr SSLAM_LSD_PRIO=3 SSLAM_LSD_PERSIST=4096
run prio3_pyr_p4608 STEP_GATE=pyr SSLAM_LSD_PRIO=3 SSLAM_LSD_PERSIST=4608
run prio3_pyr_mw4_p4096 LD_PRELOAD=$V/mw4.so STEP_GATE=pyr SSLAM_LSD_PRIO=3 SSLAM_LSD_PERSIST=4096
run prio3_pyr_mw5_p5120 LD_PRELOAD=$V/mw5.so STEP_GATE=pyr SSLAM_LSD_PRIO=3 SSLAM_LSD_PERSIST=5120
run prio3_none_p4096 STEP_GATE=none SSLAM_LSD_PRIO=3 SSLAM_LSD_PERSIST=4096
run prio3_pyr_p4096_sobel_early STEP_GATE=pyr SSLAM_LSD_PRIO=3 SSLAM_LSD_PERSIST=4096 SSLAM_LBD_SOBEL=early
SSLAM_LSD_PRIO=3 STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_prio3_one_stream.txt 2>&1; head -2 $O/step_prio3_one_stream.txt | cut -c1-420
cd /tmp && export TMPDIR=/tmp
for v in "prio3_pyr_p4096 4096" "prio3_pyr_p5120 5120"; do set -- $v
rm -rf $O/kt; (cd $R && STEP_GATE=pyr SSLAM_LSD_PRIO=3 SSLAM_LSD_PERSIST=$2 timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -- tools/step_check 12288 3 1 > $O/kt_$1.log 2>&1; python tools/rocpd_timeline.py $O/kt $R/profiles/r05_final_kernel_trace_B12288_one_stream.txt $O/timeline_$1.txt | head -40; rm -rf $O/kt)
done
