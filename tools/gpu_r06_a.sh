#!/bin/bash
# Round 6, GPU call A: this box's baseline, the kernel TIMELINE of a two-stream step, the random-sector probe, and the guest-wave headroom experiments:
# the sequential core with FEWER resident waves than the register file allows (a persistent grid with dynamic frame claiming: SSLAM_LSD_PERSIST = workgroups),
# so that the point branch's waves are co-resident from the core's first millisecond instead of waiting for core waves to retire.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06a; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants
run() { # name, env...
  n=$1; shift
  env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_$n.txt 2>&1; head -2 $O/step_$n.txt; tail -1 $O/step_$n.txt
}
timeout 60 tools/gather_probe 16 6 > $O/gather_probe.json 2>&1; cat $O/gather_probe.json
timeout 60 tools/gather_probe 16 1 > $O/gather_probe_w1.json 2>&1; cat $O/gather_probe_w1.json
run default
STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_one_stream.txt 2>&1; head -2 $O/step_one_stream.txt
run persist5120 SSLAM_LSD_PERSIST=5120
run persist4096 SSLAM_LSD_PERSIST=4096
run persist6144 SSLAM_LSD_PERSIST=6144
run prio_point STEP_POINT_PRIO=-1
run persist5120_prio SSLAM_LSD_PERSIST=5120 STEP_POINT_PRIO=-1
run mw5_persist5120 LD_PRELOAD=$V/mw5.so SSLAM_LSD_PERSIST=5120
run mw4_persist4096 LD_PRELOAD=$V/mw4.so SSLAM_LSD_PERSIST=4096
run mw4_persist4096_prio LD_PRELOAD=$V/mw4.so SSLAM_LSD_PERSIST=4096 STEP_POINT_PRIO=-1
run persist3072 SSLAM_LSD_PERSIST=3072
SSLAM_LSD_PERSIST=5120 STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_persist5120_one_stream.txt 2>&1; head -2 $O/step_persist5120_one_stream.txt
# timelines: default and the best-looking persistent form
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt; (cd $R && timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -- tools/step_check 12288 3 1 > $O/kt.log 2>&1; python tools/rocpd_summary.py $O/kt $O/kernel_trace_two_streams.txt > /dev/null; python tools/rocpd_timeline.py $O/kt $R/profiles/r05_final_kernel_trace_B12288_one_stream.txt $O/timeline_default.txt | head -60; rm -rf $O/kt)
rm -rf $O/kt; (cd $R && SSLAM_LSD_PERSIST=5120 timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -- tools/step_check 12288 3 1 > $O/kt2.log 2>&1; python tools/rocpd_timeline.py $O/kt $R/profiles/r05_final_kernel_trace_B12288_one_stream.txt $O/timeline_persist5120.txt | head -60; rm -rf $O/kt)
