#!/bin/bash
# Round 5, GPU call A (trimmed form of tools/gpu_r05_first.sh): decide the SSLAM_NFA_STREAM default and the stream / stream2 loser, prove tools/step_check against the bench
# line, and take the SQ passes the round-4 review asked for (lane occupancy of the VALU instructions of the core, the NFA stage and FAST) through the C harness (no Python under rocprofv3).
#     bash tools/build_c_harnesses.sh && gpurun --timeout 1100 -- 'bash tools/gpu_r05_a.sh'
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r05a; mkdir -p $O
[ -x tools/mix_check ] && timeout 60 tools/mix_check 2 "" "SSLAM_NFA_STREAM=1" "SSLAM_NFA_STREAM=1,SSLAM_NFA_STREAM_TICKS=0" > $O/mix_check.txt 2>&1; tail -4 $O/mix_check.txt
[ -x tools/lat_check ] && LAT_PROFILE=1 timeout 60 tools/lat_check 2 "" "SSLAM_NFA_STREAM=1" "SSLAM_NFA_STREAM=8" > $O/lat_check.txt 2>&1; cut -c1-200 $O/lat_check.txt
[ -x tools/lat_check ] && LAT_PROFILE=1 timeout 40 tools/lat_check 2 "SSLAM_NFA_STREAM=1,SSLAM_NFA_STREAM_EMIT=lds" "SSLAM_NFA_STREAM=8,SSLAM_NFA_STREAM_EMIT=lds" > $O/lat_check_lds.txt 2>&1; cut -c1-200 $O/lat_check_lds.txt
[ -x tools/mix_check ] && timeout 40 tools/mix_check 2 "SSLAM_NFA_STREAM=1,SSLAM_NFA_STREAM_EMIT=lds" > $O/mix_check_lds.txt 2>&1; tail -2 $O/mix_check_lds.txt
[ -x tools/batch_check ] && timeout 60 tools/batch_check "" "SSLAM_NFA_STREAM=1" > $O/batch_check.txt 2>&1; tail -16 $O/batch_check.txt
[ -x tools/step_check ] && STEP_PROFILE=1 timeout 120 tools/step_check 12288 5 2 > $O/step_check.txt 2>&1; cat $O/step_check.txt
[ -x tools/step_check ] && STEP_PROFILE=1 timeout 120 tools/step_check 12288 3 1 1 > $O/step_check_one_stream.txt 2>&1; cat $O/step_check_one_stream.txt
# SQ passes over one one-stream step of 3072 frames through the C harness
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
P3="SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); rm -rf $O/sq$i
  (cd $R && timeout 240 rocprofv3 --pmc $P -d $O/sq$i -- tools/step_check 3072 1 0 1 > $O/sq$i.log 2>&1; python tools/rocpd_pmc_summary.py $O/sq$i $O/sq$i.txt > /dev/null; rm -rf $O/sq$i; tail -3 $O/sq$i.log | cut -c1-200)
done
cd $R
# the experimental tests and the whole suite with the knob exported
SSLAM_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_experimental_gpu.py -x -q -m gpu > $O/pytest_experimental.txt 2>&1; echo "rc=$?" >> $O/pytest_experimental.txt; tail -4 $O/pytest_experimental.txt
SSLAM_NFA_STREAM=1 timeout 420 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_nfa_stream.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu_nfa_stream.txt; tail -4 $O/pytest_gpu_nfa_stream.txt
