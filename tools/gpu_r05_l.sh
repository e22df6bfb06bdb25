#!/bin/bash
# Round 5, GPU call L: the NFA counter's runs as three 16-byte loads instead of twelve dword loads (A/B against -DSSLAM_NFA_X4=0 on the same box), and what the address units
# (TA) and the vector L1 (TCP) are doing during one step: busy / stalled cycles per kernel, three counter passes of their own.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r05l; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants
STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_one_stream.txt 2>&1; cat $O/step_one_stream.txt
LD_PRELOAD=$V/nfax4off.so STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_one_stream_x4off.txt 2>&1; head -2 $O/step_one_stream_x4off.txt
STEP_NFA_VARIANT=0 STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_one_stream_v0.txt 2>&1; head -2 $O/step_one_stream_v0.txt; tail -1 $O/step_one_stream_v0.txt
LD_PRELOAD=$V/nfax4off.so STEP_NFA_VARIANT=0 STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_one_stream_v0_x4off.txt 2>&1; head -2 $O/step_one_stream_v0_x4off.txt
STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_default.txt 2>&1; head -2 $O/step_default.txt
LAT_PROFILE=1 timeout 80 tools/lat_check 2 "" "SSLAM_NFA_STREAM=0" > $O/lat_check.txt 2>&1; cut -c1-200 $O/lat_check.txt
cd /tmp; export TMPDIR=/tmp
i=0
for P in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  (cd $R && timeout 200 rocprofv3 --pmc $P -d $O/ta$i -- tools/step_check 3072 1 0 1 > $O/ta$i.log 2>&1; python tools/rocpd_pmc_summary.py $O/ta$i $O/ta$i.txt > /dev/null; rm -rf $O/ta$i; tail -3 $O/ta$i.log; head -5 $O/ta$i.txt)
done
cd $R
timeout 600 python -m pytest tests/test_lines_gpu.py tests/test_edge_gpu.py tests/test_stress_gpu.py tests/test_variants_gpu.py tests/test_configs_gpu.py tests/test_nfa_stream_gpu.py tests/test_batch_gpu.py -x -q -m gpu > $O/pytest_subset.txt 2>&1; echo "rc=$?" >> $O/pytest_subset.txt; tail -4 $O/pytest_subset.txt
