#!/bin/bash
# Round 5, GPU call B: the whole GPU suite on the tree with the selectable line decisions (D11 / D12 defaults flipped), the streaming NFA default, level 0 in place and the
# pruned library; then the step through the C harness: default, with the level-0 copy forced, and under the old D11; single-frame latency both NFA placements.
#     bash tools/build_c_harnesses.sh && gpurun --timeout 1000 -- 'bash tools/gpu_r05_b.sh'
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r05b; mkdir -p $O
STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_default.txt 2>&1; cat $O/step_default.txt
SSLAM_ORB_COPY_LEVEL0=1 STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_copy_level0.txt 2>&1; cat $O/step_copy_level0.txt
STEP_NFA_VARIANT=0 STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_nfa_variant0.txt 2>&1; cat $O/step_nfa_variant0.txt
STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_one_stream.txt 2>&1; cat $O/step_one_stream.txt
LAT_PROFILE=1 timeout 80 tools/lat_check 2 "" "SSLAM_NFA_STREAM=0" > $O/lat_check.txt 2>&1; cut -c1-300 $O/lat_check.txt
timeout 60 tools/mix_check 2 "" "SSLAM_NFA_STREAM=0" > $O/mix_check.txt 2>&1; tail -3 $O/mix_check.txt
timeout 700 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt; tail -15 $O/pytest_gpu.txt
