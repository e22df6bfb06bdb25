#!/bin/bash
# Round 5, GPU call Q: the initial NFA count of a large rectangle under D11 = 1 stops once 8 k > n (lsd_nfa.h: nfa_pixel_count / NFA_DENSITY_MIN_N).
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r05q; mkdir -p $O
STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_one_stream.txt 2>&1; cat $O/step_one_stream.txt
STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_default.txt 2>&1; head -2 $O/step_default.txt
SSLAM_PROF_STAGES=1 SSLAM_NFA_FUSED=0 STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_staged.txt 2>&1; head -2 $O/step_staged.txt | cut -c1-700
LAT_PROFILE=1 timeout 80 tools/lat_check 2 "" "SSLAM_NFA_STREAM=0" > $O/lat_check.txt 2>&1; cut -c1-200 $O/lat_check.txt
LAT_W=1280 LAT_H=960 LAT_NF=8 LAT_LINES=400 LAT_FRAMES=tools/lat_frames_1280x960.raw LAT_EXPECTED=tools/lat_expected_1280x960.bin timeout 80 tools/lat_check 2 "" > $O/lat_check_1280.txt 2>&1; cut -c1-200 $O/lat_check_1280.txt | head -2
timeout 60 tools/mix_check 2 "" > $O/mix_check.txt 2>&1; tail -2 $O/mix_check.txt
timeout 60 tools/batch_check "" > $O/batch_check.txt 2>&1; tail -3 $O/batch_check.txt
timeout 600 python -m pytest tests/test_lines_gpu.py tests/test_edge_gpu.py tests/test_stress_gpu.py tests/test_variants_gpu.py tests/test_configs_gpu.py tests/test_nfa_stream_gpu.py tests/test_batch_gpu.py -x -q -m gpu > $O/pytest_subset.txt 2>&1; echo "rc=$?" >> $O/pytest_subset.txt; tail -4 $O/pytest_subset.txt
