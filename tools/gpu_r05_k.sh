#!/bin/bash
# Round 5, GPU call K: the LBD diet (band sums over 21 lane-uniform steps instead of a per-row division + table gather, packed row start, rounding without the sign term).
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r05k; mkdir -p $O
STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_one_stream.txt 2>&1; cat $O/step_one_stream.txt
STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_default.txt 2>&1; cat $O/step_default.txt
LAT_PROFILE=1 timeout 80 tools/lat_check 2 "" > $O/lat_check.txt 2>&1; cut -c1-300 $O/lat_check.txt
timeout 60 tools/mix_check 2 "" > $O/mix_check.txt 2>&1; tail -2 $O/mix_check.txt
timeout 600 python -m pytest tests/test_lines_gpu.py tests/test_edge_gpu.py tests/test_stress_gpu.py tests/test_variants_gpu.py tests/test_configs_gpu.py tests/test_batch_gpu.py -x -q -m gpu > $O/pytest_subset.txt 2>&1; echo "rc=$?" >> $O/pytest_subset.txt; tail -6 $O/pytest_subset.txt
