#!/bin/bash
# Round 5, GPU call G: LBD's walk without clamps for support regions that lie inside the image, LBD's Sobel on the second stream of the single-frame path; the two fuzz tests
# that GPU call F broke (their generator yields the line path's decisions now); the core's phase table (library built with -DSSLAM_LSD_CYCLES).
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r05g; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants
STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_default.txt 2>&1; cat $O/step_default.txt
STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_one_stream.txt 2>&1; cat $O/step_one_stream.txt
LAT_PROFILE=1 timeout 80 tools/lat_check 2 "" "SSLAM_NFA_STREAM=0" > $O/lat_check.txt 2>&1; cut -c1-300 $O/lat_check.txt
timeout 60 tools/mix_check 2 "" > $O/mix_check.txt 2>&1; tail -2 $O/mix_check.txt
timeout 60 tools/batch_check "" > $O/batch_check.txt 2>&1; tail -8 $O/batch_check.txt
timeout 500 python -m pytest tests/test_lines_gpu.py tests/test_edge_gpu.py tests/test_stress_gpu.py tests/test_variants_gpu.py tests/test_configs_gpu.py tests/test_nfa_stream_gpu.py tests/test_shim_gpu.py -x -q -m gpu > $O/pytest_subset.txt 2>&1; echo "rc=$?" >> $O/pytest_subset.txt; tail -6 $O/pytest_subset.txt
[ -f $V/cyc.so ] && SSLAM_LIB=$V/cyc.so timeout 300 python tools/lsd_cycles_batch.py 12288 64 > $O/core_phase_table.txt 2>&1; cat $O/core_phase_table.txt
