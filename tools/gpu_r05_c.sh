#!/bin/bash
# Round 5, GPU call C: LBD walk (integer rounding of the coordinates, branch-free accumulation) and FAST's branch-free suppression -- parity through the C harnesses and the
# ORB / line / fuzz tests, the step through the C harness (two streams, one stream), and one variant of the core (SSLAM_LSD_EARLY_DC: the accept chain's three broadcasts together).
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r05c; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants
STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_default.txt 2>&1; cat $O/step_default.txt
STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_one_stream.txt 2>&1; cat $O/step_one_stream.txt
[ -f $V/earlydc.so ] && LD_PRELOAD=$V/earlydc.so STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_earlydc.txt 2>&1; cat $O/step_earlydc.txt
[ -f $V/earlydc.so ] && LD_PRELOAD=$V/earlydc.so STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_earlydc_one_stream.txt 2>&1; cat $O/step_earlydc_one_stream.txt
LAT_PROFILE=1 timeout 80 tools/lat_check 2 "" > $O/lat_check.txt 2>&1; cut -c1-300 $O/lat_check.txt
timeout 60 tools/mix_check 2 "" > $O/mix_check.txt 2>&1; tail -2 $O/mix_check.txt
timeout 500 python -m pytest tests/test_orb_gpu.py tests/test_lines_gpu.py tests/test_edge_gpu.py tests/test_variants_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $O/pytest_subset.txt 2>&1; echo "rc=$?" >> $O/pytest_subset.txt; tail -6 $O/pytest_subset.txt
