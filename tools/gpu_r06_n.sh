#!/bin/bash
# Round 6, GPU call N: the counting sort with tile-sorted runs (k_lsd_hist_sort + k_lsd_scatter_runs) against k_lsd_hist + k_lsd_scatter (SSLAM_LSD_SORT_RUNS=0), and the
# straight-line form of k_lbd's whole groups: parity first, then the step on one stream and on two.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06n; mkdir -p $O
STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_runs.txt 2>&1; head -2 $O/one_runs.txt | cut -c1-420; tail -1 $O/one_runs.txt | cut -c1-110
SSLAM_LSD_SORT_RUNS=0 STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_old_sort.txt 2>&1; head -2 $O/one_old_sort.txt | cut -c1-420
run() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_$n.txt 2>&1; head -2 $O/step_$n.txt | cut -c1-420; tail -1 $O/step_$n.txt | cut -c1-110; }
run runs
run old_sort SSLAM_LSD_SORT_RUNS=0
run runs_b
timeout 900 python -m pytest tests/test_lines_gpu.py tests/test_variants_gpu.py tests/test_batch_gpu.py tests/test_configs_gpu.py tests/test_edge_gpu.py tests/test_nfa_stream_gpu.py tests/test_shim_gpu.py tests/test_stress_gpu.py -q -m gpu -x > $O/pytest_lines.txt 2>&1; echo "rc=$?" >> $O/pytest_lines.txt; tail -6 $O/pytest_lines.txt
LAT_PROFILE=1 timeout 80 tools/lat_check 2 "" > $O/lat_check.txt 2>&1; tail -3 $O/lat_check.txt | cut -c1-300
