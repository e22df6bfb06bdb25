#!/bin/bash
# Round 6, GPU call M: the LSD pre-blur fused into the gradient kernel (k_lsd_grad_fused) against k_blur7 + k_lsd_grad (SSLAM_LSD_FUSED=0): parity first (the step's own
# oracle comparison, then the line-path GPU tests), then the step on two streams and on one.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06m; mkdir -p $O
run() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_$n.txt 2>&1; head -2 $O/step_$n.txt | cut -c1-420; tail -1 $O/step_$n.txt | cut -c1-110; }
run fused
run two_kernels SSLAM_LSD_FUSED=0
run fused_b
STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_fused.txt 2>&1; head -2 $O/one_fused.txt | cut -c1-420; tail -1 $O/one_fused.txt | cut -c1-110
SSLAM_LSD_FUSED=0 STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_two_kernels.txt 2>&1; head -2 $O/one_two_kernels.txt | cut -c1-420
timeout 900 python -m pytest tests/test_lines_gpu.py tests/test_variants_gpu.py tests/test_batch_gpu.py tests/test_configs_gpu.py tests/test_edge_gpu.py tests/test_nfa_stream_gpu.py tests/test_shim_gpu.py -q -m gpu -x > $O/pytest_lines.txt 2>&1; echo "rc=$?" >> $O/pytest_lines.txt; tail -12 $O/pytest_lines.txt
LAT_PROFILE=1 timeout 80 tools/lat_check 2 "" > $O/lat_check.txt 2>&1; tail -5 $O/lat_check.txt | cut -c1-300
