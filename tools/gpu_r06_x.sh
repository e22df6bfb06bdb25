#!/bin/bash
# Round 6, GPU call X: k_lbd with the gathers transposed through LDS (lbd.h, k_lbd<RPI, TB>: blocks of 16 / 8 steps; SSLAM_LBD_TB=0 = a gather per step as in rounds 1-5):
# the line suite under each form, the kernel alone, the two-stream step.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06x; mkdir -p $O
for tb in 16 8; do SSLAM_LBD_TB=$tb timeout 900 python -m pytest tests/test_lines_gpu.py tests/test_configs_gpu.py -m gpu -x -q > $O/pytest_tb$tb.txt 2>&1; tail -3 $O/pytest_tb$tb.txt; done
SSLAM_LBD_RPI=0 timeout 600 python -m pytest tests/test_lines_gpu.py -m gpu -x -q -k "oracle or golden or frames or forms" > $O/pytest_tb16_rpi0.txt 2>&1; tail -3 $O/pytest_tb16_rpi0.txt
one() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_$n.txt 2>&1; head -2 $O/one_$n.txt | cut -c1-420; tail -1 $O/one_$n.txt; }
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; head -2 $O/two_$n.txt | cut -c1-420; tail -1 $O/two_$n.txt; }
one tb16 SSLAM_LBD_TB=16
one tb8 SSLAM_LBD_TB=8
one tb0 SSLAM_LBD_TB=0
two tb16 SSLAM_LBD_TB=16
two tb8 SSLAM_LBD_TB=8
two tb0 SSLAM_LBD_TB=0
two tb16_b SSLAM_LBD_TB=16
two tb8_b SSLAM_LBD_TB=8
timeout 600 python tools/fuzz_parity.py 120 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
