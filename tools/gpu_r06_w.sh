#!/bin/bash
# Round 6, GPU call W: (1) k_lbd's walk with one conversion + one median per coordinate and packed operand selection (lbd.h, k_lbd<true>; SSLAM_LBD_RPI=0 = the previous form):
# the exhaustive rounding self-test, the line suite, the kernel alone and the two-stream step; (2) k_fast_cells' first pass without v_alignbyte / with folded max-min;
# (3) how k_nfa_all scales with the batch (4 096 / 8 192 / 12 288 frames: the tail of three rounds of one-wave workgroups) and with two / four waves per frame.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06w; mkdir -p $O
timeout 900 python -m pytest tests/test_lines_gpu.py tests/test_orb_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
one() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_$n.txt 2>&1; head -2 $O/one_$n.txt | cut -c1-420; tail -1 $O/one_$n.txt; }
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; head -2 $O/two_$n.txt | cut -c1-420; tail -1 $O/two_$n.txt; }
one rpi1 SSLAM_LBD_RPI=1
one rpi0 SSLAM_LBD_RPI=0
two rpi1 SSLAM_LBD_RPI=1
two rpi0 SSLAM_LBD_RPI=0
two rpi1_b SSLAM_LBD_RPI=1
two rpi0_b SSLAM_LBD_RPI=0
for n in 4096 8192; do STEP_PROFILE=1 timeout 100 tools/step_check $n 3 1 1 > $O/one_n$n.txt 2>&1; head -2 $O/one_n$n.txt | cut -c1-420; done
one nfa_w2 SSLAM_NFA_WAVES=2
one nfa_w4 SSLAM_NFA_WAVES=4
two nfa_w2 SSLAM_NFA_WAVES=2
