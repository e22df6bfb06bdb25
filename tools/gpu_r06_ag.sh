#!/bin/bash
# Round 6, GPU call AG: occupancy of the two register-heavy streaming kernels of the line branch: k_blur_sobel (140 VGPRs = three waves per SIMD) capped at 128 / 96 registers
# (44 / 152 bytes of scratch), k_lsd_grad_fused (128 VGPRs = four waves) capped at 96 (184 bytes of scratch); alone (SSLAM_LBD_SOBEL_MAIN=1: on the main stream) and in the step
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06ag; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants
one() { n=$1; shift; env "$@" SSLAM_LBD_SOBEL_MAIN=1 STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_$n.txt 2>&1; echo "$n: $(head -2 $O/one_$n.txt | tail -1 | grep -o 'k_blur_sobel [0-9.]*\|k_lsd_grad [0-9.]*' | tr '\n' ' ') $(tail -1 $O/one_$n.txt | cut -c1-100)"; }
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; echo "$n: $(head -1 $O/two_$n.txt) $(head -2 $O/two_$n.txt | tail -1 | grep -o 'k_blur_sobel [0-9.]*\|k_lsd_grad [0-9.]*' | tr '\n' ' ')"; }
one base X=1
one sobel_mw4 LD_PRELOAD=$V/sobel_mw4.so
one sobel_mw5 LD_PRELOAD=$V/sobel_mw5.so
one grad_mw5 LD_PRELOAD=$V/grad_mw5.so
two base X=1
two sobel_mw4 LD_PRELOAD=$V/sobel_mw4.so
two sobel_mw5 LD_PRELOAD=$V/sobel_mw5.so
two grad_mw5 LD_PRELOAD=$V/grad_mw5.so
two base_b X=1
two sobel_mw4_b LD_PRELOAD=$V/sobel_mw4.so
