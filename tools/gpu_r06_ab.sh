#!/bin/bash
# Round 6, GPU call AB: k_nfa_all at six waves per SIMD (80 VGPRs: 32 - 48 bytes of scratch; chunks of 640 rectangles) and with four-term blocks of the binomial tail
# (88 VGPRs), against five waves (92 VGPRs, chunks of 768: call AA); both D11 forms (STEP_NFA_VARIANT=0: timing only)
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06ab; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants
one() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_$n.txt 2>&1; echo "$n: $(head -2 $O/one_$n.txt | tail -1 | grep -o 'k_nfa_all [0-9.]*') $(tail -1 $O/one_$n.txt | cut -c1-90)"; }
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; echo "$n: $(head -1 $O/one_$n.txt | cut -c40-100) $(head -2 $O/two_$n.txt | tail -1 | grep -o 'k_nfa_all [0-9.]*')"; head -1 $O/two_$n.txt; }
one base X=1
one mw6_640 LD_PRELOAD=$V/nfa_mw6.so SSLAM_NFA_CH=640
one tbk4 LD_PRELOAD=$V/nfa_tbk4.so
one tbk4_mw6_640 LD_PRELOAD=$V/nfa_tbk4_mw6.so SSLAM_NFA_CH=640
one v0_base STEP_NFA_VARIANT=0
one v0_ch1024 STEP_NFA_VARIANT=0 SSLAM_NFA_CH=1024
one v0_mw6_640 STEP_NFA_VARIANT=0 LD_PRELOAD=$V/nfa_mw6.so SSLAM_NFA_CH=640
one v0_tbk4 STEP_NFA_VARIANT=0 LD_PRELOAD=$V/nfa_tbk4.so
one v0_tbk4_mw6_640 STEP_NFA_VARIANT=0 LD_PRELOAD=$V/nfa_tbk4_mw6.so SSLAM_NFA_CH=640
two base X=1
two mw6_640 LD_PRELOAD=$V/nfa_mw6.so SSLAM_NFA_CH=640
two tbk4_mw6_640 LD_PRELOAD=$V/nfa_tbk4_mw6.so SSLAM_NFA_CH=640
two base_b X=1
two mw6_640_b LD_PRELOAD=$V/nfa_mw6.so SSLAM_NFA_CH=640
