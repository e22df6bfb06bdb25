#!/bin/bash
# Round 6, GPU call AM: source rows requested ahead per lane in k_blur_sobel (1 / 2 / 3 / 4 / 6; 2 since round 4), alone (SSLAM_LBD_SOBEL_MAIN=1) and in the step
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06am; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants
one() { n=$1; shift; env "$@" SSLAM_LBD_SOBEL_MAIN=1 STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_$n.txt 2>&1; echo "$n: $(head -2 $O/one_$n.txt | tail -1 | grep -o 'k_blur_sobel [0-9.]*') $(tail -1 $O/one_$n.txt | cut -c1-100)"; }
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; echo "$n: $(head -1 $O/two_$n.txt) $(head -2 $O/two_$n.txt | tail -1 | grep -o 'k_blur_sobel [0-9.]*')"; }
one base X=1
for d in 1 3 4 6; do one ras$d LD_PRELOAD=$V/ras$d.so; done
two base X=1
two ras4 LD_PRELOAD=$V/ras4.so
