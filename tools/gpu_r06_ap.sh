#!/bin/bash
# Round 6, GPU call AP: the guest form's persistent grid on the final tree (15 / 16 / 17 / 18 workgroups per compute unit; 16 = 4 096 since call D)
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06ap; mkdir -p $O
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; echo "$n: $(head -1 $O/two_$n.txt) $(head -2 $O/two_$n.txt | tail -1 | grep -o 'k_lsd_regions [0-9.]*\|k_fast_cells [0-9.]*' | tr '\n' ' ')"; }
two g4096 X=1
two g3840 SSLAM_LSD_PERSIST=3840
two g4352 SSLAM_LSD_PERSIST=4352
two g4608 SSLAM_LSD_PERSIST=4608
two g4096_b X=1
two g4352_b SSLAM_LSD_PERSIST=4352
