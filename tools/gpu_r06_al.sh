#!/bin/bash
# Round 6, GPU call AL: the two tile kernels of the counting sort with more groups of 64 entries loaded ahead per lane (k_lsd_hist_sort: 2 / 4 / 8 / 12 / 16; k_lsd_scatter_runs: 4 / 8 / 16)
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06al; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants
one() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_$n.txt 2>&1; echo "$n: $(head -2 $O/one_$n.txt | tail -1 | grep -o 'k_lsd_hist [0-9.]*\|k_lsd_scatter [0-9.]*' | tr '\n' ' ') $(tail -1 $O/one_$n.txt | cut -c1-100)"; }
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; echo "$n: $(head -1 $O/two_$n.txt)"; }
one base X=1
for n in hs8 hs16 hs8_sr8 hs8_sr16 hs12_sr8; do one $n LD_PRELOAD=$V/$n.so; done
two base X=1
two hs8_sr8 LD_PRELOAD=$V/hs8_sr8.so
two base_b X=1
two hs8_sr8_b LD_PRELOAD=$V/hs8_sr8.so
