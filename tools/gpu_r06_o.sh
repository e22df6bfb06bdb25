#!/bin/bash
# Round 6, GPU call O: tile sizes of the counting sort now that the scatter works on tile-sorted runs (longer tiles = longer runs, fewer histograms; shorter = more parallelism).
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06o; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants
for v in default tile4k tile16k tile32k; do p=""; [ $v != default ] && p="LD_PRELOAD=$V/$v.so"; env $p STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_$v.txt 2>&1; head -2 $O/one_$v.txt | cut -c1-420; tail -1 $O/one_$v.txt | cut -c1-110; done
for v in default tile16k tile32k; do p=""; [ $v != default ] && p="LD_PRELOAD=$V/$v.so"; env $p STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_$v.txt 2>&1; head -2 $O/step_$v.txt | cut -c1-420; done
