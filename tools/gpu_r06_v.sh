#!/bin/bash
# Round 6, GPU call V: k_resize with workgroups that walk the frames (everything a thread derives from its output columns and row -- the table entries, the byte selectors, the
# coefficient pairs -- is the same for every frame: hoisted out of the frame loop): grid sizes, alone and in the two-stream step
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06v; mkdir -p $O
for g in 0 1024 2048 4096 8192 16384 65536; do echo "== resize grid $g"; SSLAM_RESIZE_GRID_WGS=$g STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 2>&1 | head -2 | tail -1 | grep -o "k_resize [0-9.]*"; SSLAM_RESIZE_GRID_WGS=$g STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_g$g.txt 2>&1; head -2 $O/step_g$g.txt | cut -c1-420; done
