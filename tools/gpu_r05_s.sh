#!/bin/bash
# Round 5, GPU call S: instruction-issue priority (s_setprio) for the line stream's long kernels -- the line stream is the step's critical path (it ends ~20 ms after the point
# stream), and its kernels lose ~35 ms to sharing the SIMDs with the point branch.  Variant libraries built out of tree: s_setprio 2 in the core; in the core, the NFA stage and LBD; 3 in all three.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r05s; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants
STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_base.txt 2>&1; head -2 $O/step_base.txt | cut -c1-500; tail -1 $O/step_base.txt
for v in prio_core prio_all prio3_all; do
  LD_PRELOAD=$V/$v.so STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_$v.txt 2>&1; echo "== $v"; head -2 $O/step_$v.txt | cut -c1-500; tail -1 $O/step_$v.txt
done
STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_base2.txt 2>&1; head -1 $O/step_base2.txt
