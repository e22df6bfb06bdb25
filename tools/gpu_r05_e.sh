#!/bin/bash
# Round 5, GPU call E: wave-slot budgets.  The core is compiled for six waves per SIMD (80 VGPRs, 12 spilled) and then owns 480 of a SIMD's 512 registers, so the point branch's
# waves only run where a core wave has left; five (96 VGPRs) or four leave room beside it.  The same question for the NFA stage (four).  Each variant: the step on two streams
# and on one (parity is checked in every run).
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r05e; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants
STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_default.txt 2>&1; head -3 $O/step_default.txt; tail -1 $O/step_default.txt
STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_default_one_stream.txt 2>&1; head -2 $O/step_default_one_stream.txt
for v in minw5 minw4 minw7 nfa3 nfa5 nfa6; do
  [ -f $V/$v.so ] || continue
  LD_PRELOAD=$V/$v.so STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_$v.txt 2>&1; head -3 $O/step_$v.txt; tail -1 $O/step_$v.txt
  LD_PRELOAD=$V/$v.so STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_${v}_one_stream.txt 2>&1; head -2 $O/step_${v}_one_stream.txt
done
