#!/bin/bash
# Round 6, GPU call AJ (timing builds, results wrong on purpose): the nested NFA counter without its row loads (constants instead), with one slot per run instead of twelve, with neither:
# what the initial count's 5.2 ms consist of
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06aj; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants
for n in base nfa_noload nfa_novotes nfa_neither; do
  pre=""; [ $n != base ] && pre="LD_PRELOAD=$V/$n.so"
  env $pre SSLAM_NFA_FUSED=0 SSLAM_PROF_STAGES=1 STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/stages_$n.txt 2>&1; echo "$n separate: $(head -2 $O/stages_$n.txt | tail -1 | grep -o 'k_nfa[a-z_/0-9]* [0-9.]*' | tr '\n' ' ')"
  env $pre STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/fused_$n.txt 2>&1; echo "$n fused: $(head -2 $O/fused_$n.txt | tail -1 | grep -o 'k_nfa_all [0-9.]*')"
done
