#!/bin/bash
# Round 5, GPU call P: the NFA stage's launch forms under nfa variant 1 (the default since this round; the forms were chosen under variant 0): one fused launch (default at
# >= 2048 frames) against the staged launches with 1 / 2 / 4 counting and evaluating waves per frame.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r05p; mkdir -p $O
run() { name=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_$name.txt 2>&1; echo "== $name: $*"; head -2 $O/step_$name.txt | cut -c1-600; tail -1 $O/step_$name.txt; }
run fused X=1
run staged11 SSLAM_NFA_FUSED=0
run staged22 SSLAM_NFA_FUSED=0 SSLAM_COUNT_WAVES=2 SSLAM_EVAL_WAVES=2
run staged41 SSLAM_NFA_FUSED=0 SSLAM_COUNT_WAVES=4 SSLAM_EVAL_WAVES=1
run staged44 SSLAM_NFA_FUSED=0 SSLAM_COUNT_WAVES=4 SSLAM_EVAL_WAVES=4
SSLAM_PROF_STAGES=1 SSLAM_NFA_FUSED=0 STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_staged11_stages.txt 2>&1; head -2 $O/step_staged11_stages.txt | cut -c1-700
