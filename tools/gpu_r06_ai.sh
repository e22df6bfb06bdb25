#!/bin/bash
# Round 6, GPU call AI: where the NFA stage's time is -- its launches apart (SSLAM_NFA_FUSED=0, one scope per counting stage), 12 288 frames, both D11 forms
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06ai; mkdir -p $O
SSLAM_NFA_FUSED=0 SSLAM_PROF_STAGES=1 STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/stages_v1.txt 2>&1; head -2 $O/stages_v1.txt | tail -1 | tr ' ' '\n' | paste - - | grep -i "nfa" 
SSLAM_NFA_FUSED=0 SSLAM_PROF_STAGES=1 STEP_NFA_VARIANT=0 STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/stages_v0.txt 2>&1; head -2 $O/stages_v0.txt | tail -1 | tr ' ' '\n' | paste - - | grep -i "nfa"
