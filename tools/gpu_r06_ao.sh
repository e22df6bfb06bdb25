#!/bin/bash
# Round 6, GPU call AO: the nested NFA counter with the NEXT rectangle's first run requested before the current rectangle's slots are evaluated (k_nfa_all 92 -> 112 VGPRs: four waves
# per SIMD instead of five); both D11 forms; parity
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06ao; mkdir -p $O
timeout 900 python -m pytest tests/test_lines_gpu.py tests/test_variants_gpu.py tests/test_nfa_stream_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
one() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_$n.txt 2>&1; echo "$n: $(head -2 $O/one_$n.txt | tail -1 | grep -o 'k_nfa_all [0-9.]*') $(tail -1 $O/one_$n.txt | cut -c1-100)"; }
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; echo "$n: $(head -1 $O/two_$n.txt) $(head -2 $O/two_$n.txt | tail -1 | grep -o 'k_nfa_all [0-9.]*')"; }
one a X=1
one b X=1
one v0 STEP_NFA_VARIANT=0
SSLAM_NFA_FUSED=0 SSLAM_PROF_STAGES=1 STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/stages.txt 2>&1; head -2 $O/stages.txt | tail -1 | grep -o 'k_nfa[a-z_/0-9]* [0-9.]*' | tr '\n' ' '; echo
two a X=1
two b X=1
timeout 300 tools/lat_check 2 > $O/lat.txt 2>&1; tail -2 $O/lat.txt
