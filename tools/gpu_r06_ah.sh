#!/bin/bash
# Round 6, GPU call AH: the NFA counters without per-vote masks (a slot that holds no defined pixel of the lane's run gets the distance +inf / an out-of-range column ONCE;
# row membership in the five-candidate counter as one unsigned compare): 13 + 27 -> 14 + 15 vector + scalar instructions per slot in the nested counter; both D11 forms
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06ah; mkdir -p $O
timeout 900 python -m pytest tests/test_lines_gpu.py tests/test_variants_gpu.py tests/test_nfa_stream_gpu.py tests/test_configs_gpu.py tests/test_edge_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
one() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_$n.txt 2>&1; echo "$n: $(head -2 $O/one_$n.txt | tail -1 | grep -o 'k_nfa_all [0-9.]*') $(tail -1 $O/one_$n.txt | cut -c1-100)"; }
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; echo "$n: $(head -1 $O/two_$n.txt) $(head -2 $O/two_$n.txt | tail -1 | grep -o 'k_nfa_all [0-9.]*')"; }
one a X=1
one b X=1
one v0 STEP_NFA_VARIANT=0
two a X=1
two b X=1
two c X=1
two v0 STEP_NFA_VARIANT=0
timeout 300 tools/lat_check 2 > $O/lat.txt 2>&1; tail -3 $O/lat.txt
timeout 600 python tools/fuzz_parity.py 300 41 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
