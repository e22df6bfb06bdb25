#!/bin/bash
# Round 6, GPU call AK: k_lsd_hist_sort bins |g|^2 in fp32 where fp32 decides (floor(sqrt32(s) * c) unless within 1e-3 of an integer: then the fp64 expression for the lanes
# concerned): the exhaustive self-test, the line suite, the kernel alone and in the step
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06ak; mkdir -p $O
timeout 900 python -m pytest tests/test_lines_gpu.py tests/test_variants_gpu.py tests/test_configs_gpu.py tests/test_edge_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
one() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_$n.txt 2>&1; echo "$n: $(head -2 $O/one_$n.txt | tail -1 | grep -o 'k_lsd_hist [0-9.]*') $(tail -1 $O/one_$n.txt | cut -c1-100)"; }
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; echo "$n: $(head -1 $O/two_$n.txt) $(head -2 $O/two_$n.txt | tail -1 | grep -o 'k_lsd_hist [0-9.]*')"; }
one a X=1
one b X=1
two a X=1
two b X=1
two c X=1
timeout 600 python tools/fuzz_parity.py 300 43 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
