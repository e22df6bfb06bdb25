#!/bin/bash
# Round 6, GPU call AE: the direction of an output line ((float)cos / sin of the KeyLine angle in double) evaluated once per line by k_keylines instead of by all 64 lanes of the
# line's wave in k_lbd (200 of a line's ~2 100 vector instructions); round 5 had measured this slower while k_lbd ran at the vector L1's line rate
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06ae; mkdir -p $O
timeout 900 python -m pytest tests/test_lines_gpu.py tests/test_configs_gpu.py tests/test_variants_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
one() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_$n.txt 2>&1; head -2 $O/one_$n.txt | tail -1 | grep -o "k_keylines [0-9.]* \|k_lbd [0-9.]*" | tr '\n' ' '; tail -1 $O/one_$n.txt; }
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; head -2 $O/two_$n.txt | cut -c1-420; tail -1 $O/two_$n.txt; }
one a X=1
one b X=1
two a X=1
two b X=1
two c X=1
timeout 600 python tools/fuzz_parity.py 150 31 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
