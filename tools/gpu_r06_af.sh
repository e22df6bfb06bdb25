#!/bin/bash
# Round 6, GPU call AF: k_lbd's transposed blocks with ONE LDS array (a block's gathered dwords take the place of its byte offsets; no gathers in flight across blocks -- eight
# waves per SIMD cover that): blocks of 16 steps (a gather instruction covers up to 16 steps x 4 rows) in the LDS of the 8-step form; against 8 steps in the same form
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06af; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants
timeout 900 python -m pytest tests/test_lines_gpu.py tests/test_configs_gpu.py tests/test_variants_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
one() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_$n.txt 2>&1; head -2 $O/one_$n.txt | tail -1 | grep -o "k_lbd [0-9.]*" | tr '\n' ' '; tail -1 $O/one_$n.txt; }
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; head -2 $O/two_$n.txt | cut -c1-420; tail -1 $O/two_$n.txt; }
one tb16 X=1
one tb8 LD_PRELOAD=$V/lbd_tb8.so
one tb16_b X=1
two tb16 X=1
two tb8 LD_PRELOAD=$V/lbd_tb8.so
two tb16_b X=1
two tb8_b LD_PRELOAD=$V/lbd_tb8.so
timeout 600 python tools/fuzz_parity.py 150 37 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
