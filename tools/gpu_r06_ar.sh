#!/bin/bash
# Round 6, GPU call AR: k_describe with its own double sin / cos on [0, 6.5] (fdlibm kernels behind a two-term reduction) instead of the library's sincos: the exhaustive
# self-test (every float of the range, after the rounding to float), parity, the kernel alone and the step (46 + 16 registers instead of 52 + 16: two waves instead of one
# in the 128 registers the guest form of the LSD core leaves)
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06ar; mkdir -p $O
timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_configs_gpu.py tests/test_edge_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
one() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_$n.txt 2>&1; echo "$n: $(head -2 $O/one_$n.txt | tail -1 | grep -o 'k_describe [0-9.]*') $(tail -1 $O/one_$n.txt | cut -c1-100)"; }
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; echo "$n: $(head -1 $O/two_$n.txt) $(head -2 $O/two_$n.txt | tail -1 | grep -o 'k_lsd_regions [0-9.]*\|k_fast_cells [0-9.]*\|k_describe [0-9.]*\|k_octree [0-9.]*' | tr '\n' ' ')"; }
one a X=1
two g4608 X=1
two g4608_b X=1
two g4864 SSLAM_LSD_PERSIST=4864
two g5120 SSLAM_LSD_PERSIST=5120
two g4608_c X=1
timeout 600 python tools/fuzz_parity.py 200 53 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
