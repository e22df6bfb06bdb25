#!/bin/bash
# Round 6, GPU call AQ: k_fast_cells at 29 VGPRs (buffer loads in the staging loop: one offset register per load; the score network in place, the ring re-read for the rare second
# polarity) instead of 51: four of its waves instead of two fit into the 128 registers the guest form of the LSD core leaves per SIMD.  Parity, the kernel alone, the step under
# persistent grids of 16 .. 20 core workgroups per compute unit
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06aq; mkdir -p $O
timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_configs_gpu.py tests/test_edge_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
one() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_$n.txt 2>&1; echo "$n: $(head -2 $O/one_$n.txt | tail -1 | grep -o 'k_fast_cells [0-9.]*') $(tail -1 $O/one_$n.txt | cut -c1-100)"; }
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; echo "$n: $(head -1 $O/two_$n.txt) $(head -2 $O/two_$n.txt | tail -1 | grep -o 'k_lsd_regions [0-9.]*\|k_fast_cells [0-9.]*\|k_describe [0-9.]*\|k_octree [0-9.]*' | tr '\n' ' ')"; }
one a X=1
two g4096 X=1
two g4352 SSLAM_LSD_PERSIST=4352
two g4608 SSLAM_LSD_PERSIST=4608
two g4864 SSLAM_LSD_PERSIST=4864
two g5120 SSLAM_LSD_PERSIST=5120
two g4096_b X=1
two g4608_b SSLAM_LSD_PERSIST=4608
timeout 600 python tools/fuzz_parity.py 200 51 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
