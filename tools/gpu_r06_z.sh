#!/bin/bash
# Round 6, GPU call Z: k_resize with two frames per trip of the frame walk (twelve loads in flight before either frame is computed), on the tree with k_lbd's transposed gathers
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06z; mkdir -p $O
timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_configs_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
one() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_$n.txt 2>&1; head -2 $O/one_$n.txt | cut -c1-420; tail -1 $O/one_$n.txt; }
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; head -2 $O/two_$n.txt | cut -c1-420; tail -1 $O/two_$n.txt; }
one a X=1
one g16k SSLAM_RESIZE_GRID_WGS=16384
one g4k SSLAM_RESIZE_GRID_WGS=4096
two a X=1
two b X=1
two g16k SSLAM_RESIZE_GRID_WGS=16384
two g4k SSLAM_RESIZE_GRID_WGS=4096
two c X=1
