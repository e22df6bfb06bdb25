#!/bin/bash
# Round 6, GPU call Y: k_lbd's transposed gathers, second pass (no workgroup barrier between the LDS phases -- the gathers of the next block stay in flight over the loop edge --,
# the instruction's LDS offsets computed once per line, packed row-start and normalisation sums): blocks of 8 (default) / 4 / 16 steps against a gather per step
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06y; mkdir -p $O
timeout 900 python -m pytest tests/test_lines_gpu.py tests/test_configs_gpu.py tests/test_variants_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
one() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_$n.txt 2>&1; head -2 $O/one_$n.txt | tail -1 | grep -o "k_lbd [0-9.]*"; tail -1 $O/one_$n.txt; }
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; head -2 $O/two_$n.txt | cut -c1-420; tail -1 $O/two_$n.txt; }
one tb8 SSLAM_LBD_TB=8
one tb4 SSLAM_LBD_TB=4
one tb16 SSLAM_LBD_TB=16
one tb0 SSLAM_LBD_TB=0
two tb8 SSLAM_LBD_TB=8
two tb4 SSLAM_LBD_TB=4
two tb0 SSLAM_LBD_TB=0
two tb8_b SSLAM_LBD_TB=8
two tb4_b SSLAM_LBD_TB=4
timeout 600 python tools/fuzz_parity.py 200 7 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
