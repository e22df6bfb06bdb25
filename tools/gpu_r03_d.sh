#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_lines_gpu.py tests/test_shim_gpu.py tests/test_group_gpu.py tests/test_match_gpu.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
export LSD_ONLY_TOP=14
for v in product nfaf64 cw3 cw3f64; do
  [ $v = product ] && unset SSLAM_LIB || export SSLAM_LIB=$R/structure-slam-pointline_amd/lib/variants/$v.so
  SSLAM_PROF_STAGES=1 timeout 300 python tools/lsd_only.py 12288 64 2 > $O/lsd_only_$v.txt 2>&1; tail -n 1 $O/lsd_only_$v.txt | tr ',' '\n' | grep "nfa_count" | tr '\n' ' '; echo
done
