#!/bin/bash
# Round-3 call B: integer rectangle counter (single-window loops), box-filtered seed re-gather; A/B against variants
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_lines_gpu.py tests/test_batch_gpu.py tests/test_configs_gpu.py tests/test_edge_gpu.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for fl in mw thr; do SSLAM_LSD_FLAVOUR=$fl timeout 400 python tools/fuzz_parity.py 150 $((41 + ${#fl})) > $O/fuzz_$fl.txt 2>&1; tail -n 2 $O/fuzz_$fl.txt; done
export LSD_ONLY_TOP=12
for v in product nobox r02; do
  [ $v = product ] && unset SSLAM_LIB || export SSLAM_LIB=$R/structure-slam-pointline_amd/lib/variants/$v.so
  SSLAM_PROF_STAGES=1 timeout 300 python tools/lsd_only.py 12288 64 2 > $O/lsd_only_$v.txt 2>&1; tail -n 1 $O/lsd_only_$v.txt
done
unset SSLAM_LIB
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_two_streams.json 2> $O/bench.err; cut -c1-300 $O/bench_two_streams.json
