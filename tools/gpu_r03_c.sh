#!/bin/bash
# Round-3 call C: row-major planes; rectangle counter integer vs fp64; 7 / 8 waves per SIMD for the sequential core
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for fl in mw thr; do SSLAM_LSD_FLAVOUR=$fl timeout 400 python tools/fuzz_parity.py 150 $((51 + ${#fl})) > $O/fuzz_$fl.txt 2>&1; tail -n 1 $O/fuzz_$fl.txt; done
export LSD_ONLY_TOP=14
for v in product nfaf64 r02; do
  [ $v = product ] && unset SSLAM_LIB || export SSLAM_LIB=$R/structure-slam-pointline_amd/lib/variants/$v.so
  SSLAM_PROF_STAGES=1 timeout 300 python tools/lsd_only.py 12288 64 2 > $O/lsd_only_$v.txt 2>&1; tail -n 1 $O/lsd_only_$v.txt
done
export LSD_ONLY_TOP=2
SSLAM_LIB=$R/structure-slam-pointline_amd/lib/variants/w7.so timeout 300 python tools/lsd_only.py 14336 64 2 > $O/lsd_only_w7.txt 2>&1; tail -n 1 $O/lsd_only_w7.txt
SSLAM_LIB=$R/structure-slam-pointline_amd/lib/variants/w8.so timeout 300 python tools/lsd_only.py 16384 64 2 > $O/lsd_only_w8.txt 2>&1; tail -n 1 $O/lsd_only_w8.txt
unset SSLAM_LIB
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_two_streams.json 2> $O/bench.err; cut -c1-300 $O/bench_two_streams.json
