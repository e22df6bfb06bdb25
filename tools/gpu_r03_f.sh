#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_batch_gpu.py tests/test_lines_gpu.py tests/test_abi_cpu.py -x -q > $O/pytest.txt 2>&1; grep -n "passed\|failed" $O/pytest.txt
timeout 120 python tools/valu_rate.py > $O/valu_rate.txt 2>&1; cat $O/valu_rate.txt
timeout 600 python tools/bench_host_batch.py 18432 0 > $O/host_batch_default.txt 2>&1; grep "sslam_frontend_batch\|equal\|Error\|error" $O/host_batch_default.txt
timeout 600 python tools/bench_host_batch.py 18432 0 1 > $O/host_batch_one_stream.txt 2>&1; grep "sslam_frontend_batch" $O/host_batch_one_stream.txt
timeout 600 python tools/bench_host_batch.py 8192 2048 > $O/host_batch_2048.txt 2>&1; grep "sslam_frontend_batch" $O/host_batch_2048.txt
SSLAM_BATCH_THREADS=1 timeout 600 python tools/bench_host_batch.py 18432 0 > $O/host_batch_1thread.txt 2>&1; grep "sslam_frontend_batch" $O/host_batch_1thread.txt
export LSD_ONLY_TOP=14
for v in product nfaint; do
  [ $v = product ] && unset SSLAM_LIB || export SSLAM_LIB=$R/structure-slam-pointline_amd/lib/variants/$v.so
  SSLAM_PROF_STAGES=1 timeout 300 python tools/lsd_only.py 12288 64 2 > $O/lsd_only_$v.txt 2>&1; tail -n 1 $O/lsd_only_$v.txt
done
