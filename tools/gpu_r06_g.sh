#!/bin/bash
# Round 6, GPU call G: 16-byte loads in the five-candidate rectangle counter (both D11 forms), chunk size / core form of sslam_frontend_batch on the bench's own frame sequence,
# the line-path GPU tests + the new guest-form batch test.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06g; mkdir -p $O
run() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_$n.txt 2>&1; head -2 $O/step_$n.txt | cut -c1-420; tail -1 $O/step_$n.txt | cut -c1-110; }
run default
run nfa_variant0 STEP_NFA_VARIANT=0
STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_one_stream.txt 2>&1; head -2 $O/step_one_stream.txt | cut -c1-420
STEP_NFA_VARIANT=0 STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_one_stream_nfa_variant0.txt 2>&1; head -2 $O/step_one_stream_nfa_variant0.txt | cut -c1-420
export HOST_BATCH_BENCH_FRAMES=1
timeout 300 python tools/bench_host_batch.py 24576 0 > $O/host_default.txt 2>&1; tail -4 $O/host_default.txt | cut -c1-200
timeout 300 python tools/bench_host_batch.py 24576 8192 > $O/host_c8192.txt 2>&1; tail -4 $O/host_c8192.txt | cut -c1-200
timeout 300 python tools/bench_host_batch.py 24576 4096 > $O/host_c4096.txt 2>&1; tail -4 $O/host_c4096.txt | cut -c1-200
SSLAM_LSD_GUEST=0 timeout 300 python tools/bench_host_batch.py 24576 6144 > $O/host_noguest_c6144.txt 2>&1; tail -4 $O/host_noguest_c6144.txt | cut -c1-200
SSLAM_LSD_GUEST=0 timeout 300 python tools/bench_host_batch.py 24576 12288 > $O/host_noguest_c12288.txt 2>&1; tail -4 $O/host_noguest_c12288.txt | cut -c1-200
timeout 300 python tools/bench_host_batch.py 24576 12288 > $O/host_c12288.txt 2>&1; tail -4 $O/host_c12288.txt | cut -c1-200
unset HOST_BATCH_BENCH_FRAMES
timeout 900 python -m pytest tests/test_lines_gpu.py tests/test_variants_gpu.py tests/test_batch_gpu.py tests/test_nfa_stream_gpu.py tests/test_configs_gpu.py -q -m gpu --durations=5 > $O/pytest_lines.txt 2>&1; echo "rc=$?" >> $O/pytest_lines.txt; tail -12 $O/pytest_lines.txt
