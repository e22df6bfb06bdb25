#!/bin/bash
# Round 5, first GPU call: the experiment that round 4 left in the tree behind a knob with only its single-frame path run on hardware, each step under its own timeout:
#     gpurun --timeout 1500 -- 'bash tools/gpu_r05_first.sh'
# 1. the default path once (the refactor of the NFA bodies into *_range forms moved a handful of instructions in k_nfa_count / k_nfa_all; k_lsd_regions_cl is
#    byte-identical to the round-4 build) -- the single-frame and line suites, then the latency leg as the baseline of this box;
# 2. SSLAM_NFA_STREAM=1 (the NFA stage next to the cluster form of the core; 640x480 single frames measured in round 4: profiles/r04_nfa_stream_c_abi_runs.txt): the experimental tests (spin on device flags -> tight timeout),
#    then the latency leg with 16 / 8 / 32 consumer waves, with --check (24 frames against the oracle).
# Outputs: gpurun_out/r05a/.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r05a; mkdir -p $O
timeout 600 python -m pytest tests/test_lines_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $O/pytest_default.txt 2>&1; tail -3 $O/pytest_default.txt
timeout 300 python tools/latency_probe.py --check > $O/latency_default.txt 2>&1; tail -2 $O/latency_default.txt
SSLAM_TEST_EXPERIMENTAL=1 timeout 420 python -m pytest tests/test_experimental_gpu.py -x -q -m gpu > $O/pytest_nfa_stream.txt 2>&1; echo "rc=$?" >> $O/pytest_nfa_stream.txt; tail -5 $O/pytest_nfa_stream.txt
if grep -q "rc=0" $O/pytest_nfa_stream.txt; then
  for W in 1 8 32; do
    SSLAM_NFA_STREAM=$W timeout 300 python tools/latency_probe.py --check > $O/latency_nfa_stream_$W.txt 2>&1; tail -2 $O/latency_nfa_stream_$W.txt
  done
  cd /tmp && export TMPDIR=/tmp
  SSLAM_NFA_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_stream -- python $R/tools/latency_probe.py > $O/prof_stream.log 2>&1
  cd $R; python tools/rocpd_summary.py $O/prof_stream $O/kernel_trace_nfa_stream.txt > /dev/null; rm -rf $O/prof_stream; head -12 $O/kernel_trace_nfa_stream.txt
fi
