#!/bin/bash
# Round 5, first GPU call: decide whether SSLAM_NFA_STREAM (the NFA stage next to the cluster form of the core) becomes the default for calls of <= 64 frames.
#     bash tools/build_c_harnesses.sh                                            (CPU, once: the C harnesses and their inputs; they travel with the snapshot)
#     gpurun --timeout 1500 -- 'bash tools/gpu_r05_first.sh'
# Round 4 measured it through the C ABI alone (profiles/r04_nfa_stream_c_abi_runs.txt: every output equal to the oracle, 5.85 / 7.00 -> 5.57 / 6.72 ms per frame, calls of
# 2 .. 64 frames -10 .. -25 %) but had no GPU minutes left for the suite.  What decides: the WHOLE GPU suite with the knob exported (every cluster-form call of every
# test then takes the streaming form), the experimental tests (consumer counts, patience settings), and the stress runs.  Each step under its own timeout; outputs in gpurun_out/r05a/.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r05a; mkdir -p $O
# 1. seconds: the C harnesses (no Python) -- default and streaming form, parity + latency on this box
[ -x tools/mix_check ] && timeout 60 tools/mix_check 2 "" "SSLAM_NFA_STREAM=1" "SSLAM_NFA_STREAM=1,SSLAM_NFA_STREAM_TICKS=0" > $O/mix_check.txt 2>&1; tail -4 $O/mix_check.txt
[ -x tools/lat_check ] && LAT_PROFILE=1 timeout 60 tools/lat_check 2 "" "SSLAM_NFA_STREAM=1" "SSLAM_NFA_STREAM=8" > $O/lat_check.txt 2>&1; cut -c1-200 $O/lat_check.txt
# 1a. NEVER RUN ON A GPU BEFORE: the hand-over through LDS and a publisher wave (SSLAM_NFA_STREAM_EMIT=lds; one candidate for the streaming kernel's slower core: its next loads wait behind the sc1 stores) --
#     own timeout, parity is checked in the run; compare the k_lsd_regions scope with the two lines above
[ -x tools/lat_check ] && LAT_PROFILE=1 timeout 40 tools/lat_check 2 "SSLAM_NFA_STREAM=1,SSLAM_NFA_STREAM_EMIT=lds" "SSLAM_NFA_STREAM=8,SSLAM_NFA_STREAM_EMIT=lds" > $O/lat_check_lds.txt 2>&1; cut -c1-200 $O/lat_check_lds.txt
[ -x tools/mix_check ] && timeout 40 tools/mix_check 2 "SSLAM_NFA_STREAM=1,SSLAM_NFA_STREAM_EMIT=lds" > $O/mix_check_lds.txt 2>&1; tail -2 $O/mix_check_lds.txt
[ -x tools/batch_check ] && timeout 60 tools/batch_check "" "SSLAM_NFA_STREAM=1" > $O/batch_check.txt 2>&1; tail -16 $O/batch_check.txt
# 1b. the main wave's code is a roll of the register allocator's dice (the streaming kernel's core is 3 % slower than the default's, same source plus thirty
#     instructions): other rolls of lines.hip, built on the CPU beforehand -- tools/build_variant.sh pm0 "-mllvm -enable-post-misched=0"; ... ilp "-mllvm -amdgpu-sched-strategy=max-ilp";
#     ... il1 "-mllvm -inline-threshold=100000" -- each through the C harness (LD_PRELOAD replaces the library), default and streaming form; parity is checked in every run
for V in pm0 ilp il1; do
  [ -f structure-slam-pointline_amd/lib/variants/$V.so ] && LD_PRELOAD=$R/structure-slam-pointline_amd/lib/variants/$V.so LAT_PROFILE=1 timeout 60 tools/lat_check 2 "" "SSLAM_NFA_STREAM=1" > $O/lat_check_$V.txt 2>&1; cut -c1-200 $O/lat_check_$V.txt
done
# 1c. NEVER RUN ON A GPU BEFORE: the bench's step through the C ABI alone (tools/step_check.c) -- if its frames/s and per-frame means agree with bench.py's line (68.3 k frames/s,
#     1004.2 keypoints, 164.45 lines, 102.7 / 118.8 matches per frame on the round-4 build), kernel experiments on the headline metric cost ~20 s of GPU time from here on
[ -x tools/step_check ] && STEP_PROFILE=1 timeout 120 tools/step_check 12288 5 2 > $O/step_check.txt 2>&1; cat $O/step_check.txt
# 2. the experimental tests (they spin on device flags: tight timeout)
SSLAM_TEST_EXPERIMENTAL=1 timeout 420 python -m pytest tests/test_experimental_gpu.py -x -q -m gpu > $O/pytest_experimental.txt 2>&1; echo "rc=$?" >> $O/pytest_experimental.txt; tail -4 $O/pytest_experimental.txt
# 3. the whole suite with the knob exported, then without (the default path after the refactor of the NFA bodies into *_range forms)
SSLAM_NFA_STREAM=1 timeout 600 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_nfa_stream.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu_nfa_stream.txt; tail -4 $O/pytest_gpu_nfa_stream.txt
timeout 600 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_default.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu_default.txt; tail -4 $O/pytest_gpu_default.txt
# 4. the bench line (its latency leg and the child-process comparison `latency_experiment_nfa_stream`)
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r05a/bench.json').read().strip().splitlines()[-1])
    print(round(d['value']), d['latency']['lines_extract_hipEvent'], d.get('latency_experiment_nfa_stream'))
except Exception as e: print('bench failed', e)
PY
