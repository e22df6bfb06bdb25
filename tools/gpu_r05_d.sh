#!/bin/bash
# Round 5, GPU call D: k_resize over host-built group records, the NFA stage's two-pass counting (initial evaluation on its own under D11's default), the accept chain's three
# broadcasts together as the default -- parity (harnesses + ORB / line / edge / variant / config / match tests), the step, one variant (the one-pass NFA chain) for A/B.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r05d; mkdir -p $O
V=$R/structure-slam-pointline_amd/lib/variants
STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/step_default.txt 2>&1; cat $O/step_default.txt
STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_one_stream.txt 2>&1; cat $O/step_one_stream.txt
[ -f $V/onepass.so ] && LD_PRELOAD=$V/onepass.so STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_onepass_one_stream.txt 2>&1; cat $O/step_onepass_one_stream.txt
STEP_NFA_VARIANT=0 STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/step_nfa_variant0_one_stream.txt 2>&1; cat $O/step_nfa_variant0_one_stream.txt
LAT_PROFILE=1 timeout 80 tools/lat_check 2 "" "SSLAM_NFA_STREAM=0" > $O/lat_check.txt 2>&1; cut -c1-300 $O/lat_check.txt
timeout 60 tools/mix_check 2 "" "SSLAM_NFA_STREAM=0" > $O/mix_check.txt 2>&1; tail -3 $O/mix_check.txt
timeout 60 tools/batch_check "" > $O/batch_check.txt 2>&1; tail -8 $O/batch_check.txt
timeout 600 python -m pytest tests/test_orb_gpu.py tests/test_lines_gpu.py tests/test_edge_gpu.py tests/test_variants_gpu.py tests/test_configs_gpu.py tests/test_nfa_stream_gpu.py tests/test_pin_gpu.py tests/test_batch_gpu.py -x -q -m gpu > $O/pytest_subset.txt 2>&1; echo "rc=$?" >> $O/pytest_subset.txt; tail -6 $O/pytest_subset.txt
