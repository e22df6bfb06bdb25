#!/bin/bash
# Round 6, GPU call AA: the NFA stage with the rectangles' geometry in scalar registers (k_nfa_all 128 VGPRs + 16 B scratch -> 92, k_nfa_count 122 -> 79) and item-list chunks of
# 768 / 640 rectangles (7.5 / 6.3 KB of LDS: five waves per SIMD) against 1 024 (four waves); both D11 forms
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06aa; mkdir -p $O
timeout 900 python -m pytest tests/test_lines_gpu.py tests/test_variants_gpu.py tests/test_nfa_stream_gpu.py tests/test_configs_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
one() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 3 1 1 > $O/one_$n.txt 2>&1; head -2 $O/one_$n.txt | tail -1 | grep -o "k_nfa_all [0-9.]*"; tail -1 $O/one_$n.txt; }
two() { n=$1; shift; env "$@" STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/two_$n.txt 2>&1; head -2 $O/two_$n.txt | cut -c1-420; tail -1 $O/two_$n.txt; }
one ch768 SSLAM_NFA_CH=768
one ch1024 SSLAM_NFA_CH=1024
one ch640 SSLAM_NFA_CH=640
one v0_ch768 SSLAM_NFA_CH=768 SSLAM_LSD_NFA_VARIANT=0
one v0_ch1024 SSLAM_NFA_CH=1024 SSLAM_LSD_NFA_VARIANT=0
two ch768 SSLAM_NFA_CH=768
two ch1024 SSLAM_NFA_CH=1024
two ch640 SSLAM_NFA_CH=640
two ch768_b SSLAM_NFA_CH=768
two ch1024_b SSLAM_NFA_CH=1024
timeout 300 tools/lat_check 2 > $O/lat.txt 2>&1; tail -4 $O/lat.txt
timeout 600 python tools/fuzz_parity.py 150 11 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
