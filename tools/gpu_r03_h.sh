#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03h; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_match_gpu.py tests/test_shim_gpu.py tests/test_batch_gpu.py -x -q > $O/pytest.txt 2>&1; grep -n "passed\|failed\|Error\|assert" $O/pytest.txt | head
timeout 300 python tools/matcher_breakdown.py > $O/matcher_breakdown.txt 2>&1; grep -v amdgpu.ids $O/matcher_breakdown.txt
timeout 600 python tools/fuzz_matchers.py 1500 > $O/fuzz_matchers.txt 2>&1; tail -n 3 $O/fuzz_matchers.txt
