#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; grep -n "passed\|failed\|Error" $O/pytest.txt | head
timeout 300 python tools/matcher_breakdown.py > $O/matcher_breakdown.txt 2>&1; grep -v amdgpu.ids $O/matcher_breakdown.txt
timeout 300 python tools/bench_matchers.py > $O/matchers.txt 2>&1; grep -v amdgpu.ids $O/matchers.txt
timeout 600 python tools/fuzz_matchers.py 1500 > $O/fuzz_matchers.txt 2>&1; tail -n 3 $O/fuzz_matchers.txt
