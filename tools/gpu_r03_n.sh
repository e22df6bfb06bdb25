#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_lines_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python tools/latency_probe.py --check 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tools/fuzz_parity.py 300 4242 2>&1 | tail -2
