#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
echo "== check"; timeout 300 python tools/cl_probe.py 64 2>&1 | grep -v amdgpu.ids | tail -3
timeout 200 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-110
timeout 600 python -m pytest tests/test_lines_gpu.py -x -q -m gpu 2>&1 | tail -2
