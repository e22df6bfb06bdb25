#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
echo "== check"; timeout 300 python tools/cl_probe.py 64 2>&1 | grep -v amdgpu.ids | tail -3
run() { echo "== $*"; env "$@" timeout 200 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-110; }
run A=1
run SSLAM_CL_WINDOW=224
run SSLAM_CL_WINDOW=320 SSLAM_CL_WGS=12
