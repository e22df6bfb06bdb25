#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
run() { echo "== $*"; env "$@" timeout 200 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-110; }
run A=1
run SSLAM_CL_WINDOW=224
run SSLAM_CL_WINDOW=224 SSLAM_CL_WGS=13
run SSLAM_CL_WINDOW=128
run SSLAM_CL_WGS=8
