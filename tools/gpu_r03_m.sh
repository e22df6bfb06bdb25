#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
V=$R/structure-slam-pointline_amd/lib/variants/clrelax.so
echo "== relaxed"; SSLAM_LIB=$V timeout 300 python tools/cl_probe.py 64 2>&1 | grep -v amdgpu.ids | tail -2
run() { echo "== $*"; env "$@" timeout 200 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-110; }
run A=1
run SSLAM_LIB=$V
