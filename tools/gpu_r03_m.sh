#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
V=$R/structure-slam-pointline_amd/lib/variants
echo "== stg64 ring2"; SSLAM_LIB=$V/clstg64.so timeout 300 python tools/cl_probe.py 64 2>&1 | grep -v amdgpu.ids | tail -2
run() { echo "== $*"; env "$@" timeout 200 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-110; }
run A=1
run SSLAM_LIB=$V/clstg64.so
run SSLAM_LIB=$V/clring4.so
