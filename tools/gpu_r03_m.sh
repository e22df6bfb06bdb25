#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
echo "== check"; SSLAM_LSD_CLUSTER=1 timeout 300 python tools/cl_probe.py 64 2>&1 | grep -v amdgpu.ids | tail -4
run() { echo "== $*"; env SSLAM_LSD_CLUSTER=1 "$@" timeout 200 python tools/cl_probe.py 32 --nocheck --cycles 2>&1 | grep -v amdgpu.ids | tail -3; }
V=$R/structure-slam-pointline_amd/lib/variants/clcyc.so
run SSLAM_LIB=$V SSLAM_CL_WGS=8 SSLAM_CL_WINDOW=32
run SSLAM_LIB=$V SSLAM_CL_WGS=8 SSLAM_CL_WINDOW=48
run SSLAM_LIB=$V SSLAM_CL_WGS=8 SSLAM_CL_WINDOW=96
run SSLAM_LIB=$V SSLAM_CL_WGS=12 SSLAM_CL_WINDOW=64
