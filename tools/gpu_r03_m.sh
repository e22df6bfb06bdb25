#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
run() { echo "== $*"; env "$@" timeout 200 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-110; }
V=$R/structure-slam-pointline_amd/lib/variants
run A=1
run SSLAM_LIB=$V/clg1.so
run SSLAM_LIB=$V/clg2.so
run SSLAM_LIB=$V/clg8.so
