#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/mk; cd $R && SSLAM_ROCTX=1 timeout 300 rocprofv3 --marker-trace --kernel-trace -d $O/mk -- python tools/latency_probe.py > $O/mk.log 2>&1; tail -2 $O/mk.log
python tools/roctx_check.py $O/mk $O/roctx_markers.txt | head -30
rm -rf $O/mk
