#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/lat && cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $O/lat -- python tools/latency_probe.py > $O/lat.log 2>&1
python tools/rocpd_summary.py $O/lat $O/kernel_trace_single_frame.txt > /dev/null; rm -rf $O/lat
head -4 $O/kernel_trace_single_frame.txt
timeout 300 python tools/cl_probe.py 64 > $O/cl_probe.txt 2>&1; tail -2 $O/cl_probe.txt
