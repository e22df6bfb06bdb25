#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r03; mkdir -p $O
timeout 600 python tools/cl_stress.py 300 6 > $O/cl_stress.txt 2>&1; tail -3 $O/cl_stress.txt
timeout 900 python tools/fuzz_parity.py 600 4242 > $O/fuzz_parity_600_4242.txt 2>&1; tail -1 $O/fuzz_parity_600_4242.txt
