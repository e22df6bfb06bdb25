#!/bin/bash
# round 4, call f: the long randomised sweeps on the final kernels (logs -> profiles/r04_fuzz_*)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O
cd $R
timeout 900 python tools/fuzz_parity.py 1500 20260926 > $O/fuzz_parity_1500.txt 2>&1; tail -2 $O/fuzz_parity_1500.txt
timeout 600 python tools/fuzz_matchers.py > $O/fuzz_matchers.txt 2>&1; tail -2 $O/fuzz_matchers.txt
timeout 600 python tools/fuzz_reuse.py > $O/fuzz_reuse.txt 2>&1; tail -1 $O/fuzz_reuse.txt
timeout 300 python tools/cl_stress.py 300 6 > $O/cl_stress.txt 2>&1; tail -1 $O/cl_stress.txt
timeout 300 python tools/bench_matchers.py > $O/matchers.txt 2>&1; tail -14 $O/matchers.txt
