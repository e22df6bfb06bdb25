#!/bin/bash
# Round 6, GPU call AN: fuzz campaign on the final tree (the late kernel changes: transposed LBD gathers, NFA votes without masks / scalar geometry, fp32-guarded bins, deeper sort loads)
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06an; mkdir -p $O
timeout 900 python tools/fuzz_parity.py 2000 20261003 > $O/fuzz_parity_2000_20261003.txt 2>&1; tail -3 $O/fuzz_parity_2000_20261003.txt
timeout 600 python tools/fuzz_parity.py 1200 555 > $O/fuzz_parity_1200_555.txt 2>&1; tail -3 $O/fuzz_parity_1200_555.txt
timeout 300 python tools/fuzz_matchers.py 3000 > $O/fuzz_matchers_3000.txt 2>&1; tail -2 $O/fuzz_matchers_3000.txt
timeout 300 python tools/fuzz_reuse.py > $O/fuzz_reuse.txt 2>&1; tail -2 $O/fuzz_reuse.txt
