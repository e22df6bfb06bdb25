#!/bin/bash
# Round 6, GPU call T: a longer randomised parity sweep on the final tree (every third case a 5-to-4 geometry: fused gradient kernel; all cases: the sort on tile-sorted runs), two seeds, + the matcher fuzzer
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06t; mkdir -p $O
timeout 900 python tools/fuzz_parity.py 1500 20261002 > $O/fuzz_parity_1500_20261002.txt 2>&1; tail -4 $O/fuzz_parity_1500_20261002.txt
timeout 600 python tools/fuzz_parity.py 900 777 > $O/fuzz_parity_900_777.txt 2>&1; tail -4 $O/fuzz_parity_900_777.txt
timeout 300 python tools/fuzz_matchers.py 4000 > $O/fuzz_matchers_4000.txt 2>&1; tail -2 $O/fuzz_matchers_4000.txt
