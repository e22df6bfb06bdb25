#!/bin/bash
# round 4, call e: the whole GPU suite on the final tree (log kept in profiles/), smoke
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04e; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu --durations=5 > $O/pytest_gpu.txt 2>&1; tail -12 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
