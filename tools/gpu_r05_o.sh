#!/bin/bash
# Round 5, GPU call O: the instruction cache during one step (k_nfa_all is ~65 KB of code, the core ~42 KB; the cache is 64 KB per two CUs).
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r05o; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/ic -- tools/step_check 3072 1 0 1 > $O/ic.log 2>&1; python tools/rocpd_pmc_summary.py $O/ic $O/ic.txt > /dev/null; rm -rf $O/ic; tail -3 $O/ic.log; head -40 $O/ic.txt)
