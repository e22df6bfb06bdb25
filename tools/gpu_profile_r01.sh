#!/bin/bash
# Round-1 measurement recipe (run through gpurun): bench line, kernel trace, PMC passes, FETCH_SIZE calibration.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R && timeout 900 python bench.py > $O/bench_r01.json 2> $O/bench_r01.err; tail -c 600 $O/bench_r01.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt && timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/bench.py --no-cpu-baseline --no-profile > $O/kt.log 2>&1
rm -rf $O/pmc_fetch && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -- python $R/bench.py --batch 512 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_fetch.log 2>&1
rm -rf $O/pmc_write && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -- python $R/bench.py --batch 512 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_write.log 2>&1
rm -rf $O/pmc_probe && (cd $R && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_probe -- python tools/fetch_probe.py > $O/pmc_probe.log 2>&1)
cd $R
python tools/rocpd_summary.py $O/kt $O/kernel_trace.txt > /dev/null
python tools/rocpd_pmc_summary.py $O/pmc_fetch $O/pmc_fetch.txt > /dev/null
python tools/rocpd_pmc_summary.py $O/pmc_write $O/pmc_write.txt > /dev/null
python tools/rocpd_pmc_summary.py $O/pmc_probe $O/pmc_probe.txt
python tools/make_pmc_traffic.py $O/pmc_fetch $O/pmc_write 512 3 2 $O/pmc_traffic.json | head -30
head -30 $O/kernel_trace.txt
