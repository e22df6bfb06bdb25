#!/bin/bash
# Round-1 measurement recipe (run through gpurun): bench line, kernel trace, PMC passes.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R && timeout 900 python bench.py > $O/bench_r01.json 2> $O/bench_r01.err; tail -c 600 $O/bench_r01.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt && timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/bench.py --no-cpu-baseline --no-profile > $O/kt.log 2>&1
rm -rf $O/pmc_fetch && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -- python $R/bench.py --batch 512 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_fetch.log 2>&1
rm -rf $O/pmc_write && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -- python $R/bench.py --batch 512 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_write.log 2>&1
ls -R $O | head -40
