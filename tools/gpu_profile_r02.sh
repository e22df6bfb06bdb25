#!/bin/bash
# Round-2 measurement recipe (run through gpurun): bench line, kernel trace of the same command, PMC passes (640x480 and the 1280x960
# "rocprof HBM-GB/s run" of BASELINE configs[3]), per-kernel instruction table.  Counters in their own passes, no trace domains mixed in.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R && timeout 900 python bench.py > $O/bench_r02.json 2> $O/bench_r02.err; tail -c 300 $O/bench_r02.err
cd $R && timeout 600 python bench.py --no-overlap --no-cpu-baseline --no-extras > $O/bench_r02_one_stream.json 2>/dev/null
cd $R && timeout 600 python bench.py --workload c4 --no-cpu-baseline > $O/bench_r02_c4.json 2>/dev/null
cd $R && timeout 600 python bench.py --unique 8 --no-overlap --no-cpu-baseline --no-extras > $O/bench_r02_8frames_one_stream.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt && timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/bench.py --no-cpu-baseline --no-extras --no-profile > $O/kt.log 2>&1
rm -rf $O/kt1 && timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt1 -- python $R/bench.py --no-overlap --no-cpu-baseline --no-extras --no-profile > $O/kt1.log 2>&1
for wl in c3 c4; do
  b=512; [ $wl = c4 ] && b=128
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${wl}_$c && timeout 600 rocprofv3 --pmc $c -d $O/pmc_${wl}_$c -- python $R/bench.py --workload $wl --batch $b --steps 1 --warmup 1 --no-overlap --no-cpu-baseline --no-extras --no-profile > $O/pmc_${wl}_$c.log 2>&1
  done
done
rm -rf $O/lat && cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $O/lat -- python tools/latency_probe.py > $O/lat.log 2>&1      # one frame per call: the multi-wave core
cd $R
python tools/rocpd_summary.py $O/lat $O/kernel_trace_single_frame.txt > /dev/null; rm -rf $O/lat
python tools/rocpd_summary.py $O/kt $O/kernel_trace_two_streams.txt > /dev/null
python tools/rocpd_summary.py $O/kt1 $O/kernel_trace_one_stream.txt > /dev/null
for wl in c3 c4; do
  b=512; [ $wl = c4 ] && b=128
  python tools/rocpd_pmc_summary.py $O/pmc_${wl}_FETCH_SIZE $O/pmc_fetch_$wl.txt > /dev/null
  python tools/rocpd_pmc_summary.py $O/pmc_${wl}_WRITE_SIZE $O/pmc_write_$wl.txt > /dev/null
  python tools/make_pmc_traffic.py $O/pmc_${wl}_FETCH_SIZE $O/pmc_${wl}_WRITE_SIZE $b 3 2 $O/pmc_traffic_$wl.json | head -12
done
rm -rf $O/kt $O/kt1 $O/pmc_c3_FETCH_SIZE $O/pmc_c3_WRITE_SIZE $O/pmc_c4_FETCH_SIZE $O/pmc_c4_WRITE_SIZE
head -12 $O/kernel_trace_one_stream.txt
