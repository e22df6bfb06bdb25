#!/bin/bash
# Round-3 measurement recipe (run through gpurun).  Order matters: the PMC passes come first and regenerate profiles/pmc_traffic*.json
# on the box, so that the bench lines written afterwards carry roofline.traffic from counters taken on the launch form they time
# (bench.py refuses a traffic file whose lsd_core differs).  Counters in their own passes, no trace domains mixed in.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in c3 c4; do
  b=6144; [ $wl = c4 ] && b=1024          # >= 1024 frames: the same k_lsd_regions<false> launch the bench line times
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${wl}_$c && timeout 900 rocprofv3 --pmc $c -d $O/pmc_${wl}_$c -- python $R/bench.py --workload $wl --batch $b --steps 1 --warmup 1 --no-overlap --no-cpu-baseline --no-extras --no-profile > $O/pmc_${wl}_$c.log 2>&1
  done
done
cd $R
for wl in c3 c4; do
  b=6144; [ $wl = c4 ] && b=1024
  python tools/rocpd_pmc_summary.py $O/pmc_${wl}_FETCH_SIZE $O/pmc_fetch_$wl.txt > /dev/null
  python tools/rocpd_pmc_summary.py $O/pmc_${wl}_WRITE_SIZE $O/pmc_write_$wl.txt > /dev/null
  python tools/make_pmc_traffic.py $O/pmc_${wl}_FETCH_SIZE $O/pmc_${wl}_WRITE_SIZE $b 3 2 $O/pmc_traffic_$wl.json | head -14
done
cp $O/pmc_traffic_c3.json $R/profiles/pmc_traffic.json; cp $O/pmc_traffic_c4.json $R/profiles/pmc_traffic_c4.json
rm -rf $O/pmc_c3_FETCH_SIZE $O/pmc_c3_WRITE_SIZE $O/pmc_c4_FETCH_SIZE $O/pmc_c4_WRITE_SIZE
timeout 900 python bench.py > $O/bench_r03.json 2> $O/bench_r03.err; tail -c 600 $O/bench_r03.err
timeout 600 python bench.py --no-overlap --no-cpu-baseline --no-extras > $O/bench_r03_one_stream.json 2>/dev/null
timeout 600 python bench.py --workload c4 --no-cpu-baseline > $O/bench_r03_c4.json 2>/dev/null
cd /tmp
rm -rf $O/kt && timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/bench.py --no-cpu-baseline --no-extras --no-profile > $O/kt.log 2>&1
rm -rf $O/kt1 && timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt1 -- python $R/bench.py --no-overlap --no-cpu-baseline --no-extras --no-profile > $O/kt1.log 2>&1
rm -rf $O/lat && cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $O/lat -- python tools/latency_probe.py > $O/lat.log 2>&1
cd $R
python tools/rocpd_summary.py $O/lat $O/kernel_trace_single_frame.txt > /dev/null; rm -rf $O/lat
python tools/rocpd_summary.py $O/kt $O/kernel_trace_two_streams.txt > /dev/null
python tools/rocpd_summary.py $O/kt1 $O/kernel_trace_one_stream.txt > /dev/null
rm -rf $O/kt $O/kt1
timeout 120 python tools/valu_rate.py > $O/valu_rate.txt 2>&1
timeout 900 python tools/fuzz_parity.py 1500 777 > $O/fuzz_parity_1500_777.txt 2>&1; tail -3 $O/fuzz_parity_1500_777.txt
timeout 600 python tools/fuzz_matchers.py > $O/fuzz_matchers.txt 2>&1; tail -3 $O/fuzz_matchers.txt
timeout 600 python tools/fuzz_reuse.py > $O/fuzz_reuse.txt 2>&1; tail -3 $O/fuzz_reuse.txt
head -14 $O/kernel_trace_one_stream.txt
python -c "
import json
for n in ('bench_r03','bench_r03_one_stream','bench_r03_c4'):
    d=json.load(open('$O/%s.json'%n)); print(n, round(d['value']), d['ms_per_step'], d['roofline'])"
