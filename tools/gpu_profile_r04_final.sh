#!/bin/bash
# Round-4 measurement recipe, final state of the round (after tools/gpu_r04_experiments.sh h..q; run through gpurun).  PMC passes first (they regenerate profiles/pmc_traffic.json on the box so that the bench line written
# afterwards carries roofline.traffic from counters taken on the kernels it times), then the GPU suite, the bench line, kernel traces.  Counters in
# their own passes, no trace domains mixed in.  Every step under its own timeout.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
b=6144
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_c3_$c && timeout 600 rocprofv3 --pmc $c -d $O/pmc_c3_$c -- python $R/bench.py --workload c3 --batch $b --steps 1 --warmup 1 --no-overlap --no-cpu-baseline --no-extras --no-profile --no-other-workloads > $O/pmc_c3_$c.log 2>&1
done
cd $R
python tools/rocpd_pmc_summary.py $O/pmc_c3_FETCH_SIZE $O/pmc_fetch_c3.txt > /dev/null
python tools/rocpd_pmc_summary.py $O/pmc_c3_WRITE_SIZE $O/pmc_write_c3.txt > /dev/null
python tools/make_pmc_traffic.py $O/pmc_c3_FETCH_SIZE $O/pmc_c3_WRITE_SIZE $b 3 2 $O/pmc_traffic_c3.json | head -24
cp $O/pmc_traffic_c3.json $R/profiles/pmc_traffic.json
rm -rf $O/pmc_c3_FETCH_SIZE $O/pmc_c3_WRITE_SIZE
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > $O/pytest_gpu.txt 2>&1; tail -14 $O/pytest_gpu.txt
timeout 1200 python bench.py > $O/bench_r04.json 2> $O/bench_r04.err; tail -c 300 $O/bench_r04.err
timeout 400 python bench.py --no-overlap --no-cpu-baseline --no-extras --no-other-workloads > $O/bench_r04_one_stream.json 2>/dev/null
cd /tmp
rm -rf $O/kt && timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/bench.py --no-cpu-baseline --no-extras --no-profile --no-other-workloads > $O/kt.log 2>&1
rm -rf $O/kt1 && timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt1 -- python $R/bench.py --no-overlap --no-cpu-baseline --no-extras --no-profile --no-other-workloads > $O/kt1.log 2>&1
rm -rf $O/lat && cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $O/lat -- python tools/latency_probe.py > $O/lat.log 2>&1
cd $R
python tools/rocpd_summary.py $O/lat $O/kernel_trace_single_frame.txt > /dev/null; rm -rf $O/lat
python tools/rocpd_summary.py $O/kt $O/kernel_trace_two_streams.txt > /dev/null
python tools/rocpd_summary.py $O/kt1 $O/kernel_trace_one_stream.txt > /dev/null
rm -rf $O/kt $O/kt1
head -12 $O/kernel_trace_one_stream.txt
python - <<'PY'
import json
for n in ('bench_r04','bench_r04_one_stream'):
    try:
        d=json.load(open('gpurun_out/r04/%s.json'%n)); print(n, round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])
        if 'latency' in d: print(d['latency']['lines_extract_hipEvent'], {k: v for k, v in d['pcie_inclusive'].items() if 'per_s' in k}); print(d.get('other_workloads')); print(d['cpu_baseline']['value'], d['cpu_baseline']['parity_vs_gpu'])
    except Exception as e: print(n, 'failed', e)
PY
