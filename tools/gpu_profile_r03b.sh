#!/bin/bash
# Round-3 measurement recipe, second part (after the cluster form of the LSD core became what single frames take): the GPU test log, the
# bench line (latency and PCIe legs changed), the single-frame kernel trace, the fuzz logs.  tools/gpu_profile_r03.sh holds the first part
# (PMC traffic on the timed launch form, batch kernel traces); the batch kernels have not changed since.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_r03.json 2> $O/bench_r03.err; tail -c 300 $O/bench_r03.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/lat && cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $O/lat -- python tools/latency_probe.py > $O/lat.log 2>&1
python tools/rocpd_summary.py $O/lat $O/kernel_trace_single_frame.txt > /dev/null; rm -rf $O/lat
timeout 300 python tools/cl_probe.py 64 > $O/cl_probe.txt 2>&1; tail -2 $O/cl_probe.txt
timeout 1200 python tools/fuzz_parity.py 1500 777 > $O/fuzz_parity_1500_777.txt 2>&1; tail -2 $O/fuzz_parity_1500_777.txt
timeout 600 python tools/fuzz_matchers.py > $O/fuzz_matchers.txt 2>&1; tail -1 $O/fuzz_matchers.txt
timeout 600 python tools/fuzz_reuse.py > $O/fuzz_reuse.txt 2>&1; tail -1 $O/fuzz_reuse.txt
head -8 $O/kernel_trace_single_frame.txt
python -c "
import json
d=json.load(open('$O/bench_r03.json')); print(round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic']); print(d['latency']['lines_extract_hipEvent'], d['latency']['lsd_core']); print({k: v for k, v in d['pcie_inclusive'].items() if 'per_s' in k or 'matches' in k})"
cd $R && timeout 600 python bench.py --workload c4 --no-cpu-baseline > $O/bench_r03_c4.json 2>/dev/null
python -c "
import json
d=json.load(open('$O/bench_r03_c4.json')); print('c4', round(d['value']), d['ms_per_step'], d['latency']['lines_extract_hipEvent'], d['latency']['lsd_core'][:40])"
timeout 300 python tools/small_batch_probe.py > $O/small_batches.txt 2>&1; SSLAM_LSD_FLAVOUR=mw timeout 300 python tools/small_batch_probe.py > $O/small_batches_mw.txt 2>&1; tail -3 $O/small_batches.txt
