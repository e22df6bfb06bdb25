#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r03; mkdir -p $O
timeout 1200 python tools/fuzz_parity.py 1500 777 > $O/fuzz_parity_1500_777.txt 2>&1; tail -1 $O/fuzz_parity_1500_777.txt
timeout 900 python bench.py > $O/bench_r03.json 2> $O/bench_r03.err
python -c "
import json
d=json.load(open('$O/bench_r03.json')); print(round(d['value']), d['ms_per_step'], d['latency']['lines_extract_hipEvent'], {k: round(v) for k, v in d['pcie_inclusive'].items() if 'per_s' in k})"
