#!/bin/bash
# Round 5, GPU call R (last of the round, final tree): a third seed of the randomised parity sweep, a longer matcher sweep, the bench line once more (box-to-box spread).
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r05r; mkdir -p $O
timeout 500 python tools/fuzz_parity.py 1500 20261001 > $O/fuzz_parity_c.txt 2>&1; tail -3 $O/fuzz_parity_c.txt
timeout 200 python tools/fuzz_matchers.py 4000 > $O/fuzz_matchers_b.txt 2>&1; tail -2 $O/fuzz_matchers_b.txt
timeout 600 python bench.py > $O/bench_again.json 2> $O/bench_again.err; tail -c 200 $O/bench_again.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05r/bench_again.json').read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['latency']['lines_extract_hipEvent'], d['other_workloads']['c4']['value'], d['other_workloads']['c3_lsd_nfa_variant_0']['value'])
PY
