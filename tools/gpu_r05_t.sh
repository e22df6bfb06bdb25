#!/bin/bash
# Round 5, GPU call T: last sanity run on the final tree (comment-only changes since the final recipe): smoke(), the whole GPU suite, the bench line.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r05t; mkdir -p $O
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 600 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
timeout 400 python bench.py --no-other-workloads > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05t/bench.json').read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['latency']['lines_extract_hipEvent'], d['cpu_baseline']['parity_vs_gpu'])
PY
