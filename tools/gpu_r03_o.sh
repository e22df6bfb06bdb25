#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03o; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
python - <<PY
import json
d=json.load(open('$O/bench.json')); print(round(d['value']), d['ms_per_step'], d['roofline']['traffic'], d.get('latency'), d.get('pcie_inclusive'))
PY
