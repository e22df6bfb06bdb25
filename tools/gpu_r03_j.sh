#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03j; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; grep -n "passed\|failed\|Error" $O/pytest.txt | head -5
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03j/bench.json"))
print("fps", d["value"], "ms", d["ms_per_step"]); print("roofline", {k:v for k,v in d["roofline"].items() if k in ("kernel","frac","traffic","traffic_source","avg_launch_ms")})
print("latency", d.get("latency",{}).get("lines_extract_hipEvent"), d.get("latency",{}).get("orb_extract_hipEvent"), d.get("latency_error"))
print("pcie", d.get("pcie_inclusive")); print("cpu", {k:v for k,v in d.get("cpu_baseline",{}).items() if k in ("value","parity_vs_gpu")})
print({k: round(v,1) for k,v in list(d["roofline"]["kernels_ms_per_step"].items())[:14]})
PY
timeout 600 python bench.py --workload c4 --no-cpu-baseline --no-extras > $O/bench_c4.json 2>> $O/bench.err; cut -c1-250 $O/bench_c4.json
