#!/bin/bash
# Round 6, GPU call K: how repeatable is the two-stream step?  The final recipe saw two states (FAST beside the core 69 or 78 ms; 159-161 or 165-166 ms per step).  Ten runs of
# tools/step_check back to back on one box, three short bench.py runs, and the grid a little under 16 per compute unit.
set -x
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06k; mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/rep_$i.txt 2>&1; head -2 $O/rep_$i.txt | cut -c1-300; done
for g in 3968 3840 3584; do SSLAM_LSD_PERSIST=$g STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/grid_$g.txt 2>&1; head -2 $O/grid_$g.txt | cut -c1-300; SSLAM_LSD_PERSIST=$g STEP_PROFILE=1 timeout 100 tools/step_check 12288 5 2 > $O/grid_${g}_b.txt 2>&1; head -1 $O/grid_${g}_b.txt; done
for i in 1 2 3; do timeout 300 python bench.py --no-extras --no-cpu-baseline --no-other-workloads > $O/bench_$i.json 2>/dev/null; python - <<PY
import json
d = json.loads(open('gpurun_out/r06k/bench_$i.json').read().strip().splitlines()[-1]); k = d['roofline']['kernels_ms_per_step']; print('bench $i', round(d['value']), round(d['ms_per_step'], 1), round(k['k_fast_cells'], 1), round(k['k_lsd_regions'], 1), round(k['k_blur_sobel'], 1))
PY
done
