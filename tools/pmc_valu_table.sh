#!/bin/bash
# VALU / SALU / LDS instruction counts per kernel for one bench step (B=6144, one stream): where the issue slots of a saturated GPU go.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_valu; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/p
(cd $R && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES -d $O/p -- python bench.py --no-overlap --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-profile > $O/p.log 2>&1)
(cd $R && python tools/rocpd_pmc_summary.py $O/p $O/table.txt > /dev/null; python - <<'PY'
import re,collections
rows=collections.defaultdict(dict)
for l in open("gpurun_out/pmc_valu/table.txt").read().splitlines()[1:]:
    p=l.split()
    if len(p)<5 or not p[0].startswith("k_"): continue
    rows[p[0].split("<")[0]][p[1]]=rows[p[0].split("<")[0]].get(p[1],0)+float(p[3])
tot=sum(r.get("SQ_INSTS_VALU",0) for r in rows.values())
print("%-20s %12s %12s %10s %8s  (2 passes of 6144 frames: previous-frame priming + 1 step; matchers 1 pass)"%("kernel","VALU/frame","SALU/frame","LDS/frame","VALU %"))
for k,r in sorted(rows.items(), key=lambda kv:-kv[1].get("SQ_INSTS_VALU",0)):
    n=6144*(1 if k in("k_knn2_batch","k_search_init","k_line_match") else 2)
    print("%-20s %12.0f %12.0f %10.0f %8.1f"%(k,r.get("SQ_INSTS_VALU",0)/n,r.get("SQ_INSTS_SALU",0)/n,r.get("SQ_INSTS_LDS",0)/n,100*r.get("SQ_INSTS_VALU",0)/tot))
PY
)
rm -rf $O/p
