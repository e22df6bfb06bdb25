#!/usr/bin/env python3
"""Lane occupancy of the vector instructions per kernel from one rocprofv3 --pmc pass (tools/gpu_profile_r05_final.sh, the third SQ pass):
   tools/sq_lanes.py <sq3.txt> [batch]     (the per-kernel sums of tools/rocpd_pmc_summary.py)
lanes = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (the ratio rocprofiler-sdk calls AvgNumActiveThreads): how many of a wave's 64 lanes the EXEC mask enables per vector
instruction, averaged over the kernel.  It counts ENABLED lanes: a wave-uniform computation that every lane executes for one pixel (the accept chain of k_lsd_regions) counts 64."""
import sys, collections
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
rows = collections.defaultdict(dict)
for l in open(sys.argv[1]).read().splitlines()[1:]:
    p = l.rsplit(None, 4)
    if len(p) < 5: continue
    k = p[0].split("(")[0].strip()
    try: rows[k][p[1]] = rows[k].get(p[1], 0) + float(p[3])
    except ValueError: pass
print("%-30s %7s %12s %12s %10s" % ("kernel", "lanes", "VALU/frame", "SALUcyc/frm", "GUI Mcyc"))
for k, r in sorted(rows.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    a = r.get("SQ_ACTIVE_INST_VALU", 0)
    if not k.startswith("k_") or a <= 0: continue
    passes = 1 if k.split("<")[0] in ("k_knn2_mfma", "k_knn2_expand", "k_search_init_lds", "k_search_init", "k_line_match") else 2      # previous-frame priming + 1 step
    print("%-30s %7.1f %12.0f %12.0f %10.1f" % (k[:30], r.get("SQ_THREAD_CYCLES_VALU", 0) / a, r.get("SQ_INSTS_VALU", 0) / (B * passes), r.get("SQ_INST_CYCLES_SALU", 0) / (B * passes), r.get("GRBM_GUI_ACTIVE", 0) / 1e6))
