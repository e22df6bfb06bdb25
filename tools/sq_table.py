#!/usr/bin/env python3
"""Per-kernel SQ utilisation from two rocprofv3 --pmc passes over `bench.py --no-overlap --steps 1 --warmup 0` (tools/gpu_r03_e.sh):
   tools/sq_table.py <sq1.txt> <sq2.txt> [batch]   (the per-kernel sums of tools/rocpd_pmc_summary.py)
SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md), GRBM_GUI_ACTIVE and SQ_BUSY_CU_CYCLES cycles;
utilisation = 4 x ACTIVE / (GUI_ACTIVE x 1024 SIMDs)."""
import sys, collections
B = int(sys.argv[3]) if len(sys.argv) > 3 else 12288
rows = collections.defaultdict(dict)
for f in sys.argv[1:3]:
    for l in open(f).read().splitlines()[1:]:
        p = l.rsplit(None, 4)
        if len(p) < 5: continue
        k = p[0].split("<")[0].split("(")[0].strip()
        try: rows[k][p[1]] = rows[k].get(p[1], 0) + float(p[3])
        except ValueError: pass
NS = 1024
tot = collections.Counter()
print("%-16s %9s %8s %7s %7s %7s %7s %7s | %9s %9s" % ("kernel", "GUI_Mcyc", "busyCU%", "VALU%", "SALU%", "VMEM%", "LDS%", "WAIT%", "VALU/frm", "SALU/frm"))
for k, r in sorted(rows.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    if not k.startswith("k_"): continue
    gui = r.get("GRBM_GUI_ACTIVE", 0) / 2          # both passes recorded it
    if gui <= 0: continue
    passes = 1 if k in ("k_knn2_batch", "k_search_init", "k_line_match") else 2      # previous-frame priming + 1 step
    fr = B * passes
    simd = gui * NS
    print("%-16s %9.1f %8.1f %7.1f %7.1f %7.1f %7.1f %7.1f | %9.0f %9.0f" % (
        k, gui / passes / 1e6, 100 * r.get("SQ_BUSY_CU_CYCLES", 0) / (gui * 256), 100 * 4 * r.get("SQ_ACTIVE_INST_VALU", 0) / simd,
        100 * 4 * r.get("SQ_ACTIVE_INST_SCA", 0) / simd, 100 * 4 * r.get("SQ_ACTIVE_INST_VMEM", 0) / simd, 100 * 4 * r.get("SQ_ACTIVE_INST_LDS", 0) / simd,
        100 * r.get("SQ_WAIT_ANY", 0) / max(r.get("SQ_WAVE_CYCLES", 1), 1), r.get("SQ_INSTS_VALU", 0) / fr, r.get("SQ_INSTS_SALU", 0) / fr))
    for c in ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_SALU"): tot[c] += r.get(c, 0) / passes
    tot["gui"] += gui / passes
print("one step of %d frames, kernels back to back: %.1f M GPU cycles; VALU pipes active %.1f %% of all SIMD cycles (4 cycles per wave-instruction as counted); "
      "by the 2-cycle issue figure %.1f %%; %.2f M VALU + %.2f M SALU wave-instructions per frame"
      % (B, tot["gui"] / 1e6, 100 * 4 * tot["SQ_ACTIVE_INST_VALU"] / (tot["gui"] * NS), 100 * 2 * tot["SQ_INSTS_VALU"] / (tot["gui"] * NS), tot["SQ_INSTS_VALU"] / B / 1e6, tot["SQ_INSTS_SALU"] / B / 1e6))
