#!/usr/bin/env python3
"""Per-kernel SQ table from the three rocprofv3 --pmc passes of tools/gpu_profile_r05_final.sh over `tools/step_check 3072 1 0 1` (one stream; previous-frame priming + 1 step):
   tools/sq_table5.py <sq1.txt> <sq2.txt> <sq3.txt> [batch]      (the per-kernel sums of tools/rocpd_pmc_summary.py)
GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3, SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md).
  ms        GUI cycles of one pass / 8 XCDs / 2.4 GHz (the kernel with the chip to itself)
  VALU% / SALU%   4 x SQ_ACTIVE_INST_VALU (_SCA) / (GUI / 8 x 1024 SIMDs): share of all SIMD cycles the vector (scalar) pipe is busy
  WAIT%     SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of the resident waves' cycles spent parked at a wait
  lanes     SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU: lanes the EXEC mask enables per vector instruction
  */frame   wave-instructions per frame"""
import sys, collections
B = int(sys.argv[4]) if len(sys.argv) > 4 else 3072
rows = collections.defaultdict(dict)
for f in sys.argv[1:4]:
    for l in open(f).read().splitlines()[1:]:
        p = l.rsplit(None, 4)
        if len(p) < 5: continue
        k = p[0].split("(")[0].strip()
        try: rows[k][p[1]] = float(p[3])          # (a counter that is in two passes has the same value in both)
        except ValueError: pass
MATCH = ("k_knn2_mfma", "k_knn2_expand", "k_search_init_lds", "k_search_init", "k_line_match", "k_knn2_batch")
print("%-24s %7s %6s %6s %6s %6s | %9s %9s %8s %8s %8s" % ("kernel", "ms", "VALU%", "SALU%", "WAIT%", "lanes", "VALU/frm", "SALU/frm", "LDS/frm", "VMEMrd", "VMEMwr"))
tot = collections.Counter()
for k, r in sorted(rows.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    gui = r.get("GRBM_GUI_ACTIVE", 0)
    if not k.startswith("k_") or gui <= 0: continue
    passes = 1 if k.split("<")[0] in MATCH else 2
    fr = B * passes
    simd = gui / 8 * 1024
    a = r.get("SQ_ACTIVE_INST_VALU", 0)
    print("%-24s %7.2f %6.1f %6.1f %6.1f %6.1f | %9.0f %9.0f %8.0f %8.0f %8.0f" % (
        k[:24], gui / 8 / passes / 2.4e6, 100 * 4 * a / simd, 100 * 4 * r.get("SQ_ACTIVE_INST_SCA", 0) / simd, 100 * r.get("SQ_WAIT_ANY", 0) / max(r.get("SQ_WAVE_CYCLES", 1), 1),
        r.get("SQ_THREAD_CYCLES_VALU", 0) / a if a else 0, r.get("SQ_INSTS_VALU", 0) / fr, r.get("SQ_INSTS_SALU", 0) / fr, r.get("SQ_INSTS_LDS", 0) / fr, r.get("SQ_INSTS_VMEM_RD", 0) / fr, r.get("SQ_INSTS_VMEM_WR", 0) / fr))
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU"): tot[c] += r.get(c, 0) / passes
    tot["gui"] += gui / passes
print("one step of %d frames, kernels back to back: %.1f ms of GPU time; vector pipes busy %.1f %% of all SIMD cycles; %.2f M VALU + %.2f M SALU wave-instructions per frame"
      % (B, tot["gui"] / 8 / 2.4e6, 100 * 4 * tot["SQ_ACTIVE_INST_VALU"] / (tot["gui"] / 8 * 1024), tot["SQ_INSTS_VALU"] / B / 1e6, tot["SQ_INSTS_SALU"] / B / 1e6))
