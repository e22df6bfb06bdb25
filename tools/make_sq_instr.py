#!/usr/bin/env python3
"""profiles/sq_instr.json from a per-kernel SQ table (tools/sq_table5.py output: columns VALU/frm, SALU/frm, ...): the wave-instruction counts per frame that
bench.py's roofline.issue block prices against the chip's issue rate (1 024 SIMDs, one VALU wave-instruction per 4 cycles per SIMD).  Counters cannot be collected
in the bench process itself (separate rocprofv3 --pmc passes), so the table is committed and bench.py reads it -- as profiles/pmc_traffic.json feeds roofline.traffic.
    python tools/make_sq_instr.py <sq_table.txt> <batch> <out.json> [source note] [tcp_table.txt]
The optional TCP table (tools/tcp_table.py output) adds l2_reads_per_frame / l1_accesses_per_frame: what roofline.random_sector prices against tools/gather_probe."""
import sys, json
rows = {}
for l in open(sys.argv[1]).read().splitlines():
    if "|" not in l or l.startswith("kernel"): continue
    left, right = l.split("|")
    name = left.split()[0].split("<")[0]
    r = right.split()
    d = rows.setdefault(name, {"valu_per_frame": 0.0, "salu_per_frame": 0.0, "lds_per_frame": 0.0, "vmem_rd_per_frame": 0.0, "vmem_wr_per_frame": 0.0})
    for k, v in zip(("valu_per_frame", "salu_per_frame", "lds_per_frame", "vmem_rd_per_frame", "vmem_wr_per_frame"), r): d[k] += float(v)
if len(sys.argv) > 5:
    for l in open(sys.argv[5]).read().splitlines():
        if "|" not in l or l.startswith("kernel") or l.startswith("#"): continue
        left, right = l.split("|")
        name = left.split()[0].split("<")[0]; r = right.split()
        full = left.split("|")[0].rsplit(None, 3)[0].strip()      # kernel name with its template arguments
        if name in rows and len(r) >= 3:
            # two launch forms of one kernel (the sequential core: the six-wave form primes the previous frames, the guest form runs the two-stream step): the step's form wins
            if "l2_reads_per_frame" in rows[name] and "<false, 4>" not in full: continue
            try: rows[name]["l2_reads_per_frame"] = float(r[0]); rows[name]["l2_latency_cycles"] = float(r[1]); rows[name]["l1_accesses_per_frame"] = float(r[2]); rows[name]["tcp_row"] = full
            except ValueError: pass
step = {k: v for k, v in rows.items() if k not in ("k_grad_smin", "k_grad_table")}      # (one-time table kernels are not part of a step)
out = {"source": sys.argv[4] if len(sys.argv) > 4 else "rocprofv3 --pmc SQ_INSTS_* passes over tools/step_check (one stream), tools/sq_table5.py", "batch": int(sys.argv[2]), "kernels": step,
       "valu_per_frame_total": sum(v["valu_per_frame"] for v in step.values()), "salu_per_frame_total": sum(v["salu_per_frame"] for v in step.values())}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(out["valu_per_frame_total"], out["salu_per_frame_total"])
