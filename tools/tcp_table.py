"""Per-kernel table of the vector L1 (TCP) counters from two rocprofv3 --pmc passes of tools/step_check 3072 1 0 1 (summaries by tools/rocpd_pmc_summary.py):
pass 1: TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
pass 2: TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum
   python tools/tcp_table.py pass1.txt pass2.txt <frames per launch>"""
import sys, collections
def load(path):
    d = collections.defaultdict(dict)
    for ln in open(path):
        f = ln.split()
        if len(f) < 5 or f[0] == "kernel": continue
        try: launches, total = int(f[-3]), float(f[-2])
        except ValueError: continue
        d[" ".join(f[:-4])][f[-4]] = (launches, total)
    return d
a, b = load(sys.argv[1]), load(sys.argv[2]); nf = int(sys.argv[3])
print("%-26s %9s %9s %9s | %11s %10s %9s" % ("kernel", "pending%", "tagconf%", "L1 hit%", "L2 rd/frame", "L2 lat cyc", "acc/frame"))
rows = []
for k in b:
    g = b[k].get("TCP_GATE_EN1_sum")
    if not g or k not in a: continue
    launches, gate = g
    get = lambda d, n: d[k].get(n, (launches, 0.0))[1]
    frames = nf * launches if not k.startswith("k_knn2") and not k.startswith("k_search") and not k.startswith("k_line_match") else nf
    rd, acc = get(a, "TCP_TCC_READ_REQ_sum"), get(b, "TCP_TOTAL_CACHE_ACCESSES_sum")
    rows.append((gate, "%-26s %9.1f %9.1f %9.1f | %11.0f %10.0f %9.0f" % (k[:26], 100 * get(a, "TCP_PENDING_STALL_CYCLES_sum") / gate, 100 * get(a, "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum") / gate,
                100 * (1 - rd / acc) if acc else float("nan"), rd / frames, get(b, "TCP_TCC_READ_REQ_LATENCY_sum") / rd if rd else float("nan"), acc / frames)))
for _, r in sorted(rows, reverse=True): print(r)
print("pending% / tagconf%: TCP_PENDING_STALL_CYCLES / TCP_READ_TAGCONFLICT_STALL_CYCLES over TCP_GATE_EN1 (cycles the 256 L1s were clocked during the kernel); L1 hit% = 1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES;")
print("L2 lat = TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ (cycles from the L1's request to the L2's answer, hits and misses of the L2 together)")
