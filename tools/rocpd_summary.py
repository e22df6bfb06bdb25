#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace into a per-kernel stats table
(name, calls, total ms, avg us, min us, max us, % of GPU kernel time)."""
import sqlite3, sys, glob, os

def main(path, out=None):
    dbs = [path] if path.endswith('.db') else glob.glob(os.path.join(path, '**', '*.db'), recursive=True)
    rows = {}
    for db in dbs:
        c = sqlite3.connect(db)
        cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
        name_col = 'name' if 'name' in cols else cols[0]
        for name, start, end in c.execute("select %s, start, end from kernels" % name_col):
            d = rows.setdefault(name, [0, 0.0, 1e30, 0.0])
            dur = (end - start) / 1e3
            d[0] += 1; d[1] += dur; d[2] = min(d[2], dur); d[3] = max(d[3], dur)
    tot = sum(v[1] for v in rows.values()) or 1.0
    lines = ["%-70s %8s %12s %12s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct")]
    for name, v in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        short = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][-70:]
        lines.append("%-70s %8d %12.3f %12.2f %12.2f %12.2f %6.2f%%" % (short, v[0], v[1] / 1e3, v[1] / v[0], v[2], v[3], 100 * v[1] / tot))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, 'w').write(txt + "\n")

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
