#!/usr/bin/env python3
"""Per-kernel sum of one rocprofv3 PMC counter (rocpd .db): name, launches, total KB, KB per launch."""
import sqlite3, sys, glob, os

def main(path, out=None):
    dbs = [path] if path.endswith('.db') else glob.glob(os.path.join(path, '**', '*.db'), recursive=True)
    agg = {}
    for db in dbs:
        c = sqlite3.connect(db)
        for name, cname, val in c.execute("select name, counter_name, counter_value from pmc_events"):
            short = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][-60:]
            d = agg.setdefault((short, cname), [0, 0.0])
            d[0] += 1; d[1] += val
    lines = ["%-62s %-12s %8s %16s %16s" % ("kernel", "counter", "launches", "total_KB", "KB_per_launch")]
    for (k, cn), v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("%-62s %-12s %8d %16.1f %16.1f" % (k, cn, v[0], v[1], v[1] / v[0]))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, 'w').write(txt + "\n")

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
