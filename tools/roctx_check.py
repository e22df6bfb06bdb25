#!/usr/bin/env python3
"""Lists what a `SSLAM_ROCTX=1 rocprofv3 --marker-trace ...` run recorded: tables of the rocpd database that hold marker regions, and per
range name the count and the mean host-side duration.  usage: roctx_check.py <rocprofv3 output dir> [out.txt]"""
import sqlite3, sys, glob, os
path = sys.argv[1]
out = []
for db in glob.glob(os.path.join(path, '**', '*.db'), recursive=True):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    cand = [t for t in tabs if 'region' in t.lower() or 'marker' in t.lower()]
    out.append("tables with regions / markers: " + ", ".join(cand))
    for t in cand:
        cols = [r[1] for r in c.execute("pragma table_info(%s)" % t)]
        out.append("  %s: %s" % (t, ", ".join(cols)))
        ncol = 'name' if 'name' in cols else next((x for x in cols if 'name' in x.lower()), None)
        if ncol and 'start' in cols and 'end' in cols:
            rows = {}
            for name, s, e in c.execute("select %s, start, end from %s" % (ncol, t)):
                if not str(name).startswith('k_'): continue
                d = rows.setdefault(name, [0, 0.0]); d[0] += 1; d[1] += (e - s) / 1e3
            if rows:
                out.append("%s: %d range names" % (t, len(rows)))
                for n, v in sorted(rows.items(), key=lambda kv: -kv[1][0])[:40]:
                    out.append("  %-24s %6d ranges  mean %8.2f us (host side: push .. pop around the launch)" % (n, v[0], v[1] / v[0]))
                break
for db in glob.glob(os.path.join(path, '**', '*.db'), recursive=True):
    c = sqlite3.connect(db)
    try:
        out.append("regions by category / name:")
        for cat, name, n in c.execute("select category, name, count(*) from regions group by category, name order by count(*) desc limit 12"):
            out.append("  %-28s %-40s %d" % (cat, str(name)[:40], n))
        out.append("marker ranges by message (extdata):")
        for ext, n, mean in c.execute("select extdata, count(*), avg(end - start) from regions where category like 'MARKER%' group by extdata order by count(*) desc limit 40"):
            out.append("  %-48s %6d ranges  mean %8.2f us host side" % (str(ext)[:48], n, mean / 1e3))
        out.append("region_args (name = value), most frequent:")
        for name, value, n in c.execute("select name, value, count(*) from region_args group by name, value order by count(*) desc limit 40"):
            out.append("  %-16s %-32s %d" % (name, str(value)[:32], n))
    except Exception as e:      # noqa: BLE001
        out.append("query failed: %s" % e)
txt = "\n".join(out); print(txt)
if len(sys.argv) > 2: open(sys.argv[2], 'w').write(txt + "\n")
