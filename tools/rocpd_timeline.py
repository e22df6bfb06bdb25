#!/usr/bin/env python3
"""Kernel TIMELINE of one step from a rocprofv3 --kernel-trace database (rocpd .db): every dispatch of the LAST complete step with its start / end relative to the
step's first kernel, the queue it ran on, and -- when a stand-alone table is given (tools/rocpd_summary.py output of a one-stream run) -- how much longer it took
than with the chip to itself.  A dispatch's `start` is when its packet began to execute, not when its first wave got a slot: a kernel that waits for wave slots
behind another queue's resident waves shows as a long row that ENDS about one stand-alone duration after it really began.

    python tools/rocpd_timeline.py <dir-or-db> [standalone_table.txt] [out.txt]
"""
import sqlite3, sys, glob, os


def short(name):
    return name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]


def standalone(path):
    t = {}
    if not path or not os.path.exists(path):
        return t
    for ln in open(path).read().splitlines()[1:]:
        p = ln.split()
        if len(p) >= 7:
            try:
                t[p[0]] = float(p[3]) / 1e3       # avg_us -> ms
            except ValueError:
                pass
    return t


def main(path, alone_path=None, out=None):
    dbs = [path] if path.endswith('.db') else glob.glob(os.path.join(path, '**', '*.db'), recursive=True)
    rows = []
    for db in dbs:
        c = sqlite3.connect(db)
        cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
        qcol = 'queue_id' if 'queue_id' in cols else ('queue' if 'queue' in cols else None)
        scol = 'stream_id' if 'stream_id' in cols else ('stream' if 'stream' in cols else None)
        sel = "select name, start, end, %s, %s from kernels" % (qcol or "0", scol or "0")
        rows += [(short(n), s, e, q, st) for n, s, e, q, st in c.execute(sel)]
    rows.sort(key=lambda r: r[1])
    # a step starts with k_zero_misc (the line branch's first launch); take the last one that is followed by a complete step
    starts = [i for i, r in enumerate(rows) if r[0] == 'k_zero_misc']
    if len(starts) < 2:
        print("no complete step in the trace"); return
    i0, i1 = starts[-2], starts[-1]
    # the point branch of the step may end after the next step's k_zero_misc was dispatched only if steps are not joined; they are
    step = rows[i0:i1]
    t0 = step[0][1]
    alone = standalone(alone_path)
    queues = sorted(set(r[3] for r in step))
    lines = ["one step of the two-stream schedule: %d dispatches, %.2f ms from the first start to the last end; queues %s" % (len(step), (max(r[2] for r in step) - t0) / 1e6, queues),
             "%-28s %5s %10s %10s %10s %12s %10s" % ("kernel", "queue", "start_ms", "end_ms", "dur_ms", "alone_ms", "excess_ms")]
    for n, s, e, q, st in step:
        d = (e - s) / 1e6
        if d < 0.05 and n not in alone:
            continue
        a = alone.get(n)
        lines.append("%-28s %5s %10.2f %10.2f %10.2f %12s %10s" % (n[-28:], q, (s - t0) / 1e6, (e - t0) / 1e6, d, "%.2f" % a if a is not None else "-", "%.2f" % (d - a) if a is not None else "-"))
    # per queue: busy intervals and gaps
    for q in queues:
        iv = sorted((s, e) for n, s, e, qq, st in step if qq == q)
        busy = sum(e - s for s, e in iv) / 1e6
        lines.append("queue %s: first start %.2f ms, last end %.2f ms, sum of dispatch durations %.2f ms" % (q, (iv[0][0] - t0) / 1e6, (iv[-1][1] - t0) / 1e6, busy))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, 'w').write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else None)
