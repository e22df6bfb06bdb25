#!/usr/bin/env python3
"""Diffs what the reference binary (oracle/_ref/ref_dump) produced for every fixture with the CPU oracle and writes pin_report.json:
per fixture and per array, equal / first mismatch.  Exit status 0 either way -- tests/test_pin_cpu.py turns the report into a verdict."""
import glob, json, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
import oracle_lib

fx, report_path = sys.argv[1], sys.argv[2]
orc = oracle_lib.Oracle()
report = {"fixtures": {}, "all_equal": True}
for pgm in sorted(glob.glob(os.path.join(fx, "*.pgm"))):
    stem = pgm[:-4]; name = os.path.basename(stem); nfeat = int(name.rsplit("_", 1)[1])
    with open(pgm, "rb") as f:
        assert f.readline().strip() == b"P5"; w, h = map(int, f.readline().split()); f.readline()
        img = np.frombuffer(f.read(), np.uint8).reshape(h, w)
    rd = lambda n, dt: np.fromfile(stem + "_" + n + ".bin", dtype=dt)
    kp, desc = orc.orb_extract(img, nfeat)
    kl, ld, fn, raw = orc.lines_extract(img, 40)
    exp = {"kp": kp.view(np.uint8).reshape(-1), "desc": desc.reshape(-1), "kl": kl.view(np.uint8).reshape(-1), "ldesc": ld.reshape(-1), "linefn": fn.reshape(-1).view(np.uint8)}
    res = {}
    for k, e in exp.items():
        got = rd(k, np.uint8)
        eq = got.size == e.size and bool(np.array_equal(got, e))
        res[k] = {"equal": eq, "reference_bytes": int(got.size), "oracle_bytes": int(e.size),
                  "first_mismatch_byte": None if eq else int(np.argmax(got[:min(got.size, e.size)] != e[:min(got.size, e.size)])) if min(got.size, e.size) else 0}
        report["all_equal"] &= eq
    report["fixtures"][name] = res
json.dump(report, open(report_path, "w"), indent=1)
print(json.dumps({k: {a: v["equal"] for a, v in r.items()} for k, r in report["fixtures"].items()}, indent=1))
print("oracle == reference binary on every fixture:", report["all_equal"])
