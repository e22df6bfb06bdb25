"""Inputs of tools/lat_check.c (CPU): the bench's 64 varied 640x480 frames and the CPU oracle's lines for them (max 200 lines, as bench.py's latency leg);
or `lat_check_prepare.py 1280 960 8 400`: configs[3]'s frames -> tools/lat_frames_1280x960.raw, tools/lat_expected_1280x960.bin."""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, bench, oracle_lib
w, h, n, lines = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (640, 480, 64, 200)
tag = "" if (w, h) == (640, 480) else "_%dx%d" % (w, h)
cur, prev = bench.synth_frames(w, h, n, 0)
np.ascontiguousarray(np.stack(cur)).tofile("tools/lat_frames%s.raw" % tag)
orc = oracle_lib.Oracle()
with open("tools/lat_expected%s.bin" % tag, "wb") as f:
    for img in cur:
        kl, ld, fn, raw = orc.lines_extract(img, lines)
        f.write(np.int32(len(kl)).tobytes()); f.write(np.ascontiguousarray(kl).tobytes()); f.write(np.ascontiguousarray(ld).tobytes()); f.write(np.ascontiguousarray(fn).tobytes())
print("frames", len(cur))
