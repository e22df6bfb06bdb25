"""Inputs of tools/lat_check.c (CPU): the bench's 64 varied 640x480 frames and the CPU oracle's lines for them (max 200 lines, as bench.py's latency leg)."""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, bench, oracle_lib
cur, prev = bench.synth_frames(640, 480, 64, 0)
np.ascontiguousarray(np.stack(cur)).tofile("tools/lat_frames.raw")
orc = oracle_lib.Oracle()
with open("tools/lat_expected.bin", "wb") as f:
    for img in cur:
        kl, ld, fn, raw = orc.lines_extract(img, 200)
        f.write(np.int32(len(kl)).tobytes()); f.write(np.ascontiguousarray(kl).tobytes()); f.write(np.ascontiguousarray(ld).tobytes()); f.write(np.ascontiguousarray(fn).tobytes())
print("frames", len(cur))
