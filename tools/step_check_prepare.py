"""Inputs of tools/step_check.c (CPU): the bench's 64 current / previous frames (tools/step_frames.raw: cur then prev) and the CPU oracle's ORB keypoints + descriptors
for the first 16 current frames (tools/step_expected_orb.bin); the lines of all 64 are tools/lat_expected.bin (tools/lat_check_prepare.py)."""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, bench, oracle_lib
cur, prev = bench.synth_frames(640, 480, 64, 0)
with open("tools/step_frames.raw", "wb") as f:
    f.write(np.ascontiguousarray(np.stack(cur)).tobytes()); f.write(np.ascontiguousarray(np.stack(prev)).tobytes())
orc = oracle_lib.Oracle()
with open("tools/step_expected_orb.bin", "wb") as f:
    for img in cur[:16]:
        kp, d = orc.orb_extract(img, 1000)
        f.write(np.int32(len(kp)).tobytes()); f.write(np.ascontiguousarray(kp).tobytes()); f.write(np.ascontiguousarray(d).tobytes())
if not os.path.exists("tools/lat_expected.bin"):
    os.system(sys.executable + " tools/lat_check_prepare.py")
print("ok")
