"""Inputs of tools/mix_check.c (CPU): frames of the shapes the GPU suite's line tests use (tests/test_lines_gpu.py, test_edge_gpu.py) -- sizes from 96x80 to 1280x960 (beyond
the LDS bitmap of the cluster form), noise, constant and low-contrast frames, line caps 40..400 -- with the CPU oracle's lines for each: tools/mix_frames.bin."""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, oracle_lib
from synth import synth_frame, noise_frame
orc = oracle_lib.Oracle()
rng = np.random.default_rng(99)
cases = [(synth_frame(2000), 200), (synth_frame(1235, w=1280, h=960), 400), (noise_frame(3, w=320, h=240), 200), (synth_frame(91, w=333, h=251), 40),
         (np.full((240, 320), 255, np.uint8), 40), (np.zeros((480, 640), np.uint8), 200), (synth_frame(77, w=800, h=600), 400), (noise_frame(5, w=320, h=240), 200),
         (synth_frame(5, w=96, h=80), 40), (synth_frame(6, w=900, h=120), 200), (synth_frame(7, w=131, h=577), 200), (synth_frame(1236, w=1280, h=960), 40),
         (synth_frame(8, w=1100, h=830), 400), (noise_frame(9, w=640, h=480), 200)]
for i in range(10):
    cases.append((synth_frame(int(rng.integers(1, 1 << 30)), w=int(rng.integers(96, 1000)), h=int(rng.integers(80, 760))), int(rng.choice([40, 200, 400]))))
with open("tools/mix_frames.bin", "wb") as f:
    f.write(np.int32(len(cases)).tobytes())
    for img, cap in cases:
        img = np.ascontiguousarray(img); h, w = img.shape
        kl, ld, fn, raw = orc.lines_extract(img, cap)
        f.write(np.array([w, h, cap, len(kl)], np.int32).tobytes()); f.write(img.tobytes())
        f.write(np.ascontiguousarray(kl).tobytes()); f.write(np.ascontiguousarray(ld).tobytes()); f.write(np.ascontiguousarray(fn).tobytes())
        print(w, h, cap, len(kl))
