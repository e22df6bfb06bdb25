#!/usr/bin/env python3
"""Writes the fixture images of tests/golden/make_fixtures.py (the reference's real frame + seeded synthetic frames) as binary PGM files
for oracle/ref_pin/ref_dump: <outdir>/<name>_<nfeatures>.pgm"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
from synth import synth_frame, noise_frame

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
gold = os.path.join(HERE, "..", "..", "tests", "golden")
frames = {"icl_1000": np.load(os.path.join(gold, "icl_input_gray.npz"))["gray"], "icl_2000": np.load(os.path.join(gold, "icl_input_gray.npz"))["gray"],
          "synth1234_1000": synth_frame(1234), "synth2000_1000": synth_frame(2000), "synthsmall_500": synth_frame(4321, w=320, h=240),
          "noise_1000": noise_frame(7), "big_2000": synth_frame(1235, w=1280, h=960), "big_4000": synth_frame(1235, w=1280, h=960),
          # non-default extractor parameters (compare_stub.py PARAMS): 4 levels at scale 1.5; FAST thresholds 35 / 12
          "synth4lev15_700": synth_frame(99, w=480, h=360), "synthth_1200": synth_frame(4242),
          "wide_600": synth_frame(31, w=900, h=200), "portrait_600": synth_frame(32, w=360, h=480), "flat_500": np.full((240, 320), 128, np.uint8),
          "synth2001_1000": synth_frame(2001), "synth2002_1000": synth_frame(2002), "synth2003_2000": synth_frame(2003)}
for name, img in frames.items():
    with open(os.path.join(out, name + ".pgm"), "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0])); f.write(np.ascontiguousarray(img, np.uint8).tobytes())
print("wrote", len(frames), "fixtures to", out)
