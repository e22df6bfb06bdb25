#!/usr/bin/env python3
"""A larger draw for `make -C oracle/ref_pin pin-stub`'s first half: N random frames (170..960 x 170..720, four kinds: shapes, noise, low-contrast shapes that need the
minThFAST pass, 8x8 blocks with tied responses on a lattice; 100..3000 features) written as fixtures into a scratch directory and compared by compare_stub.py --
the reference's own src/ORBextractor.cc (stub cv:: layer) against the CPU oracle, byte by byte, plus the four error-bar builds.

    campaign_orb.py <oracle/_ref> <scratch dir> <report.json> [N=120] [seed=4242] [--params]

Frames the reference itself cannot run are left out by construction: width >= 0.8 x height (it divides by nIni = round(W / H) = 0 on images taller than 2:1,
src/ORBextractor.cc:543-568: SIGSEGV) and both sides >= 170 (a pyramid level smaller than its border makes the cell grid negative: std::length_error, e.g. 524 x 79)."""
import os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
from synth import synth_frame, noise_frame

ref, out, report = sys.argv[1], sys.argv[2], sys.argv[3]
vary = "--params" in sys.argv      # also draw scaleFactor / nlevels / the FAST thresholds (levels kept large enough for the reference to run: see above)
if vary: sys.argv.remove("--params")
n = int(sys.argv[4]) if len(sys.argv) > 4 else 120
rng = np.random.default_rng(int(sys.argv[5]) if len(sys.argv) > 5 else 4242)
os.makedirs(out, exist_ok=True)
for i in range(n):
    h = int(rng.integers(170, 720)); w = int(rng.integers(max(170, int(0.8 * h)), 960)); kind = i % 4
    if kind == 0: img = synth_frame(int(rng.integers(1, 1 << 30)), w=w, h=h)
    elif kind == 1: img = noise_frame(int(rng.integers(1, 1 << 30)), w=w, h=h)
    elif kind == 2:
        img = synth_frame(int(rng.integers(1, 1 << 30)), w=w, h=h).astype(np.int32)
        img = (img * int(rng.integers(20, 100)) // 100 + int(rng.integers(0, 100))).clip(0, 255).astype(np.uint8)
    else:
        img = np.kron(rng.integers(0, 256, (h // 8 + 1, w // 8 + 1), dtype=np.uint8), np.ones((8, 8), np.uint8))[:h, :w].copy()
    nf = int(rng.choice([100, 300, 500, 1000, 1500, 2000, 3000]))
    fname = "c%03d_%d.pgm" % (i, nf)
    if vary:
        for _ in range(100):
            sf = float(rng.choice([1.1, 1.2, 1.3, 1.5, 2.0])); nl = int(rng.choice([2, 3, 4, 6, 8, 10])); ini = int(rng.choice([10, 20, 35, 50])); mn = int(rng.choice([3, 5, 7, 12]))
            if min(w, h) / sf ** (nl - 1) >= 60 and mn <= ini: break
        else:
            sf, nl, ini, mn = 1.2, 8, 20, 7
        fname = "p%04d_%d_%d_%d_%d_%d.pgm" % (i, nf, int(round(sf * 100)), nl, ini, mn)
    with open(os.path.join(out, fname), "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (w, h)); f.write(np.ascontiguousarray(img).tobytes())
sys.exit(subprocess.call([sys.executable, os.path.join(HERE, "compare_stub.py"), ref, out, report]))
