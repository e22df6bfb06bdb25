"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5's aux hook): the checker every parity claim rests on
must not read out of bounds or lean on undefined behaviour.  The oracle's sources are rebuilt with -fsanitize=address,undefined into a
temporary library and a child interpreter (libasan preloaded) drives the extractors and matchers over frames that hit the edges: an odd
size, a wide one, noise, a constant image.  Skips when the host compiler has no sanitizer runtime."""
import glob, os, subprocess, sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CHILD = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import oracle_lib
from synth import synth_frame, noise_frame
o = oracle_lib.Oracle()
frames = [synth_frame(11, w=333, h=251), synth_frame(12, w=640, h=96), noise_frame(3, w=160, h=120), np.full((120, 160), 77, np.uint8), synth_frame(13, w=320, h=240)]
tot = 0
prev = None
for f in frames:
    kp, d = o.orb_extract(f, 500, 1.2, 8, 20, 7)
    kl, ld, fn, raw = o.lines_extract(f, 200)
    tot += len(kp) + len(kl)
    if prev is not None and len(kp) and len(prev[0]):
        pm = np.stack([prev[0]["x"], prev[0]["y"]], axis=1).astype(np.float32)
        o.search_for_initialization(prev[0], prev[1], kp, d, pm, 100, 0.9, True, (0.0, float(f.shape[1]), 0.0, float(f.shape[0])))
        o.knn2(prev[1], d)
        if len(ld) >= 2 and len(prev[2]):
            o.line_match(prev[2], ld, 0.5, False)
    prev = (kp, d, ld)
assert tot > 500, tot
print("sanitized oracle ok", tot)
"""


def test_oracle_under_asan_ubsan(tmp_path):
    lib = str(tmp_path / "liboracle_san.so")
    srcs = sorted(glob.glob(os.path.join(ROOT, "oracle", "*.cpp")))
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-shared", "-o", lib] + srcs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer-capable compiler here: " + r.stderr[-200:])
    asan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan.so not found")
    env = dict(os.environ, LD_PRELOAD=asan, SSLAM_ORACLE_LIB=lib, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", CHILD % HERE], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "sanitized oracle ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
