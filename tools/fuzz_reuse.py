"""Handle-reuse parity sweep (run through gpurun): ONE OrbExtractor and ONE LineExtractor process a stream of random frames
(same size for a while, then another size), so any state left behind in the workspace by an earlier frame would surface as a
mismatch with the oracle.  usage: python tools/fuzz_reuse.py [n_frames] [seed]"""
import sys, time; sys.path.insert(0, 'tests')
import numpy as np
from synth import synth_frame, noise_frame


def main():
    import pkg, oracle_lib
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    fe = pkg.frontend(); ctx = fe.Context(0); orc = oracle_lib.Oracle()
    ox = fe.OrbExtractor(ctx, 1000); lx = fe.LineExtractor(ctx, 200)
    bad = []; t0 = time.time(); w, h = 640, 480
    for it in range(n):
        if it % 25 == 0: w, h = int(rng.integers(160, 900)), int(rng.integers(120, 700))
        seed = int(rng.integers(0, 1 << 30)); r = rng.random()
        if r < 0.1: img = noise_frame(seed, w, h)
        elif r < 0.2: img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)           # constant frame between busy ones
        else: img = synth_frame(seed, w, h, nshapes=int(rng.integers(1, 150)), nstrokes=int(rng.integers(0, 90)), noise=float(rng.choice([0.0, 2.0, 6.0])))
        kp, d = ox(img); okp, od = orc.orb_extract(img, 1000)
        if len(kp) != len(okp) or not np.array_equal(kp.view(np.uint8), okp.view(np.uint8)) or not np.array_equal(d, od): bad.append("ORB it %d %dx%d" % (it, w, h))
        kl, ld, fn = lx(img); raw = lx.debug_segments(0)
        okl, old, ofn, oraw = orc.lines_extract(img, 200)
        ok = raw.shape == oraw.shape and np.array_equal(raw, oraw) and len(kl) == len(okl) and np.array_equal(ld, old) and np.array_equal(fn, ofn)
        if ok:
            for f in kl.dtype.names:
                if f != "angle" and not np.array_equal(kl[f], okl[f]): ok = False
        if not ok: bad.append("lines it %d %dx%d" % (it, w, h))
    print("fuzz_reuse: %d frames through one pair of handles in %.1f s, %d mismatches" % (n, time.time() - t0, len(bad)))
    for b in bad[:20]: print("  MISMATCH", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
