"""Randomised GPU-vs-oracle parity sweep (run through gpurun): random sizes, shape densities, noise levels and extractor
parameters -- and, since round 5, random settings of the line path's selectable decisions (D11 nfa variant, D12 LBD bit order, D7 resize variant, D2 seed order) on both
sides; every ORB keypoint / descriptor, LSD segment, KeyLine field, LBD byte, line equation and knn-2 result is compared
with the CPU oracle exactly as tests/ does.  usage: python tools/fuzz_parity.py [n_frames] [seed]"""
import sys, time; sys.path.insert(0, 'tests')
import numpy as np
from synth import synth_frame, noise_frame


def cases(n, rng):
    for it in range(n):
        w = int(rng.integers(96, 900)); h = int(rng.integers(80, 700))
        if it % 3 == 0: w, h = (w // 20) * 20, (h // 5) * 5      # every third case a 5-to-4 geometry (w = 5m, 4 | m): the fused gradient kernel of round 6; the others take k_blur7 + k_lsd_grad
        seed = int(rng.integers(0, 1 << 30))
        kind = rng.random()
        if kind < 0.1: img = noise_frame(seed, w, h)
        else: img = synth_frame(seed, w, h, nshapes=int(rng.integers(2, 120)), nstrokes=int(rng.integers(0, 80)), noise=float(rng.choice([0.0, 1.0, 2.0, 4.0, 8.0])))
        nfeat = int(rng.choice([300, 1000, 2000])); nlev = int(rng.choice([4, 8])); sf = float(rng.choice([1.2, 1.2, 1.5]))
        ini, mn = (20, 7) if rng.random() < 0.8 else (int(rng.integers(10, 40)), int(rng.integers(3, 10)))
        cap = int(rng.choice([40, 200, 400]))
        # round 5: the line path's decisions -- mostly the defaults (1, 1, 0, 0), each alternative now and then, the host-sorted seed order rarely (15 ms per frame) -- and two
        # more scale factors, drawn from a generator of their own so that the frames and parameters of the earlier rounds' sweeps (and the cases tests/ name by index) stay
        drng = np.random.default_rng([seed, 5])
        dec = (int(drng.random() < 0.75), int(drng.random() < 0.75), int(drng.random() < 0.2), int(drng.random() < 0.06))
        if drng.random() < 0.15: sf = float(drng.choice([1.1, 2.0]))
        yield it, img, nfeat, nlev, sf, ini, mn, cap, dec


SETTERS = (("set_nfa_variant", "orc_set_lsd_nfa_variant"), ("set_lbd_bit_order", "orc_set_lbd_bit_order"), ("set_resize_variant", "orc_set_lsd_resize"), ("set_seed_order", "orc_set_lsd_seed_sort"))


def lines_both(lx, orc, img, cap, dec):
    """one line extraction on the library (handle lx) and on the oracle under the decisions `dec`; the oracle's process-wide settings are restored"""
    for (ls, os_), v in zip(SETTERS, dec): getattr(lx, ls)(v)
    kl, ld, fn = lx(img); raw = lx.debug_segments(0)
    olds = [getattr(orc.L, os_)(v) for (ls, os_), v in zip(SETTERS, dec)]
    try: okl, old, ofn, oraw = orc.lines_extract(img, cap)
    finally:
        for (ls, os_), v in zip(SETTERS, olds): getattr(orc.L, os_)(v)
    return (kl, ld, fn, raw), (okl, old, ofn, oraw)


def main():
    import pkg, oracle_lib
    from test_lines_gpu import _ulp_diff
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
    fe = pkg.frontend(); ctx = fe.Context(0); orc = oracle_lib.Oracle()
    bad = []; stats = dict(frames=0, kp=0, lines=0, lbd_bits=0, angle_ulp=0)
    t0 = time.time()
    ndec = np.zeros((4, 2), int)
    for it, img, nfeat, nlev, sf, ini, mn, cap, dec in cases(n_frames, rng):
        h, w = img.shape
        tag = "it %d %dx%d nfeat %d lev %d sf %.1f th %d/%d cap %d decisions %s" % (it, w, h, nfeat, nlev, sf, ini, mn, cap, dec)
        for k, v in enumerate(dec): ndec[k, v] += 1
        try:
            ox = fe.OrbExtractor(ctx, nfeat, sf, nlev, ini, mn)
            kp, d = ox(img); okp, od = orc.orb_extract(img, nfeat, sf, nlev, ini, mn); ox.close()
            if len(kp) != len(okp) or not np.array_equal(kp.view(np.uint8), okp.view(np.uint8)) or not np.array_equal(d, od): bad.append("ORB " + tag)
            lx = fe.LineExtractor(ctx, cap)
            (kl, ld, fn, raw), (okl, old, ofn, oraw) = lines_both(lx, orc, img, cap, dec); lx.close()
            if raw.shape != oraw.shape or not np.array_equal(raw, oraw): bad.append("LSD segments " + tag)
            elif len(kl) != len(okl): bad.append("KeyLine count " + tag)
            else:
                for f in kl.dtype.names:
                    if f == "angle":
                        u = int(_ulp_diff(kl[f], okl[f]).max(initial=0)); stats["angle_ulp"] = max(stats["angle_ulp"], u)
                        if u > 1: bad.append("KeyLine.angle %d ulp " % u + tag)
                    elif not np.array_equal(kl[f], okl[f]): bad.append("KeyLine." + f + " " + tag)
                ham = np.unpackbits(ld ^ old, axis=1).sum(axis=1) if len(ld) else np.zeros(0, int)
                same = kl["angle"].view(np.uint32) == okl["angle"].view(np.uint32)
                if (ham[same] > 0).any() or ham.max(initial=0) > 8: bad.append("LBD %s " % ham[ham > 0] + tag)
                stats["lbd_bits"] += int(ham.sum())
                if not np.array_equal(fn, ofn): bad.append("line equations " + tag)
            if len(kp) > 1:
                idx, dist = ctx.hamming_knn2(d, d[::-1].copy()); oi, odist = orc.knn2(d, d[::-1].copy())
                if not (np.array_equal(idx, oi) and np.array_equal(dist, odist)): bad.append("knn2 " + tag)
            stats["frames"] += 1; stats["kp"] += len(kp); stats["lines"] += len(kl)
        except Exception as e:
            bad.append("EXC %r %s" % (e, tag))
    print("fuzz_parity: %d frames, %d keypoints, %d lines compared in %.1f s; max KeyLine.angle diff %d ulp; LBD bits differing %d; %d mismatches"
          % (stats["frames"], stats["kp"], stats["lines"], time.time() - t0, stats["angle_ulp"], stats["lbd_bits"], len(bad)))
    print("line decisions drawn (value 0 / value 1): nfa %d / %d, LBD bit order %d / %d, resize %d / %d, seed order %d / %d" % tuple(ndec.reshape(-1)))
    for b in bad[:40]: print("  MISMATCH", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
