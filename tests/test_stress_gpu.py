"""Race / fuzz evidence inside the driver-run GPU suite (rounds 1-3 kept it in builder-run tools and text files under profiles/).

* the cluster form of the LSD core (lsd_cluster.h) is a lock-free protocol between a main wave, a feeder wave and up to 27 helper waves on several
  compute units: the same frames are extracted over and over for a fixed time budget -- every schedule differs -- alone and with a batch of ORB
  extraction running on another stream (helper workgroups then start late or not at all), and every extraction must equal the oracle; no bounded wait
  may expire;
* 150 random frames (sizes, aspect ratios, densities, noise, extractor parameters: tools/fuzz_parity.py's generator with its own seed) through both
  extractors and the dense matcher against the oracle."""
import ctypes as C, os, sys, time
import numpy as np
import pytest
import torch
from synth import synth_frame, noise_frame

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


def test_cluster_form_stress_time_boxed(fe, ctx, oracle):
    budget_s = float(os.environ.get("SSLAM_STRESS_SECONDS", "60"))
    frames = [synth_frame(2000 + i * 7, nshapes=ns, nstrokes=nst, noise=nz) for i, (ns, nst, nz) in enumerate([(60, 40, 2.0), (90, 60, 2.0), (35, 25, 3.5)])]
    frames += [synth_frame(1235, w=1280, h=960), noise_frame(3, w=320, h=240)]
    want = [oracle.lines_extract(f, 400) for f in frames]
    lx = fe.LineExtractor(ctx, 400); orb = fe.OrbExtractor(ctx, 1000)
    busy = torch.from_numpy(np.stack([synth_frame(3000 + i) for i in range(16)])).cuda().repeat(4, 1, 1).contiguous()
    d_kp = torch.zeros(64 * orb.cap * 28, dtype=torch.uint8, device="cuda"); d_desc = torch.zeros(64 * orb.cap * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(64, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    bad = []; total = 0; noisy_runs = 0; expired = 0
    t0 = time.time(); r = 0
    try:
        while time.time() - t0 < budget_s:
            fi = r % len(frames); noisy = (r // len(frames)) % 3 == 2
            if noisy:
                with torch.cuda.stream(side):
                    orb.extract_batch_dev(busy, 640, 480, 640, 640 * 480, 64, d_kp, d_desc, d_n, orb.cap, side.cuda_stream)
                noisy_runs += 1
            kl, ld, fn = lx(frames[fi])
            okl, old, ofn, oraw = want[fi]
            if not (np.array_equal(lx.debug_segments(0), oraw) and np.array_equal(ld, old) and np.array_equal(fn, ofn)):
                bad.append((r, fi, noisy))
            out = (C.c_longlong * 8)(); fe.lib().sslam_lines_debug_cycles(lx.h, 0, out); expired += int(out[7])
            total += 1; r += 1
        side.synchronize()
    finally:
        lx.close(); orb.close()
    print("cluster stress: %d extractions (%d under a concurrent ORB batch) in %.1f s, %d mismatches, %d bounded waits expired" % (total, noisy_runs, time.time() - t0, len(bad), expired))
    assert not bad, bad[:10]
    assert expired == 0
    assert total >= 200 and noisy_runs >= 50, (total, noisy_runs)


def test_fuzz_parity_150(fe, ctx, oracle):
    from fuzz_parity import cases, lines_both
    from test_lines_gpu import _ulp_diff
    rng = np.random.default_rng(20260926)
    bad = []; nkp = nl = 0
    for it, img, nfeat, nlev, sf, ini, mn, cap, dec in cases(150, rng):
        tag = "case %d %dx%d nfeat %d lev %d sf %.1f th %d/%d cap %d" % (it, img.shape[1], img.shape[0], nfeat, nlev, sf, ini, mn, cap)
        ox = fe.OrbExtractor(ctx, nfeat, sf, nlev, ini, mn)
        kp, d = ox(img); ox.close()
        okp, od = oracle.orb_extract(img, nfeat, sf, nlev, ini, mn)
        if len(kp) != len(okp) or not np.array_equal(kp.view(np.uint8), okp.view(np.uint8)) or not np.array_equal(d, od): bad.append("ORB " + tag)
        lx = fe.LineExtractor(ctx, cap)
        (kl, ld, fn, raw), (okl, old, ofn, oraw) = lines_both(lx, oracle, img, cap, dec); lx.close()
        if raw.shape != oraw.shape or not np.array_equal(raw, oraw): bad.append("LSD segments " + tag)
        elif len(kl) != len(okl): bad.append("KeyLine count " + tag)
        else:
            for f in kl.dtype.names:
                if f == "angle":
                    if int(_ulp_diff(kl[f], okl[f]).max(initial=0)) > 1: bad.append("KeyLine.angle " + tag)
                elif not np.array_equal(kl[f], okl[f]): bad.append("KeyLine." + f + " " + tag)
            same = kl["angle"].view(np.uint32) == okl["angle"].view(np.uint32)
            if not np.array_equal(ld[same], old[same]): bad.append("LBD " + tag)
            if not np.array_equal(fn, ofn): bad.append("line equations " + tag)
        if len(kp) > 1:
            idx, dist = ctx.hamming_knn2(d, d[::-1].copy()); oi, odist = oracle.knn2(d, d[::-1].copy())
            if not (np.array_equal(idx, oi) and np.array_equal(dist, odist)): bad.append("knn2 " + tag)
        nkp += len(kp); nl += len(kl)
    print("fuzz parity: 150 frames, %d keypoints, %d lines compared, %d mismatches" % (nkp, nl, len(bad)))
    assert not bad, bad[:10]
    assert nkp > 50000 and nl > 5000
