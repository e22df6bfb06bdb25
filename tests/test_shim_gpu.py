"""GPU: the C++ drop-in surface (StructureSLAM::ORBextractor, LineSegment::ExtractLineSegment and the
matcher bodies) called the way Frame.cc / Tracking.cc call them, compared with the oracle."""
import os, subprocess, tempfile
import numpy as np
import pytest
import pkg
from synth import synth_frame, warp_prev, synthetic_vocab, write_vocab_text

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("blur_variant", [0, 1])
def test_cpp_shim_end_to_end(oracle, blur_variant):
    """blur_variant 1: the drop-in classes read SSLAM_ORB_BLUR_VARIANT in their constructors (OpenCV 3.4.0's 8-bit GaussianBlur for the rBRIEF and LBD
    pre-blurs, include/sslam_frontend.h); everything downstream (matches, distinctive descriptor, BoW) is then compared with the oracle under the same variant."""
    exe = pkg.builder().build_shim(force=False, verbose=False)
    prev_variant = oracle.set_gauss_variant(blur_variant)
    try:
        _shim_end_to_end(oracle, exe, blur_variant)
    finally:
        oracle.set_gauss_variant(prev_variant)


def _shim_end_to_end(oracle, exe, blur_variant):
    cur = synth_frame(1234); prev = warp_prev(cur)
    with tempfile.TemporaryDirectory() as d:
        cur.tofile(os.path.join(d, "cur.raw")); prev.tofile(os.path.join(d, "prev.raw"))
        out = os.path.join(d, "o")
        rng = np.random.default_rng(21)
        L, ptr, ch, nd, word, weight = synthetic_vocab(rng, k=8, L=3)
        vpath = os.path.join(d, "voc.txt"); write_vocab_text(vpath, 8, L, ptr, ch, nd, weight, weight_fmt="%.6g")
        ov = oracle.vocab_load_text(vpath)
        env = dict(os.environ)
        if blur_variant: env["SSLAM_ORB_BLUR_VARIANT"] = str(blur_variant)
        r = subprocess.run([exe, os.path.join(d, "cur.raw"), "640", "480", os.path.join(d, "prev.raw"), out, "1000", "40", vpath], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        rd = lambda n, dt: np.fromfile(out + "_" + n + ".bin", dtype=dt)
        meta = rd("meta", np.int32)
        kp = rd("kp", np.uint8).reshape(-1, 28); desc = rd("desc", np.uint8).reshape(-1, 32)
        kl = rd("kl", np.uint8).reshape(-1, 68); ldesc = rd("ldesc", np.uint8).reshape(-1, 32); fn = rd("fn", np.float64).reshape(-1, 3)
        m12 = rd("m12", np.int32); lm = rd("lm", np.int32).reshape(-1, 2)
        bow = rd("bow", np.float64).reshape(-1, 2); fvflat = rd("fv", np.int32)
    okp, odesc = oracle.orb_extract(cur, 1000)
    okl, old, ofn, _ = oracle.lines_extract(cur, 40)          # reference cap 40 (src/ExtractLineSegment.cpp:42)
    assert meta[0] == len(okp) and meta[1] == 40 and meta[4] == 1 and meta[5] == 8
    np.testing.assert_array_equal(kp, okp.view(np.uint8).reshape(-1, 28))
    np.testing.assert_array_equal(desc, odesc)
    ka = kl.copy().view(np.uint8); kb = okl.view(np.uint8).reshape(-1, 68).copy()
    ka[:, 0:4] = 0; kb[:, 0:4] = 0                               # KeyLine.angle: atan2 (<= 1 ulp), compared in test_lines_gpu
    np.testing.assert_array_equal(ka, kb)
    assert np.unpackbits(ldesc ^ old, axis=1).sum(axis=1).max() <= 8
    np.testing.assert_array_equal(fn, ofn)
    kp1, d1 = oracle.orb_extract(prev, 1000)
    pm = np.stack([kp1["x"], kp1["y"]], axis=1).astype(np.float32)
    om12, _, on = oracle.search_for_initialization(kp1, d1, okp, odesc, pm, 100, 0.9, True)
    assert meta[2] == on
    np.testing.assert_array_equal(m12, om12)
    l1 = oracle.lines_extract(prev, 40)
    opairs, _, _ = oracle.line_match(l1[1], old, 0.5, False)
    assert meta[3] == len(opairs)
    np.testing.assert_array_equal(lm, opairs)
    nobs = min(len(odesc), 37)                                   # MapPoint::ComputeDistinctiveDescriptors through the shim
    assert meta[6] == oracle.distinctive(odesc[:nobs], np.array([0, nobs], np.int32))[0]

    # Frame::ComputeBoW through sslam_shim::ORBVocabulary (text file -> device tree -> BowVector / FeatureVector)
    assert meta[7] == ov["nwords"]
    bw, bv, fn_, fp, ff = oracle.compute_bow(ov["levels"], ov["child_ptr"], ov["children"], ov["node_desc"], ov["word_id"], ov["weight"], odesc, 2)
    np.testing.assert_array_equal(bow[:, 0], bw.astype(np.float64)); np.testing.assert_array_equal(bow[:, 1], bv)
    exp = []
    for j in range(len(fn_)):
        exp += [int(fn_[j]), int(fp[j + 1] - fp[j])] + ff[fp[j]:fp[j + 1]].tolist()
    assert fvflat.tolist() == exp and len(bw) > 100


def test_matcher_class_dropins(oracle, fe):
    """shim/ORBmatcher.h / shim/LSDmatcher.h: the reference's class names and call syntax (member templates over the frame types) --
    ORBmatcher::SearchByProjection(F, vpMapPoints, th) as Tracking::SearchLocalPoints calls it, LSDmatcher::SearchForTriangulation,
    SearchByProjection(KF, F, ...) and SearchByDescriptor(KF, KF, ...) -- against the oracle's restatements of the reference bodies.
    (SearchForInitialization / SerachForInitialize through the classes are asserted equal to the free functions inside shim_test.)"""
    exe = pkg.builder().build_shim(force=False, verbose=False)
    cur = synth_frame(1234); prev = warp_prev(cur)
    with tempfile.TemporaryDirectory() as d:
        cur.tofile(os.path.join(d, "cur.raw")); prev.tofile(os.path.join(d, "prev.raw"))
        out = os.path.join(d, "o")
        r = subprocess.run([exe, os.path.join(d, "cur.raw"), "640", "480", os.path.join(d, "prev.raw"), out, "1000", "40"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        rd = lambda n: np.fromfile(out + "_" + n + ".bin", dtype=np.int32)
        own, tri, kf2f, kf2kf = rd("cls_local"), rd("cls_tri").reshape(-1, 2), rd("cls_kf2f"), rd("cls_kf2kf")
    kp1, d1 = oracle.orb_extract(prev, 1000); kp2, d2 = oracle.orb_extract(cur, 1000)
    scale = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    sc = [np.float32(1.0)]
    for _ in range(7): sc.append(np.float32(sc[-1] * np.float32(1.2)))
    # the queries Tracking::SearchLocalPoints' loop forms (src/ORBmatcher.cc:51-70) for the stand-in map points of shim_test
    idx = [i for i in range(len(kp1)) if i % 3 != 0 and i % 7 != 0]
    q = np.zeros(len(idx), fe.PQ_DTYPE)
    for j, i in enumerate(idx):
        lvl = int(kp1["octave"][i]); r0 = np.float32(2.5) if i % 2 else np.float32(4.0)
        q[j] = (np.float32(kp1["x"][i]) + np.float32(2.5), np.float32(kp1["y"][i]) - np.float32(1.5), 0, 0, np.float32(np.float32(r0 * np.float32(3.0)) * sc[lvl]), lvl - 1, lvl, 0.0, -1.0, 1, 1)
    occupied = np.array([1 if i % 10 == 0 else 0 for i in range(len(kp2))], np.uint8)
    res = oracle.search_by_projection(0, 0, kp2, d2, q, d1[idx], occupied=occupied, uright=np.full(len(kp2), -1, np.float32), nnratio=0.8, th_dist=100, check_orientation=False)
    assigned, n = res[0], res[1]
    exp = np.array([idx[a] if a >= 0 else (-2 if i % 10 == 0 else -3 if i % 5 == 0 else -1) for i, a in enumerate(assigned)], np.int32)
    assert own[-1] == n and n > 100
    np.testing.assert_array_equal(own[:-1], exp)
    l1 = oracle.lines_extract(prev, 40); l2 = oracle.lines_extract(cur, 40)
    p01, _, _ = oracle.line_match(l1[1], l2[1], 0.1, False)
    np.testing.assert_array_equal(tri, np.array([p for p in p01 if not (p[0] % 4 == 0 or p[1] % 6 == 1)], np.int32).reshape(-1, 2))
    pr, _, _ = oracle.line_match(l1[1], l2[1], 0.5, True)                       # keyframe -> frame: ratio gate, keyframe lines 0, 4, 8, ... hold map lines
    e = np.full(len(l2[0]), -1, np.int32)
    for a, b in pr:
        if a % 4 == 0: e[b] = a
    np.testing.assert_array_equal(kf2f[:-1], e); assert kf2f[-1] == sum(1 for a, b in pr if a % 4 == 0)
    p05, _, _ = oracle.line_match(l1[1], l2[1], 0.5, False)                     # keyframe -> keyframe: MAD gate, the SECOND keyframe's map lines
    e2 = np.full(len(l1[0]), -1, np.int32)
    for a, b in p05:
        if b % 6 == 1: e2[a] = len(l1[0]) + b
    np.testing.assert_array_equal(kf2kf, e2)


def test_tracking_call_class_dropins(oracle, fe):
    """shim/ORBmatcher.h / LSDmatcher.h: SearchByProjection(CurrentFrame, LastFrame, th, bMono) -- the call Tracking::TrackWithMotionModel makes for
    every tracked frame (src/Tracking.cc:1227, :1243; bodies src/ORBmatcher.cc:1331-1473, src/LSDmatcher.cpp:22-141) -- through stand-in frames
    with poses, intrinsics and map points / lines at known world positions.  The test forms the reference's queries itself (the projection loop
    in float32, same operation order) and asks the oracle's restatement of the window search; the frame's pointer arrays after the call must
    agree, including keypoints that were matched and then removed by the rotation check (NULL, not "unchanged")."""
    exe = pkg.builder().build_shim(force=False, verbose=False)
    cur = synth_frame(1234); prev = warp_prev(cur)
    with tempfile.TemporaryDirectory() as d:
        cur.tofile(os.path.join(d, "cur.raw")); prev.tofile(os.path.join(d, "prev.raw"))
        out = os.path.join(d, "o")
        r = subprocess.run([exe, os.path.join(d, "cur.raw"), "640", "480", os.path.join(d, "prev.raw"), out, "1000", "40"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        trk = np.fromfile(out + "_trk.bin", dtype=np.int32); wp = np.fromfile(out + "_trk_wp.bin", dtype=np.float32).reshape(-1, 3)
        wl = np.fromfile(out + "_trk_wl.bin", dtype=np.float64).reshape(-1, 6); cam = np.fromfile(out + "_trk_cam.bin", dtype=np.float32)
    f32 = np.float32
    fx, fy, cx, cy, mbf, mb = (f32(v) for v in cam[:6]); TL = cam[6:22].reshape(4, 4); TC = cam[22:38].reshape(4, 4)
    kp1, d1 = oracle.orb_extract(prev, 1000); kp2, d2 = oracle.orb_extract(cur, 1000)
    kl1, ld1 = oracle.lines_extract(prev, 40)[:2]; kl2, ld2 = oracle.lines_extract(cur, 40)[:2]
    sc = [f32(1.0)]
    for _ in range(7): sc.append(f32(sc[-1] * f32(1.2)))

    def rx_plus_t(T, x):          # gemm's row order in float32 (shim/FrontendMatchers.h RxPlusT, the stand-in path)
        return [f32(f32(f32(f32(T[r, 0] * x[0]) + f32(T[r, 1] * x[1])) + f32(T[r, 2] * x[2])) + T[r, 3]) for r in range(3)]
    twc = [f32(f32(f32(f32(-TC[0, r]) * TC[0, 3]) + f32(f32(-TC[1, r]) * TC[1, 3])) + f32(f32(-TC[2, r]) * TC[2, 3])) for r in range(3)]
    tlc = rx_plus_t(TL, twc)
    W, H, th = f32(640), f32(480), f32(15.0)
    pos = 0; pruned_seen = False
    for bMono in (True, False):
        fwd = (tlc[2] > mb) and not bMono; bwd = (-tlc[2] > mb) and not bMono
        assert fwd == (not bMono) and not bwd
        def levels(o):
            return (o, -1) if fwd else (0, o) if bwd else (o - 1, o + 1)
        # ---- points
        q = []; qd = []; owner = []
        for i in range(len(kp1)):
            if i % 4 == 0 or i % 9 == 0: continue                          # no map point / outlier
            xc, yc, zc = rx_plus_t(TC, wp[i])
            invzc = f32(1.0 / np.float64(zc))
            if invzc < 0: continue
            u = f32(f32(f32(fx * xc) * invzc) + cx); v = f32(f32(f32(fy * yc) * invzc) + cy)
            if u < 0 or u > W or v < 0 or v > H: continue
            o = int(kp1["octave"][i]); lo, hi = levels(o)
            q.append((u, v, 0, 0, f32(th * sc[o]), lo, hi, f32(kp1["angle"][i]), f32(u - f32(mbf * invzc)), 1, 0 if i % 11 == 0 else 1)); qd.append(d1[i]); owner.append(i)
        qa = np.array(q, fe.PQ_DTYPE)
        occupied = np.array([1 if i % 10 == 0 else 0 for i in range(len(kp2))], np.uint8)
        assigned, n = oracle.search_by_projection(0, 1, kp2, d2, qa, np.stack(qd), occupied=occupied, uright=np.full(len(kp2), -1, np.float32), nnratio=0.9, th_dist=100, check_orientation=True)[:2]
        exp = np.array([owner[a] if a >= 0 else -1 if a == -2 else (-2 if i % 10 == 0 else -3 if i % 5 == 0 else -1) for i, a in enumerate(assigned)], np.int32)
        got = trk[pos:pos + len(kp2)]; pos += len(kp2)
        np.testing.assert_array_equal(got, exp)
        assert trk[pos] == n and n > 50, (trk[pos], n); pos += 1
        pruned_seen = pruned_seen or bool((assigned == -2).any())
        # ---- lines
        q = []; qd = []; owner = []
        for i in range(len(kl1)):
            if i % 3 == 1 or i % 13 == 5 or i % 8 == 7: continue           # no map line / bad / outlier
            sp = [f32(wl[i, k]) for k in range(3)]; ep = [f32(wl[i, 3 + k]) for k in range(3)]
            s3 = rx_plus_t(TC, sp); e3 = rx_plus_t(TC, ep)
            if s3[2] < 0 or e3[2] < 0: continue
            iz1 = f32(f32(1.0) / s3[2]); u1 = f32(f32(f32(fx * s3[0]) * iz1) + cx); v1 = f32(f32(f32(fy * s3[1]) * iz1) + cy)
            if u1 < 0 or u1 > W or v1 < 0 or v1 > H: continue
            iz2 = f32(f32(1.0) / e3[2]); u2 = f32(f32(f32(fx * e3[0]) * iz2) + cx); v2 = f32(f32(f32(fy * e3[1]) * iz2) + cy)
            if u2 < 0 or u2 > W or v2 < 0 or v2 > H: continue
            o = int(kp1["octave"][i]); lo, hi = levels(o)                   # LastFrame.mvKeys[i].octave: the reference's own expression (src/LSDmatcher.cpp:80)
            q.append((u1, v1, u2, v2, f32(th * sc[o]), lo, hi, 0.0, 0.0, 1, 0 if i % 6 == 0 else 1)); qd.append(ld1[i]); owner.append(i)
        got = trk[pos:pos + len(kl2)]; pos += len(kl2)
        init = np.array([(-2 if i % 8 == 0 else -3 if i % 4 == 0 else -1) for i in range(len(kl2))], np.int32)
        if q:
            qa = np.array(q, fe.PQ_DTYPE)
            occ = np.array([1 if i % 8 == 0 else 0 for i in range(len(kl2))], np.uint8)
            assigned, n = oracle.search_by_projection(1, 0, kl2, ld2, qa, np.stack(qd), occupied=occ, uright=None, nnratio=0.6, th_dist=100, check_orientation=False)[:2]
            exp = np.array([owner[a] if a >= 0 else init[i] for i, a in enumerate(assigned)], np.int32)
        else:
            exp, n = init, 0
        np.testing.assert_array_equal(got, exp)
        assert trk[pos] == n; pos += 1
    assert pos == len(trk)
    assert pruned_seen, "no keypoint was matched and then removed by the rotation check: the NULL-vs-unchanged distinction went untested"
