"""GPU: the C++ drop-in surface (StructureSLAM::ORBextractor, LineSegment::ExtractLineSegment and the
matcher bodies) called the way Frame.cc / Tracking.cc call them, compared with the oracle."""
import os, subprocess, tempfile
import numpy as np
import pytest
import pkg
from synth import synth_frame, warp_prev, synthetic_vocab, write_vocab_text

pytestmark = pytest.mark.gpu


def test_cpp_shim_end_to_end(oracle):
    exe = pkg.builder().build_shim(force=False, verbose=False)
    cur = synth_frame(1234); prev = warp_prev(cur)
    with tempfile.TemporaryDirectory() as d:
        cur.tofile(os.path.join(d, "cur.raw")); prev.tofile(os.path.join(d, "prev.raw"))
        out = os.path.join(d, "o")
        rng = np.random.default_rng(21)
        L, ptr, ch, nd, word, weight = synthetic_vocab(rng, k=8, L=3)
        vpath = os.path.join(d, "voc.txt"); write_vocab_text(vpath, 8, L, ptr, ch, nd, weight, weight_fmt="%.6g")
        ov = oracle.vocab_load_text(vpath)
        r = subprocess.run([exe, os.path.join(d, "cur.raw"), "640", "480", os.path.join(d, "prev.raw"), out, "1000", "40", vpath], capture_output=True, text=True, timeout=300)
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
