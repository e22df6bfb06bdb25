"""GPU parity on the configurations BASELINE.json names that the per-stage suites did not reach on the HIP path:
configs[0] stand-in (the reference's only real frame, images/input.png, with the reference's own parameters: 1000 and -- during
initialisation, src/Tracking.cc:120 -- 2000 features, line cap 40, src/ExtractLineSegment.cpp:42), configs[3] (1280x960, 2000 kp /
400 lines) including the two matchers, LSDmatcher::SearchForTriangulation's 0.1 gate (src/LSDmatcher.cpp:396), and the stage tap of
the 7x7 Gaussian blur (src/ORBextractor.cc:1085-1086)."""
import os
import numpy as np
import pytest
from synth import synth_frame, warp_prev, noise_frame

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _icl():
    return np.load(os.path.join(GOLD, "icl_input_gray.npz"))["gray"]


def _eq_kp(kp, desc, okp, odesc):
    assert len(kp) == len(okp)
    np.testing.assert_array_equal(kp.view(np.uint8).reshape(-1, 28), okp.view(np.uint8).reshape(-1, 28))      # every field incl. angle bits
    np.testing.assert_array_equal(desc, odesc)


def _eq_lines(res, ores, raw=None, oraw=None):
    kl, ld, fn = res; okl, old, ofn = ores
    assert len(kl) == len(okl)
    a = kl.view(np.uint8).reshape(-1, 68).copy(); b = okl.view(np.uint8).reshape(-1, 68).copy()
    ang = np.abs(a[:, 0:4].copy().view(np.int32).astype(np.int64) - b[:, 0:4].copy().view(np.int32).astype(np.int64))
    assert ang.max(initial=0) <= 1                       # KeyLine.angle: atan2, stated tolerance 1 ulp (tests/test_lines_gpu.py)
    a[:, 0:4] = 0; b[:, 0:4] = 0
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ld, old)
    np.testing.assert_array_equal(fn, ofn)
    if raw is not None:
        np.testing.assert_array_equal(raw, oraw)


def test_icl_real_frame_orb_1000_vs_golden(fe, ctx, oracle):
    """configs[0] stand-in, ORBextractor.nFeatures 1000 (Examples/ICL.yaml:41): HIP == committed golden == oracle run now"""
    g = np.load(os.path.join(GOLD, "oracle_golden.npz"))
    img = _icl()
    ex = fe.OrbExtractor(ctx, 1000, 1.2, 8, 20, 7)
    kp, desc = ex(img)
    ex.close()
    np.testing.assert_array_equal(kp.view(np.uint8).reshape(-1, 28), g["icl_kp"])
    np.testing.assert_array_equal(desc, g["icl_desc"])
    _eq_kp(kp, desc, *oracle.orb_extract(img, 1000))


def test_icl_real_frame_orb_2000_init_extractor(fe, ctx, oracle):
    """mpIniORBextractor = 2*nFeatures (src/Tracking.cc:120) on the real frame"""
    img = _icl()
    ex = fe.OrbExtractor(ctx, 2000, 1.2, 8, 20, 7)
    kp, desc = ex(img)
    for l in range(8):
        np.testing.assert_array_equal(ex.debug_candidates(0, l), oracle.candidates(img, l, 2000), err_msg="FAST candidates level %d" % l)
    ex.close()
    _eq_kp(kp, desc, *oracle.orb_extract(img, 2000))
    assert len(kp) > 1500


def test_icl_real_frame_lines_cap40_vs_golden(fe, ctx, oracle):
    """the reference's hard cap of 40 lines (src/ExtractLineSegment.cpp:42) on the real frame"""
    g = np.load(os.path.join(GOLD, "oracle_golden.npz"))
    img = _icl()
    lx = fe.LineExtractor(ctx, 40)
    kl, ld, fn = lx(img)
    raw = lx.debug_segments(0)
    lx.close()
    np.testing.assert_array_equal(raw, g["icl_segs"])
    np.testing.assert_array_equal(ld, g["icl_ldesc"])
    np.testing.assert_array_equal(fn, g["icl_linefn"])
    a = kl.view(np.uint8).reshape(-1, 68).copy(); b = g["icl_kl"].copy()
    a[:, 0:4] = 0; b[:, 0:4] = 0
    np.testing.assert_array_equal(a, b)
    okl, old, ofn, oraw = oracle.lines_extract(img, 40)
    _eq_lines((kl, ld, fn), (okl, old, ofn), raw, oraw)
    assert len(kl) == 40


def test_icl_real_frame_lines_cap200(fe, ctx, oracle):
    img = _icl()
    lx = fe.LineExtractor(ctx, 200)
    res = lx(img); raw = lx.debug_segments(0)
    lx.close()
    okl, old, ofn, oraw = oracle.lines_extract(img, 200)
    _eq_lines(res, (okl, old, ofn), raw, oraw)


def test_config3_1280x960_extract_and_match(fe, ctx, oracle):
    """configs[3]: 1280x960, 2000 ORB + 400 lines, extract + SearchForInitialization + LSD knn-2/MAD match vs the previous frame --
    every stage against the oracle (the oracle needs a few seconds per frame at this size)."""
    w, h = 1280, 960
    cur = synth_frame(1235, w=w, h=h); prev = warp_prev(cur)
    ox = fe.OrbExtractor(ctx, 2000); lx = fe.LineExtractor(ctx, 400)
    kp1, d1 = ox(prev); kp2, d2 = ox(cur)
    l1 = lx(prev); l2 = lx(cur)
    ox.close(); lx.close()
    okp1, od1 = oracle.orb_extract(prev, 2000); okp2, od2 = oracle.orb_extract(cur, 2000)
    _eq_kp(kp1, d1, okp1, od1); _eq_kp(kp2, d2, okp2, od2)
    ol1 = oracle.lines_extract(prev, 400); ol2 = oracle.lines_extract(cur, 400)
    _eq_lines(l1, ol1[:3]); _eq_lines(l2, ol2[:3])
    assert len(l2[0]) == 400
    bounds = (0.0, float(w), 0.0, float(h))
    pm = np.stack([kp1["x"], kp1["y"]], axis=1).astype(np.float32)
    m12, pmo, n = ctx.search_for_initialization(kp1, d1, kp2, d2, pm, 100, 0.9, True, bounds)
    om12, opmo, on = oracle.search_for_initialization(okp1, od1, okp2, od2, pm, 100, 0.9, True, bounds)
    assert n == on and n > 50
    np.testing.assert_array_equal(m12, om12); np.testing.assert_array_equal(pmo, opmo)
    idx, dist = ctx.hamming_knn2(d1, d2)
    oi, odist = oracle.knn2(od1, od2)
    np.testing.assert_array_equal(idx, oi); np.testing.assert_array_equal(dist, odist)
    for gate, ratio in ((0.5, False), (0.1, False), (0.5, True)):
        pairs, mad, mad12 = ctx.line_match(l1[1], l2[1], gate, ratio)
        opairs, omad, omad12 = oracle.line_match(ol1[1], ol2[1], gate, ratio)
        np.testing.assert_array_equal(pairs, opairs)
        assert mad == omad and mad12 == omad12
        assert len(pairs) > 20


@pytest.mark.parametrize("n1,n2", [(40, 40), (200, 187), (400, 400), (3, 2), (1, 5)])
def test_line_match_triangulation_gate(ctx, oracle, n1, n2):
    """LSDmatcher::SearchForTriangulation accepts a pair when d2 - d1 > 0.1 * MAD12 (src/LSDmatcher.cpp:396,408)"""
    rng = np.random.default_rng(100 * n1 + n2)
    t = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
    q = t[rng.integers(0, n2, n1)].copy()
    flips = rng.integers(0, 256, (n1, 32), dtype=np.uint8) & rng.integers(0, 256, (n1, 32), dtype=np.uint8) & rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    q ^= flips                                           # noisy copies of train rows: a spread of nearest / second-nearest gaps
    pairs, mad, mad12 = ctx.line_match(q, t, 0.1, False)
    opairs, omad, omad12 = oracle.line_match(q, t, 0.1, False)
    np.testing.assert_array_equal(pairs, opairs)
    assert mad == omad and mad12 == omad12
    p5, _, _ = ctx.line_match(q, t, 0.5, False)
    assert len(p5) <= len(pairs)                         # the 0.5 gate of SerachForInitialize is the stricter one


def test_line_match_gate01_on_extracted_lines(fe, ctx, oracle):
    cur = synth_frame(2000); prev = warp_prev(cur)
    lx = fe.LineExtractor(ctx, 200)
    l1 = lx(prev); l2 = lx(cur)
    lx.close()
    pairs, mad, mad12 = ctx.line_match(l1[1], l2[1], 0.1, False)
    opairs, omad, omad12 = oracle.line_match(l1[1], l2[1], 0.1, False)
    np.testing.assert_array_equal(pairs, opairs)
    assert (mad, mad12) == (omad, omad12) and len(pairs) > 20


@pytest.mark.parametrize("make", [lambda: synth_frame(1234), lambda: noise_frame(11, w=320, h=240), _icl])
def test_blur_stage_tap(fe, ctx, oracle, make):
    """a7: the 7x7 sigma-2 blur is fused into the descriptor kernel; its two passes, tapped around every keypoint, must equal the
    oracle's GaussianBlur of the whole level (reflect-101 borders included: keypoints sit as close as 19 px to the level edge)"""
    img = make()
    ex = fe.OrbExtractor(ctx, 1000)
    kp, desc = ex(img)
    tkp, pat = ex.debug_blur_patches(0)
    np.testing.assert_array_equal(tkp.view(np.uint8), kp.view(np.uint8))
    scales = ex.scales()[0]
    levels = [oracle.blur7(ex.debug_level(0, l)) for l in range(8)]
    ex.close()
    near_edge = 0
    for i in range(len(kp)):
        l = int(kp["octave"][i])
        x = int(round(float(kp["x"][i]) / float(scales[l]))) if l else int(kp["x"][i])
        y = int(round(float(kp["y"][i]) / float(scales[l]))) if l else int(kp["y"][i])
        B = levels[l]
        hh, ww = B.shape
        ys = np.arange(y - 18, y + 19); xs = np.arange(x - 18, x + 19)
        assert ys.min() >= 0 and ys.max() < hh and xs.min() >= 0 and xs.max() < ww      # EDGE_THRESHOLD 19 keeps the window inside
        near_edge += int(x - 21 < 0 or y - 21 < 0 or x + 21 >= ww or y + 21 >= hh)
        np.testing.assert_array_equal(pat[i], B[np.ix_(ys, xs)], err_msg="keypoint %d level %d" % (i, l))
    assert len(kp) > 300 and near_edge > 0


def test_batch_overflow_is_reported(fe, ctx):
    """the batch entry points clamp rows to the caller's capacity; the status calls (and sslam_frontend_batch itself) must report what the
    single-frame calls report: SSLAM_ERR_CAPACITY for a frame with more keypoints / lines than the capacity"""
    import ctypes as C
    import torch
    L = fe.lib()
    frames = np.stack([synth_frame(7000 + i, 320, 240) for i in range(3)] + [np.full((240, 320), 90, np.uint8)])
    orb = fe.OrbExtractor(ctx, 500); lines = fe.LineExtractor(ctx, 100)
    out = fe.frontend_batch(orb, lines, frames)                    # full capacity: fine
    nk = [len(o[0]) for o in out]; nl = [len(o[2]) for o in out]
    assert max(nk) > 300 and nk[3] == 0 and max(nl) > 20
    tr, first = C.c_int(-1), C.c_int(-2)
    assert L.sslam_orb_batch_status(orb.h, orb.cap, None, C.byref(tr), C.byref(first)) == 0 and tr.value == 0 and first.value == -1
    small = max(nk) - 7
    rc = L.sslam_orb_batch_status(orb.h, small, None, C.byref(tr), C.byref(first))
    assert rc == -3 and tr.value == sum(k > small for k in nk) and first.value == [k > small for k in nk].index(True)
    un = C.c_int(-1)
    assert L.sslam_lines_batch_status(lines.h, 100, None, C.byref(tr), C.byref(un), C.byref(first)) == 0 and tr.value == 0 and un.value == 0
    lsmall = max(nl) - 1
    rc = L.sslam_lines_batch_status(lines.h, lsmall, None, C.byref(tr), C.byref(un), C.byref(first))
    assert rc == -3 and tr.value == sum(k > lsmall for k in nl) and un.value == 0
    # sslam_frontend_batch with a too-small capacity: the error comes back (rows are truncated, counts clamped)
    with pytest.raises(fe.SslamError, match="more lines than lcap"):
        fe.frontend_batch(orb, lines, frames, max_lines=lsmall)
    orb.close(); lines.close()
