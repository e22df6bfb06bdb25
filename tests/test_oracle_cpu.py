"""CPU suite (no GPU): known-answer tests of the oracle's leaf math, the oracle against the
committed golden fixtures, and host-side helpers.  The reference ships no tests of its own
(SURVEY.md §4), so the KATs are the hand-derivable ones listed there."""
import os
import numpy as np
import pytest
from synth import synth_frame, warp_prev, const_frame

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_popcount_distance(oracle):
    z = np.zeros(32, np.uint8); o = np.full(32, 255, np.uint8)
    assert oracle.descriptor_distance(z, o) == 256          # src/ORBmatcher.cc:1650-1666
    assert oracle.descriptor_distance(z, z) == 0
    a = np.zeros(32, np.uint8); a[5] = 0b10110001
    assert oracle.descriptor_distance(a, z) == 4
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (50, 32), dtype=np.uint8); y = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    ref = np.unpackbits(x ^ y, axis=1).sum(axis=1)
    assert [oracle.descriptor_distance(x[i], y[i]) for i in range(50)] == list(ref)


def test_orb_constructor_tables(oracle):
    scale, per_level, umax = oracle.orb_params(1000, 1.2, 8)
    assert list(umax) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]       # src/ORBextractor.cc:454-469
    assert list(per_level) == [217, 181, 151, 126, 105, 87, 73, 60]                         # :436-446
    np.testing.assert_allclose(scale, 1.2 ** np.arange(8), rtol=1e-6)
    _, per_level2, _ = oracle.orb_params(2000, 1.2, 8)
    assert list(per_level2) == [434, 362, 302, 251, 209, 175, 145, 122]


def test_rbrief_pattern_table():
    inc = os.path.join(os.path.dirname(GOLD), "..", "oracle", "orb_pattern.inc")
    txt = open(inc).read().split("*/", 1)[1]
    nums = np.array([int(t) for t in txt.replace("\n", "").split(",") if t.strip()])
    assert nums.size == 1024                                      # 512 points, src/ORBextractor.cc:150-408
    assert list(nums[:4]) == [8, -3, 9, 5] and list(nums[-4:]) == [-1, -6, 0, -11]
    assert nums.min() == -13 and nums.max() == 13 or (nums.min() >= -13 and nums.max() <= 13)
    r = np.sqrt((nums.reshape(-1, 2) ** 2).sum(axis=1)).max()
    assert r < 19                                                 # taps stay inside EDGE_THRESHOLD
    prod = os.path.join(os.path.dirname(GOLD), "..", "structure-slam-pointline_amd", "csrc", "orb_pattern.inc")
    assert open(prod).read() == open(inc).read()
    # scikit-image carries its own transcription of the same OpenCV table (fixture from tests/golden/make_skimage_fixtures.py)
    sk = np.load(os.path.join(GOLD, "skimage_fast_orient.npz"))["orb_positions"]
    assert sk.shape == (256, 4) and (sk.astype(np.int64).reshape(-1) == nums).all()


def test_fast_atan2_axes(oracle):
    assert oracle.fast_atan2(0, 1) == 0.0
    assert abs(oracle.fast_atan2(1, 0) - 90.0) < 1e-4
    assert abs(oracle.fast_atan2(0, -1) - 180.0) < 1e-4
    assert abs(oracle.fast_atan2(-1, 0) - 270.0) < 1e-4
    assert abs(oracle.fast_atan2(1, 1) - 45.0) < 0.02
    for y, x in [(3, 4), (-2, 7), (5, -1), (-9, -9)]:
        assert abs(oracle.fast_atan2(y, x) - (np.degrees(np.arctan2(y, x)) % 360)) < 0.05     # accuracy ~0.3 deg worst case


def test_reflect101(oracle):
    L = oracle.L
    assert [L.orc_reflect101(i, 5) for i in (-2, -1, 0, 4, 5, 6)] == [2, 1, 0, 4, 3, 2]


def test_gauss_taps(oracle):
    assert list(oracle.gauss_taps(7, 2.0)) == [18, 34, 48, 56, 48, 34, 18]      # 8.8 fixed point, sum == 256 (decision D6)
    assert sum(oracle.gauss_taps(7, 0.75)) == 256 and sum(oracle.gauss_taps(5, 1.0)) == 256
    # decision D6's selectable alternative: OpenCV 3.4.0 rounds every tap of the float kernel (sum 257) and saturates
    try:
        oracle.set_gauss_variant(1)
        assert list(oracle.gauss_taps(7, 2.0)) == [18, 34, 49, 55, 49, 34, 18]
        assert (oracle.blur7(np.full((40, 50), 255, np.uint8)) == 255).all() and (oracle.blur7(np.full((40, 50), 100, np.uint8)) == 101).all()      # 100 * 257^2 / 65536 = 100.78
    finally:
        assert oracle.set_gauss_variant(0) == 1
    assert (oracle.blur7(np.full((40, 50), 100, np.uint8)) == 100).all()


def test_blur_constant_and_impulse(oracle):
    c = np.full((20, 30), 77, np.uint8)
    np.testing.assert_array_equal(oracle.blur7(c), c)             # taps sum to exactly 1.0
    imp = np.zeros((21, 21), np.uint8); imp[10, 10] = 255
    b = oracle.blur7(imp)
    t = np.array([18, 34, 48, 56, 48, 34, 18])
    exp = ((np.outer(t, t) * 255 + 32768) >> 16).astype(np.uint8)
    np.testing.assert_array_equal(b[7:14, 7:14], exp)


def test_fast_score_bright_arc(oracle):
    p = np.full((7, 7), 100, np.uint8)
    assert oracle.fast_score(p) <= 0                               # flat patch: no corner
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    q = p.copy()
    for k in range(9):                                             # 9 contiguous ring pixels brighter by 50
        dx, dy = ring[k]; q[3 + dy, 3 + dx] = 150
    assert oracle.fast_score(q) == 49                              # score = arc contrast - 1
    q2 = p.copy()
    for k in range(8):
        dx, dy = ring[k]; q2[3 + dy, 3 + dx] = 150
    assert oracle.fast_score(q2) <= 0                              # 8 contiguous is not FAST-9


def test_oracle_against_golden():
    import oracle_lib
    orc = oracle_lib.Oracle()
    g = np.load(os.path.join(GOLD, "oracle_golden.npz"))
    icl = np.load(os.path.join(GOLD, "icl_input_gray.npz"))["gray"]
    frames = {"icl": (icl, 1000, 40), "synth1234": (synth_frame(1234), 1000, 200), "synth_small": (synth_frame(4321, w=320, h=240), 500, 200)}
    import hashlib
    for name, (img, nfeat, nln) in frames.items():
        assert hashlib.sha256(img.tobytes()).digest() == g[name + "_sha"].tobytes(), "fixture image bytes changed: " + name
        kp, desc = orc.orb_extract(img, nfeat)
        np.testing.assert_array_equal(kp.view(np.uint8).reshape(len(kp), 28), g[name + "_kp"])
        np.testing.assert_array_equal(desc, g[name + "_desc"])
        kl, ld, fn, raw = orc.lines_extract(img, nln)
        np.testing.assert_array_equal(raw, g[name + "_segs"])
        np.testing.assert_array_equal(kl.view(np.uint8).reshape(len(kl), 68), g[name + "_kl"])
        np.testing.assert_array_equal(ld, g[name + "_ldesc"])
        np.testing.assert_array_equal(fn, g[name + "_linefn"])


def test_oracle_match_golden(oracle):
    g = np.load(os.path.join(GOLD, "oracle_golden.npz"))
    cur = synth_frame(1234); prev = warp_prev(cur)
    kp1, d1 = oracle.orb_extract(prev, 1000); kp2, d2 = oracle.orb_extract(cur, 1000)
    pm = np.stack([kp1["x"], kp1["y"]], axis=1).astype(np.float32)
    m12, pmo, n = oracle.search_for_initialization(kp1, d1, kp2, d2, pm, 100, 0.9, True)
    np.testing.assert_array_equal(m12, g["match_m12"]); assert n == int(g["match_n"][0]) and n > 50
    assert n == int((m12 >= 0).sum())
    # every surviving match respects TH_LOW and the window
    for i in np.nonzero(m12 >= 0)[0]:
        assert oracle.descriptor_distance(d1[i], d2[m12[i]]) <= 50
        assert abs(kp2["x"][m12[i]] - kp1["x"][i]) < 100 and abs(kp2["y"][m12[i]] - kp1["y"][i]) < 100
    l1 = oracle.lines_extract(prev, 200); l2 = oracle.lines_extract(cur, 200)
    pairs, mad, mad12 = oracle.line_match(l1[1], l2[1], 0.5, False)
    np.testing.assert_array_equal(pairs, g["lmatch_pairs"])
    assert [mad, mad12] == list(g["lmatch_mad"])


def test_orb_structure_properties(oracle):
    img = synth_frame(55)
    kp, desc = oracle.orb_extract(img, 1000)
    assert 900 <= len(kp) <= 1016 and desc.shape == (len(kp), 32)
    assert (np.diff(kp["octave"]) >= 0).all()                      # levels concatenated 0..7 (:1075-1103)
    sizes = {0: 31, 1: 37, 2: 44, 3: 53, 4: 64, 5: 77, 6: 92, 7: 111}
    assert all(kp["size"][i] == sizes[int(kp["octave"][i])] for i in range(len(kp)))
    assert ((kp["angle"] >= 0) & (kp["angle"] < 360)).all() and (kp["class_id"] == -1).all()
    l0 = kp[kp["octave"] == 0]
    assert (l0["x"] >= 19).all() and (l0["x"] < 640 - 19).all() and (l0["y"] >= 19).all() and (l0["y"] < 480 - 19).all()
    assert (l0["x"] == np.round(l0["x"])).all()
    kc, dc = oracle.orb_extract(const_frame(), 1000)
    assert len(kc) == 0


def test_knn2_degenerate(oracle):
    q = np.zeros((3, 32), np.uint8); t = np.zeros((1, 32), np.uint8)
    idx, dist = oracle.knn2(q, t)
    assert (idx[:, 0] == 0).all() and (idx[:, 1] == -1).all() and (dist[:, 1] == -1).all()
    pairs, mad, mad12 = oracle.line_match(q, t)                    # <2 train rows: defined as no matches
    assert len(pairs) == 0


def test_line_equations_and_keylines(oracle):
    kl, ld, fn, raw = oracle.lines_extract(synth_frame(1234), 40)
    assert len(kl) == 40 and (np.diff(kl["response"]) <= 0).all() and list(kl["class_id"]) == list(range(40))
    for i in range(len(kl)):
        sx, sy, ex, ey = [float(kl[f][i]) for f in ("startPointX", "startPointY", "endPointX", "endPointY")]
        l = np.cross([sx, sy, 1.0], [ex, ey, 1.0]); l = l / np.hypot(l[0], l[1])
        np.testing.assert_allclose(fn[i], l, rtol=1e-12, atol=1e-12)
        assert abs(fn[i] @ np.array([sx, sy, 1.0])) < 1e-6 and abs(np.hypot(fn[i][0], fn[i][1]) - 1) < 1e-12
        assert kl["numOfPixels"][i] == max(abs(round(ex) - round(sx)), abs(round(ey) - round(sy))) + 1


def test_vocabulary_text_loader_and_bow_vector(oracle, tmp_path):
    """oracle side of ORBVocabulary::loadFromTextFile + transform(features, BowVector, FeatureVector, levelsup) against a
    brute-force numpy restatement on a small ragged tree"""
    from synth import synthetic_vocab, write_vocab_text
    rng = np.random.default_rng(2)
    L, ptr, ch, nd, word, weight = synthetic_vocab(rng, k=5, L=3)
    for fmt, nl in (("%r", True), ("%.6g", False)):
        path = tmp_path / "voc.txt"
        write_vocab_text(path, 5, L, ptr, ch, nd, weight, weight_fmt=fmt, trailing_newline=nl)
        v = oracle.vocab_load_text(path)
        assert (v["k"], v["levels"], v["scoring"], v["weighting"]) == (5, L, 0, 0)
        np.testing.assert_array_equal(v["child_ptr"], ptr); np.testing.assert_array_equal(v["children"], ch); np.testing.assert_array_equal(v["node_desc"][1:], nd[1:])      # the root is not in the file
        leaf = ptr[1:] == ptr[:-1]
        np.testing.assert_array_equal(v["word_id"][leaf], word[leaf]); assert v["nwords"] == leaf.sum()
        np.testing.assert_array_equal(v["weight"][1:], np.array([float(fmt % float(x)) for x in weight[1:]]))
    feat = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    w = v["weight"]
    # brute force: descend by minimum Hamming distance, first child wins ties
    words, nodes, wts = [], [], []
    for f in feat:
        cur, lvl, nid = 0, 0, 0
        while ptr[cur + 1] > ptr[cur]:
            kids = ch[ptr[cur]:ptr[cur + 1]]
            dist = np.unpackbits(nd[kids] ^ f, axis=1).sum(axis=1)
            cur = int(kids[int(np.argmin(dist))]); lvl += 1
            if lvl == L - 1:
                nid = cur
        words.append(int(v["word_id"][cur])); nodes.append(nid); wts.append(float(w[cur]))
    bow, fv = {}, {}
    for i, (wd, nn, wt) in enumerate(zip(words, nodes, wts)):
        if wt > 0:
            bow[wd] = bow.get(wd, 0.0) + wt if wd in bow else wt
            fv.setdefault(nn, []).append(i)
    keys = sorted(bow); norm = 0.0
    for kk in keys:
        norm += abs(bow[kk])
    bw, bv, fn, fp, ff = oracle.compute_bow(L, ptr, ch, nd, v["word_id"], w, feat, levelsup=1)
    assert bw.tolist() == keys and bv.tolist() == [bow[kk] / norm for kk in keys]
    assert fn.tolist() == sorted(fv) and [ff[fp[j]:fp[j + 1]].tolist() for j in range(len(fn))] == [fv[kk] for kk in sorted(fv)]


def test_projection_overloads_are_parameterisations_of_the_ordered_matcher(oracle):
    """oracle vs oracle: the relocalisation (src/ORBmatcher.cc:1475-1602) and loop-closing (:293-406, src/LSDmatcher.cpp:558-683) overloads, restated
    line by line, equal the generic ordered window matcher (mode 1) under the argument mapping INTEGRATION.md gives"""
    import pkg
    from synth import synth_frame, warp_prev
    PQ = pkg.frontend().PQ_DTYPE
    rng = np.random.default_rng(9)
    cur = synth_frame(77, w=320, h=240); prev = warp_prev(cur)
    kp1, d1 = oracle.orb_extract(prev, 500); kp2, d2 = oracle.orb_extract(cur, 500)
    scales = oracle.orb_params()[0]
    b = (0.0, 320.0, 0.0, 240.0)
    q = np.zeros(len(kp1), PQ)
    q["u"] = kp1["x"] + 3; q["v"] = kp1["y"] - 2
    pred = np.clip(kp1["octave"] + rng.integers(-1, 2, len(kp1)), 0, 7)
    q["radius"] = 10 * scales[pred]; q["angle"] = kp1["angle"]; q["valid"] = rng.random(len(kp1)) < 0.9; q["obs_positive"] = 1
    occ = (rng.random(len(kp2)) < 0.3).astype(np.uint8)
    qr = q.copy(); qr["min_level"] = pred - 1; qr["max_level"] = pred + 1
    qd = q.copy(); qd["max_level"] = pred
    a0, n0 = oracle.search_by_projection_reloc(kp2, d2, qd, d1, occ, 80, True, bounds=b)
    a1, n1 = oracle.search_by_projection(0, 1, kp2, d2, qr, d1, occ, None, 0.0, 80, True, bounds=b)
    assert n0 == n1 and n0 > 20
    np.testing.assert_array_equal(a0, a1)
    qs = q.copy(); qs["min_level"] = pred - 1; qs["max_level"] = pred
    a2, n2 = oracle.search_by_projection_sim3(0, kp2, d2, qd, d1, occ, bounds=b)
    a3, n3 = oracle.search_by_projection(0, 1, kp2, d2, qs, d1, occ, None, 0.0, 50, False, bounds=b)
    assert n2 == n3 and n2 > 20
    np.testing.assert_array_equal(a2, a3)


def test_fast_and_orientation_against_scikit_image(oracle):
    """The one independent implementation available offline (scikit-image 0.18.3 in the build container, fixtures from
    tests/golden/make_skimage_fixtures.py): the FAST-9/16 corner SET at thresholds 20 and 7, OpenCV's cornerScore (= the largest
    threshold at which a pixel is still a corner, by definition) and the intensity-centroid orientation (exact atan2 vs fastAtan2's
    documented 0.3 degrees).  NMS, the cell logic and everything downstream have no independent counterpart."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "skimage_fast_orient.npz"))
    ncorner = nkp = 0
    for name in ("synth11", "synth12", "noise5"):
        img = g[name + "_img"]; h, w = img.shape
        maxth = g[name + "_fast_maxth"]
        for t in (7, 20):
            sc, kps = oracle.fast_image(img, t)
            mask = np.unpackbits(g["%s_fast_mask_t%d" % (name, t)])[:w * h].reshape(h, w).astype(bool)
            np.testing.assert_array_equal(sc > 0, mask, err_msg="%s: FAST corner set at threshold %d" % (name, t))
            np.testing.assert_array_equal(sc[mask], maxth[mask], err_msg="%s: cornerScore vs largest passing threshold" % name)
            assert (maxth[mask] >= t).all()
            ncorner += int(mask.sum())
            # what NMS keeps is a subset of the corner set, with the scores of the map
            assert all(mask[y, x] and sc[y, x] == s for x, y, s in kps)
        # orientation of the level-0 keypoints (their coordinates are image pixels)
        ref = {(int(x), int(y)): float(a) for (x, y), a in zip(g[name + "_orient_xy"], g[name + "_orient_deg"])}
        kp, _ = oracle.orb_extract(img, 500)
        for k in kp[kp["octave"] == 0]:
            key = (int(k["x"]), int(k["y"]))
            assert key in ref, key                                  # every level-0 ORB keypoint is a FAST corner at threshold 7
            d = abs(float(k["angle"]) - ref[key]) % 360.0
            assert min(d, 360.0 - d) < 0.35, (key, float(k["angle"]), ref[key])
            nkp += 1
    assert ncorner > 5000 and nkp > 150, (ncorner, nkp)


def test_sobel_and_nfa_against_scipy(oracle):
    """two more leaves with an independent implementation at hand: cv::Sobel 3x3 / BORDER_REFLECT_101 (scipy.ndimage.correlate, mode
    'mirror', exact integers) and the LSD number of false alarms (exact binomial tail from scipy.stats; the restated algorithm truncates
    the tail once the remainder is below 10 % of |NFA| x tail, so it may only err upwards and by a bounded amount).  The NFA half runs under decision D11's variant 0
    (log_gamma(n + 1), von Gioi's mathematical form): that proves variant 0 IS the binomial tail -- it does not say which form OpenCV ships; variant 1, the default since
    round 5, differs from it by exactly (lgamma(n + 1) - (n + 1)) / ln 10 in every value that passes through log1term (tests/test_variants_cpu.py)."""
    import ctypes as C
    from scipy import ndimage, stats
    img = synth_frame(5, w=200, h=120)
    gx = np.zeros(img.shape, np.int16); gy = np.zeros(img.shape, np.int16)
    oracle.L.orc_sobel3(img.ctypes.data_as(C.c_void_p), img.shape[1], img.shape[0], gx.ctypes.data_as(C.c_void_p), gy.ctypes.data_as(C.c_void_p))
    k = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]])
    np.testing.assert_array_equal(gx, ndimage.correlate(img.astype(np.int32), k, mode="mirror"))
    np.testing.assert_array_equal(gy, ndimage.correlate(img.astype(np.int32), k.T, mode="mirror"))
    oracle.L.orc_lsd_nfa.restype = C.c_double
    w, h = 512, 384
    log_nt = 5 * (np.log10(w) + np.log10(h)) / 2 + np.log10(11.0)
    rng = np.random.default_rng(3)
    checked = 0
    old_variant = oracle.L.orc_set_lsd_nfa_variant(0)
    for _ in range(3000):
        n = int(rng.integers(2, 600)); p = float(rng.choice([0.125, 0.0625, 0.03125]))
        kk = int(rng.integers(max(1, int(n * p)), n + 1))
        got = oracle.L.orc_lsd_nfa(w, h, n, kk, C.c_double(p))
        ref = -stats.binom.logsf(kk - 1, n, p) / np.log(10.0) - log_nt
        if ref > 250:                                     # first term below ~1e-260: the prescribed algorithm runs on denormals there (few significant bits)
            continue
        lg = 5e-5 * (1.0 + abs(ref))                      # the Windschitl / Lanczos log-gamma approximations the algorithm prescribes
        slack = np.log10(1.0 + 0.1 * abs(got)) + lg
        assert -lg <= got - ref <= slack, (n, kk, p, got, ref)
        checked += 1
    oracle.L.orc_set_lsd_nfa_variant(old_variant)
    assert checked > 2500
    assert oracle.L.orc_lsd_nfa(w, h, 0, 0, C.c_double(0.125)) == -log_nt                    # n == 0 or k == 0: -logNT


def test_fixed_point_leaves_track_their_real_valued_definitions(oracle):
    """The 8-bit resize, Gaussian blur and fastAtan2 are fixed-point / polynomial forms of plain real-valued operators; scipy / numpy
    evaluate those operators in floating point with the same sampling conventions (pixel-centre mapping, edge clamp, reflect-101).  The
    oracle must stay within the rounding those forms allow: this pins geometry, taps and borders, not the last bit."""
    from scipy import ndimage
    img = synth_frame(6)                                             # 640x480
    # cv::resize INTER_LINEAR, level 1 of the ORB pyramid = level 0 / 1.2 (src/ORBextractor.cc:1107-1132)
    lvl1 = oracle.pyramid_level(img, 1)
    oh, ow = lvl1.shape
    assert (ow, oh) == (533, 400)
    ref = ndimage.zoom(img.astype(np.float64), (oh / img.shape[0], ow / img.shape[1]), order=1, mode="nearest", grid_mode=True)
    assert ref.shape == lvl1.shape
    d = np.abs(lvl1.astype(np.float64) - ref)
    assert d.max() <= 1.0 and d.mean() < 0.3, (d.max(), d.mean())
    # the LSD 0.8x stage: 7x7 sigma 0.75 blur, then INTER_LINEAR_EXACT to 512x384
    sc = oracle.lsd_scaled(img)
    assert sc.shape == (384, 512)
    # cv::GaussianBlur 7x7 sigma 2, BORDER_REFLECT_101 (src/ORBextractor.cc:1086)
    x = np.arange(-3, 4, dtype=np.float64); k = np.exp(-x * x / (2 * 2.0 * 2.0)); k /= k.sum()
    refb = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    db = np.abs(oracle.blur7(img).astype(np.float64) - refb)
    assert db.max() <= 1.5 and db.mean() < 0.3, (db.max(), db.mean())      # taps quantised to 1/256 + two roundings
    k2 = np.exp(-x * x / (2 * 0.75 * 0.75)); k2 /= k2.sum()
    blur = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), k2, axis=1, mode="mirror"), k2, axis=0, mode="mirror")
    refs = ndimage.zoom(blur, (384 / 480, 512 / 640), order=1, mode="nearest", grid_mode=True)
    ds = np.abs(sc.astype(np.float64) - refs)
    assert ds.max() <= 1.5 and ds.mean() < 0.35, (ds.max(), ds.mean())
    # cv::fastAtan2: documented accuracy ~0.3 degrees, range [0, 360)
    rng = np.random.default_rng(8)
    ys = rng.integers(-40000, 40000, 20000).astype(np.float32); xs = rng.integers(-40000, 40000, 20000).astype(np.float32)
    got = np.array([oracle.fast_atan2(float(y), float(x)) for y, x in zip(ys, xs)])
    want = np.degrees(np.arctan2(ys.astype(np.float64), xs.astype(np.float64))) % 360.0
    dd = np.abs(got - want); dd = np.minimum(dd, 360.0 - dd)
    assert dd.max() < 0.3 and (got >= 0).all() and (got < 360.0 + 1e-4).all(), dd.max()


def test_cpu_allcores_script():
    """the all-cores leg of bench.py's cpu_baseline: two pinned oracle processes for a second each, one JSON line back"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "cpu_allcores.py"), "320", "240", "300", "50", "1", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["cores"] in (1, 2) and out["value"] > 1 and out["frames"] >= 2


def test_nfa_tables_are_pinned(oracle):
    """decision D8: the NFA stage's log-gamma / log(p) tables are evaluated on the HOST (the reference's own libm expressions), so their bits
    depend on the libm of the box that runs the library.  Goldens generated in the build container (glibc 2.35): the library's table, the
    oracle's log_gamma and the goldens must agree bit for bit wherever the suite runs."""
    import ctypes as C
    import pkg
    g = np.load(os.path.join(GOLD, "nfa_tables.npz"))
    pkg.builder().build(force=False, verbose=False)
    L = C.CDLL(pkg.builder().TEST_LIB)      # (a test entry point: include/sslam_testing.h; the table code itself is the product's)
    n = int(g["j"].max()) + 4
    tab = np.zeros(2 * n + 48)
    assert L.sslam_debug_nfa_tables(n, C.c_void_p(tab.ctypes.data)) == 0
    np.testing.assert_array_equal(tab[g["j"]].view(np.uint64), g["lgam_bits"])
    np.testing.assert_array_equal(tab[n:n + 48].view(np.uint64), g["plog_bits"])
    oracle.L.orc_log_gamma.restype = C.c_double; oracle.L.orc_log_gamma.argtypes = [C.c_double]
    some = g["j"][::37]
    assert [np.float64(oracle.L.orc_log_gamma(float(j))).view(np.uint64) for j in some] == list(g["lgam_bits"][::37])
    assert tab[1] == tab[2] == 0.0 or abs(tab[1]) < 1e-12                    # log Gamma(1) = log Gamma(2) = 0 up to the approximation
    assert (tab[n + 48 + 1:n + 48 + 5] == [1.0, 0.5, 1.0 / 3.0, 0.25]).all()
