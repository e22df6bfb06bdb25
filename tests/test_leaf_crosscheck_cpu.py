"""Independent cross-checks of two of the oracle's OpenCV leaves (UPSTREAM-RECALL: restated from memory, no OpenCV in this image) against other people's code that
happens to be installed -- not a pin (different libraries), but the parts whose definition leaves no freedom must agree exactly:

* FAST-9/16: WHICH pixels pass the segment test at a threshold is the published definition (>= 9 contiguous ring pixels all brighter than p + t or all darker than
  p - t); scikit-image's corner_fast (0.18, /opt/conda's python3.9 -- not importable from this interpreter, hence the subprocess) implements the same test on its own.
  Compared: the corner set before non-maximum suppression.  cv::FAST's score and its NMS are NOT covered (scikit-image's response is a different quantity).
* Sobel 3x3 with BORDER_REFLECT_101 (LBD's gradient): integer arithmetic, scipy.ndimage.sobel with mode="mirror"."""
import os, subprocess, sys
import numpy as np
import pytest
from synth import synth_frame, noise_frame

CONDA_PY = "/opt/conda/bin/python3.9"
SK_SCRIPT = """
import sys, numpy as np
from skimage.feature import corner_fast
img = np.load(sys.argv[1]).astype(np.float64)          # integer-valued doubles and an integer threshold: every comparison is exact
np.save(sys.argv[3], corner_fast(img, n=9, threshold=float(sys.argv[2])) > 0)
"""


def _have_skimage():
    if not os.path.exists(CONDA_PY):
        return False
    return subprocess.run([CONDA_PY, "-c", "import skimage.feature"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 0


@pytest.mark.skipif(not _have_skimage(), reason="no scikit-image interpreter at /opt/conda/bin/python3.9")
@pytest.mark.parametrize("threshold", [20, 7])
def test_fast_segment_test_equals_scikit_image(oracle, tmp_path, threshold):
    script = tmp_path / "sk.py"; script.write_text(SK_SCRIPT)
    for k, img in enumerate((synth_frame(1234), noise_frame(7)[:240, :320].copy(), np.full((64, 80), 127, np.uint8))):
        score_map, _ = oracle.fast_image(img, threshold)
        np.save(tmp_path / "img.npy", img)
        subprocess.run([CONDA_PY, str(script), str(tmp_path / "img.npy"), str(threshold), str(tmp_path / "out.npy")], check=True, stderr=subprocess.DEVNULL)
        theirs = np.load(tmp_path / "out.npy")
        ours = score_map > 0
        assert not ours[:3].any() and not ours[-3:].any() and not ours[:, :3].any() and not ours[:, -3:].any()          # cv::FAST skips the 3-pixel rim
        inner = np.zeros_like(ours); inner[3:-3, 3:-3] = True
        assert np.array_equal(ours & inner, theirs & inner), "image %d: %d corners differ" % (k, int(((ours != theirs) & inner).sum()))
        if k < 2:
            assert ours.sum() > 1000


def test_sobel3_reflect101_equals_scipy(oracle):
    import ctypes as C
    from scipy import ndimage
    from oracle_lib import _p
    for img in (noise_frame(3)[:200, :300].copy(), synth_frame(77, w=331, h=113)):
        h, w = img.shape
        gx = np.zeros((h, w), np.int16); gy = np.zeros((h, w), np.int16)
        oracle.L.orc_sobel3(_p(np.ascontiguousarray(img)), w, h, _p(gx), _p(gy))
        i32 = img.astype(np.int32)
        assert np.array_equal(gx, ndimage.sobel(i32, axis=1, mode="mirror")) and np.array_equal(gy, ndimage.sobel(i32, axis=0, mode="mirror"))


def test_resize_blur_atan2_agree_with_float_references_to_within_a_grey_level(oracle):
    """Not bit-level (OpenCV's 8-bit resize and blur are fixed-point, the references below are float64), but the sampling geometry, the border rule and the kernel are
    what a whole grey level of disagreement would mean: cv::resize INTER_LINEAR samples at (x + 0.5) * scale - 0.5 (torch's align_corners=False), GaussianBlur
    7x7 sigma 2 reflects without repeating the edge pixel (scipy's "mirror"), fastAtan2 is within 0.01 degrees of atan2."""
    import ctypes as C
    import torch
    from scipy import ndimage
    img = synth_frame(1234)
    prev = img
    for level in (1, 2, 3):
        cur = oracle.pyramid_level(img, level)                      # level l is resized from level l - 1 (src/ORBextractor.cc:1118-1122)
        ref = torch.nn.functional.interpolate(torch.from_numpy(prev.astype(np.float64))[None, None], size=cur.shape, mode="bilinear", align_corners=False)[0, 0].numpy()
        assert np.abs(cur.astype(np.float64) - ref).max() < 1.0
        prev = cur
    x = np.arange(7) - 3.0; k = np.exp(-x * x / 8.0); k /= k.sum()
    ref = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    assert np.abs(oracle.blur7(img).astype(np.float64) - ref).max() < 1.0
    oracle.L.orc_fast_atan2.restype = C.c_float
    rng = np.random.default_rng(1)
    ys = (rng.normal(size=5000) * 100).astype(np.float32); xs = (rng.normal(size=5000) * 100).astype(np.float32)
    a = np.array([oracle.L.orc_fast_atan2(C.c_float(float(y)), C.c_float(float(x))) for y, x in zip(ys, xs)])
    d = np.abs(a - np.degrees(np.arctan2(ys.astype(np.float64), xs.astype(np.float64))) % 360); d = np.minimum(d, 360 - d)
    assert d.max() < 0.01 and a.min() >= 0 and a.max() < 360


def _ring(img):
    """the 16 pixels of the radius-3 Bresenham circle around every interior pixel, in order"""
    off = [(0, -3), (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3), (0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1), (-2, -2), (-1, -3)]
    h, w = img.shape
    c = img[3:h - 3, 3:w - 3].astype(np.int32)
    return c, [img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx].astype(np.int32) for dx, dy in off]


@pytest.mark.parametrize("threshold", [20, 7])
def test_fast_score_is_the_largest_threshold_that_keeps_the_corner(oracle, threshold):
    """cv::FAST's response (cornerScore<16>) by its DEFINITION, not by its formula: the largest t for which the pixel still passes the segment test
    (9 contiguous ring pixels all > p + t or all < p - t), found by trying every t.  The oracle computes it the way OpenCV does (maxima of arc minima)."""
    for img in (synth_frame(1234)[100:340, 200:520].copy(), noise_frame(7)[:120, :160].copy()):
        score_map, _ = oracle.fast_image(img, threshold)
        c, ring = _ring(img)
        best = np.full(c.shape, -1, np.int32)
        for t in range(threshold, 256):
            br = [r > c + t for r in ring]; dk = [r < c - t for r in ring]
            ok = np.zeros(c.shape, bool)
            for s in range(16):
                b = br[s]; d = dk[s]
                for k in range(1, 9):
                    b = b & br[(s + k) & 15]; d = d & dk[(s + k) & 15]
                ok |= b | d
            if not ok.any():
                break
            best[ok] = t
        want = np.where(best >= threshold, best, 0)
        got = score_map[3:-3, 3:-3].astype(np.int32)
        assert (want > 0).sum() > 300
        assert np.array_equal(got, want), "%d pixels differ" % int((got != want).sum())


def test_fast_suppression_keeps_strict_3x3_maxima_of_the_response(oracle):
    """cv::FAST(nonmaxSuppression = true) as recalled: a corner stays iff its response is strictly greater than the responses of its eight neighbours (a pixel that is no
    corner has response 0).  A second statement of the same rule in numpy over the oracle's own response map -- it checks the oracle's code, not the recollection."""
    for img, t in ((synth_frame(1234), 20), (noise_frame(7)[:240, :320].copy(), 7), (synth_frame(91, w=333, h=251), 7)):
        sc, kps = oracle.fast_image(img, t)
        s = sc.astype(np.int32); h, w = s.shape
        nb = np.zeros_like(s)
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dy or dx:
                    sh = np.zeros_like(s); sh[max(0, -dy):h - max(0, dy), max(0, -dx):w - max(0, dx)] = s[max(0, dy):h - max(0, -dy), max(0, dx):w - max(0, -dx)]
                    nb = np.maximum(nb, sh)
        keep = (s > 0) & (s > nb)
        got = np.zeros_like(keep); got[kps[:, 1], kps[:, 0]] = True
        assert np.array_equal(got, keep) and len(kps) == int(keep.sum()) > 100
        assert np.array_equal(kps[:, 2], s[kps[:, 1], kps[:, 0]])                     # the keypoint's response is the map's
        assert np.array_equal(np.lexsort((kps[:, 0], kps[:, 1])), np.arange(len(kps)))      # raster order


def test_lsd_nfa_is_the_binomial_tail_within_its_own_tolerance(oracle):
    """LSD's nfa(n, k, p) = -log10(NT * P[Binomial(n, p) >= k]) with NT = (w h)^(5/2) * 11 (LSD_REFINE_ADV's eleven tolerances), computed by upstream with log-gamma
    approximations and a tail sum that stops at 10 % relative error (so up to log10(1.1) = 0.041 off by design).  Against scipy's exact tail: 0.0014 at most over these
    cases; its log-gamma (Lanczos below 15, Windschitl above) within 1e-12 of scipy's.  This is decision D11's variant 0 (log_gamma(n + 1): the mathematical form);
    the default since round 5, variant 1, is what OpenCV's source is recalled to compute instead and is NOT the binomial tail (tests/test_variants_cpu.py)."""
    import ctypes as C
    from scipy import stats, special
    L = oracle.L
    L.orc_lsd_nfa.restype = C.c_double; L.orc_lsd_nfa.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
    L.orc_log_gamma.restype = C.c_double; L.orc_log_gamma.argtypes = [C.c_double]
    xs = np.concatenate([np.arange(1, 200), np.random.default_rng(0).integers(200, 400000, 500)]).astype(float)
    lg = np.array([L.orc_log_gamma(x) for x in xs]); ref = special.gammaln(xs)
    assert np.max(np.abs(lg - ref) / np.maximum(1, np.abs(ref))) < 1e-11
    rng = np.random.default_rng(1); w, h = 512, 384
    log_nt = 5 * (np.log10(w) + np.log10(h)) / 2 + np.log10(11.0)
    worst = 0.0
    old_variant = L.orc_set_lsd_nfa_variant(0)
    for it in range(4000):
        p = float(rng.choice([0.125, 0.0625, 0.03125, 0.015625, 1 / 128]))
        n = int(rng.integers(1, 30000)) if it % 3 else int(rng.integers(1, 200))
        lo = int(n * p * 0.5)
        k = int(min(n, rng.integers(lo, lo + max(2, int(4 * np.sqrt(n * p * (1 - p)) + n * p)))))
        got = L.orc_lsd_nfa(w, h, n, k, p)
        want = -log_nt if k == 0 else -(stats.binom.logsf(k - 1, n, p) / np.log(10) + log_nt)
        if np.isfinite(want):
            worst = max(worst, abs(got - want))
    L.orc_set_lsd_nfa_variant(old_variant)
    assert worst < 0.0414, worst
    assert worst < 0.005, worst            # what it actually achieves
