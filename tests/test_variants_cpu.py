"""CPU checks of the oracle's selectable decisions of the line path (D11 nfa first term, D12 LBD bit order) and of one identity that settles a third question
(cv::resize(INTER_LINEAR) at an exact 2x decimation takes the INTER_AREA path upstream: the two give the same bytes).  The library's side of D11 / D12 / D7 / D2 is
tests/test_variants_gpu.py."""
import ctypes as C
import math
import numpy as np
import pytest
from synth import synth_frame, noise_frame


@pytest.fixture(scope="module")
def orc():
    import oracle_lib
    return oracle_lib.Oracle()


def _lg(x, orc):
    orc.L.orc_log_gamma.restype = C.c_double; orc.L.orc_log_gamma.argtypes = [C.c_double]
    return orc.L.orc_log_gamma(float(x))


def test_nfa_variant_is_the_first_term_only(orc):
    """orc_set_lsd_nfa_variant(1): log1term's first term is (n + 1) instead of log_gamma(n + 1); nothing else of nfa() changes.  Checked on the closed form of the branch
    that returns before the tail loop (term underflows to 0 under variant 1 for large n) and, for small n, on a restatement of the tail loop."""
    L = orc.L
    L.orc_lsd_nfa.restype = C.c_double; L.orc_lsd_nfa.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
    w, h = 512, 384
    logNT = 5 * (math.log10(w) + math.log10(h)) / 2 + math.log10(11.0)

    def nfa_py(n, k, p, variant):
        if n == 0 or k == 0: return -logNT
        if n == k: return -logNT - n * math.log10(p)
        first = (n + 1.0) if variant else _lg(n + 1.0, orc)
        log1 = first - _lg(k + 1.0, orc) - _lg(n - k + 1.0, orc) + k * math.log(p) + (n - k) * math.log(1.0 - p)
        try: term = math.exp(log1)
        except OverflowError: term = float("inf")
        if term <= 100 * 5e-324:      # double_equal(term, 0): an exact zero or a denormal of at most 100 units
            return (-log1 / math.log(10.0) - logNT) if k > n * p else -logNT
        tail = term; pt = p / (1 - p)
        for i in range(k + 1, n + 1):
            bt = (n - i + 1) / i; mt = bt * pt; term *= mt; tail += term
            if bt < 1:
                err = term * ((1 - mt ** (n - i + 1)) / (1 - mt) - 1)
                if err < 0.1 * abs(-math.log10(tail) - logNT) * tail: break
        return -math.log10(tail) - logNT
    rng = np.random.default_rng(5)
    cases = [(int(n), int(rng.integers(1, n)), float(0.125 / 2 ** int(rng.integers(0, 5)))) for n in rng.integers(2, 3000, 400)]
    for variant in (0, 1):
        old = L.orc_set_lsd_nfa_variant(variant)
        try:
            for n, k, p in cases:
                got = L.orc_lsd_nfa(w, h, n, k, p); want = nfa_py(n, k, p, variant)
                assert got == pytest.approx(want, rel=1e-9, abs=1e-9), (variant, n, k, p)
        finally:
            L.orc_set_lsd_nfa_variant(old)
    # what the variant means for the detector: the rectangle of a small, poorly aligned region is meaningful under 1 and not under 0
    old = L.orc_set_lsd_nfa_variant(0)
    try:
        v0 = L.orc_lsd_nfa(w, h, 60, 18, 0.125)
        L.orc_set_lsd_nfa_variant(1)
        v1 = L.orc_lsd_nfa(w, h, 60, 18, 0.125)
    finally: L.orc_set_lsd_nfa_variant(old)
    assert old == 1, "variant 1 (OpenCV as recalled) is the default since round 5"
    assert v0 < 0 < v1 and v1 - v0 == pytest.approx((_lg(61.0, orc) - 61.0) / math.log(10.0), rel=1e-6)


def test_nfa_variant_accepts_more_segments_and_keeps_the_candidates(orc):
    for img in (synth_frame(1234), synth_frame(91, w=333, h=251)):
        b = orc.lines_extract(img, 400)[3]                      # the default: variant 1
        old = orc.L.orc_set_lsd_nfa_variant(0)
        try: a = orc.lines_extract(img, 400)[3]
        finally: orc.L.orc_set_lsd_nfa_variant(old)
        assert len(b) > 1.5 * len(a)
        sa = set(map(bytes, a.view(np.uint8).reshape(len(a), -1))); sb = set(map(bytes, b.view(np.uint8).reshape(len(b), -1)))
        assert len(sa & sb) > 0.8 * len(sa)          # a rectangle that passed at its first rect_nfa under 0 passes unchanged under 1
        c = orc.lines_extract(img, 400)[3]
        np.testing.assert_array_equal(b, c)


def test_lbd_bit_order_reverses_every_byte_and_no_distance(orc):
    rev = np.array([int("{:08b}".format(i)[::-1], 2) for i in range(256)], np.uint8)
    res = []
    for img in (synth_frame(2000), synth_frame(2001)):
        kl1, ld1, fn1, raw1 = orc.lines_extract(img, 200)      # the default: 0x80 >> i
        old = orc.L.orc_set_lbd_bit_order(0)
        try: kl0, ld0, fn0, raw0 = orc.lines_extract(img, 200)
        finally: orc.L.orc_set_lbd_bit_order(old)
        assert old == 1
        np.testing.assert_array_equal(kl0, kl1); np.testing.assert_array_equal(raw0, raw1); np.testing.assert_array_equal(fn0, fn1)
        np.testing.assert_array_equal(ld1, rev[ld0])
        assert (ld1 != ld0).mean() > 0.5
        res.append((ld0, ld1))
    (a0, a1), (b0, b1) = res
    np.testing.assert_array_equal(orc.hamming_matrix(a0, b0), orc.hamming_matrix(a1, b1))
    for gate, ratio in ((0.5, False), (0.1, False), (0.5, True)):
        p0 = orc.line_match(a0, b0, gate, ratio); p1 = orc.line_match(a1, b1, gate, ratio)
        np.testing.assert_array_equal(p0[0], p1[0]); assert p0[1:] == p1[1:]


@pytest.mark.parametrize("w,h", [(640, 480), (320, 240), (128, 96)])
def test_resize_linear_at_exact_2x_equals_the_area_average(orc, w, h):
    """cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) switches to INTER_AREA when both scale factors are exactly 2 (imgproc/src/resize.cpp; reached by
    src/ORBextractor.cc:1120 with scaleFactor 2.0 on even sizes).  INTER_AREA's 2x2 fast path is (a + b + c + d + 2) >> 2.  INTER_LINEAR's own 8u arithmetic at that
    scale has fx = fy = 0.5 exactly, coefficients 1024 / 1024, and ((1024 * ((a + b) * 1024 >> 4)) >> 16) == a + b: the same bytes.  So the restated leaf (and k_resize,
    which follows it) needs no second path; this test is the evidence."""
    for img in (synth_frame(7, w=w, h=h), noise_frame(8, w=w, h=h)):
        lvl1 = orc.pyramid_level(img, 1, scale=2.0, nlevels=3)
        assert lvl1.shape == (h // 2, w // 2)
        i = img.astype(np.int32)
        area = (i[0::2, 0::2] + i[0::2, 1::2] + i[1::2, 0::2] + i[1::2, 1::2] + 2) >> 2
        np.testing.assert_array_equal(lvl1, area.astype(np.uint8))
        lvl2 = orc.pyramid_level(img, 2, scale=2.0, nlevels=3)
        j = lvl1.astype(np.int32)
        np.testing.assert_array_equal(lvl2, ((j[0::2, 0::2] + j[0::2, 1::2] + j[1::2, 0::2] + j[1::2, 1::2] + 2) >> 2).astype(np.uint8))
