"""GPU parity: HIP ORB path vs the CPU oracle, bit-exact, stage by stage and end to end."""
import numpy as np
import pytest
from synth import synth_frame, noise_frame, const_frame

pytestmark = pytest.mark.gpu


def _cmp_orb(fe, ctx, oracle, img, nfeat=1000, nlevels=8, scale=1.2):
    ex = fe.OrbExtractor(ctx, nfeat, scale, nlevels, 20, 7)
    try:
        kp, desc = ex(img)
        okp, odesc = oracle.orb_extract(img, nfeat, scale, nlevels)
        for l in range(nlevels):
            np.testing.assert_array_equal(ex.debug_level(0, l), oracle.pyramid_level(img, l, scale, nlevels), err_msg="pyramid level %d" % l)
            np.testing.assert_array_equal(ex.debug_candidates(0, l), oracle.candidates(img, l, nfeat, scale, nlevels), err_msg="FAST candidates level %d" % l)
        assert len(kp) == len(okp)
        for f in ("x", "y", "size", "response", "octave", "class_id"):
            np.testing.assert_array_equal(kp[f], okp[f], err_msg=f)
        np.testing.assert_array_equal(kp["angle"].view(np.uint32), okp["angle"].view(np.uint32), err_msg="angle bits")
        np.testing.assert_array_equal(desc, odesc)
        return len(kp)
    finally:
        ex.close()


def test_orb_synth_640(fe, ctx, oracle):
    n = _cmp_orb(fe, ctx, oracle, synth_frame(1234))
    assert 900 <= n <= 1016


def test_orb_synth_2000(fe, ctx, oracle):
    _cmp_orb(fe, ctx, oracle, synth_frame(1236), nfeat=2000)


def test_orb_noise(fe, ctx, oracle):
    _cmp_orb(fe, ctx, oracle, noise_frame(7))


def test_orb_constant(fe, ctx, oracle):
    ex = fe.OrbExtractor(ctx)
    kp, desc = ex(const_frame())
    assert len(kp) == 0 and desc.shape == (0, 32)
    ex.close()


def test_orb_odd_size(fe, ctx, oracle):
    _cmp_orb(fe, ctx, oracle, synth_frame(77, w=333, h=251), nfeat=500)


def test_orb_wide(fe, ctx, oracle):
    _cmp_orb(fe, ctx, oracle, synth_frame(78, w=800, h=300), nfeat=700)      # nIni = 3 roots


def test_orb_1280(fe, ctx, oracle):
    _cmp_orb(fe, ctx, oracle, synth_frame(1235, w=1280, h=960), nfeat=2000)


def test_orb_blur_variant_opencv_340(fe, ctx, oracle):
    """Decision D6 has a selectable alternative: the 8-bit GaussianBlur of OpenCV 3.4.0 (every tap of the float kernel rounded: 18 34 49 55 49 34 18, sum 257,
    saturated) instead of the bit-exact filter of >= 3.4.1.  Same keypoints, other descriptors; the library under variant 1 equals the oracle under variant 1
    (which equals the reference compiled with that leaf: oracle/ref_pin), and switching back restores variant 0."""
    for img, nfeat in ((synth_frame(1234), 1000), (synth_frame(1235, w=1280, h=960), 2000), (np.full((240, 320), 255, np.uint8), 300)):
        ex = fe.OrbExtractor(ctx, nfeat)
        try:
            kp0, d0 = ex(img)
            ex.set_blur_variant(1)
            kp1, d1 = ex(img)
            try:
                oracle.set_gauss_variant(1)
                okp1, od1 = oracle.orb_extract(img, nfeat)
            finally:
                oracle.set_gauss_variant(0)
            np.testing.assert_array_equal(kp1.view(np.uint8), okp1.view(np.uint8)); np.testing.assert_array_equal(d1, od1)
            np.testing.assert_array_equal(kp1.view(np.uint8), kp0.view(np.uint8))                    # the blur only feeds the descriptors
            if len(kp0):
                frac = np.unpackbits(d0 ^ d1).mean()
                assert 0.01 < frac < 0.10, frac                                                       # ~3.7 % of the bits
            ex.set_blur_variant(0)
            kp2, d2 = ex(img)
            np.testing.assert_array_equal(d2, d0)
            with pytest.raises(fe.SslamError):
                ex.set_blur_variant(2)
        finally:
            ex.close()


@pytest.mark.parametrize("scale,nlevels", [(1.1, 8), (1.5, 5), (2.0, 4), (2.6, 3), (3.4, 3)])
def test_orb_scale_factors(fe, ctx, oracle, scale, nlevels):
    """k_resize's group records (round 5: offsets, v_perm selectors and coefficient pairs per four outputs, built on the host) over the scale factors the constructor accepts:
    1.1 (windows overlap almost completely), 2.0 (the exact 2x decimation: INTER_LINEAR == INTER_AREA there, tests/test_variants_cpu.py), 2.6 (a pair's taps still inside its
    8-byte window) and 3.4 (they are not: the byte path).  Every level's bytes, the candidates and the keypoints against the oracle; odd and even sizes."""
    for img in (synth_frame(77, w=640, h=480), synth_frame(78, w=333, h=251)):
        _cmp_orb(fe, ctx, oracle, img, 800, nlevels, scale)


def test_sincos_selftest(fe, ctx):
    """k_describe's own double sin / cos on [0, 6.5] (common.h, sincos_0_2pi) against the library's sincos after the rounding to float, for every float of the range"""
    import ctypes as C
    bad = C.c_longlong(-1)
    rc = fe.testing_lib().sslam_selftest_sincos(ctx.h, C.byref(bad))
    assert rc == 0 and bad.value == 0, (rc, bad.value)
