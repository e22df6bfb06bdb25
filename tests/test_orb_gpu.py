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
