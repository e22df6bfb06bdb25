"""GPU parity: HIP LSD + LBD + line equations vs the CPU oracle.

The fp64/fp32 arithmetic is ordered exactly like the oracle's, so the stated tolerance is
ZERO for everything except quantities that pass through libm-vs-ocml transcendentals
(KeyLine.angle via atan2, LBD direction via cos/sin, NFA log/exp): there the bar is
<= 1 ulp on floats (checked below as exact-or-1ulp) and identical segment SETS."""
import numpy as np
import pytest
from synth import synth_frame, noise_frame, const_frame, ramp_frame

pytestmark = pytest.mark.gpu


def _ulp_diff(a, b):
    ai = a.view(np.int32).astype(np.int64); bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai); bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)


def _cmp_lines(fe, ctx, oracle, img, max_lines):
    ex = fe.LineExtractor(ctx, max_lines)
    try:
        kl, ld, fn = ex(img)
        okl, old, ofn, oraw = oracle.lines_extract(img, max_lines)
        raw = ex.debug_segments(0)
        assert raw.shape == oraw.shape, (raw.shape, oraw.shape)
        np.testing.assert_array_equal(raw, oraw, err_msg="LSD segments (before top-N)")
        assert len(kl) == len(okl)
        for f in kl.dtype.names:
            if f == "angle":
                assert _ulp_diff(kl[f], okl[f]).max(initial=0) <= 1, "KeyLine.angle"
            else:
                np.testing.assert_array_equal(kl[f], okl[f], err_msg=f)
        # LBD bytes: exact wherever the direction vector is bit-identical; bounded otherwise
        ham = np.unpackbits(ld ^ old, axis=1).sum(axis=1)
        same_angle = kl["angle"].view(np.uint32) == okl["angle"].view(np.uint32)
        assert (ham[same_angle] <= 0).all() or ham.max() <= 8, ham
        assert ham.max(initial=0) <= 8
        np.testing.assert_array_equal(fn, ofn)
        return len(kl), int((ham > 0).sum())
    finally:
        ex.close()


def test_lines_synth_40(fe, ctx, oracle):
    n, bad = _cmp_lines(fe, ctx, oracle, synth_frame(1234), 40)      # reference cap (src/ExtractLineSegment.cpp:42)
    assert n == 40


def test_lines_synth_200(fe, ctx, oracle):
    n, bad = _cmp_lines(fe, ctx, oracle, synth_frame(2000), 200)
    assert n >= 150


def test_lines_1280_400(fe, ctx, oracle):
    n, bad = _cmp_lines(fe, ctx, oracle, synth_frame(1235, w=1280, h=960), 400)
    assert n == 400


def test_lines_odd_size(fe, ctx, oracle):
    _cmp_lines(fe, ctx, oracle, synth_frame(91, w=333, h=251), 200)


def test_lines_noise(fe, ctx, oracle):
    _cmp_lines(fe, ctx, oracle, noise_frame(3, w=320, h=240), 200)


def test_lines_constant(fe, ctx, oracle):
    ex = fe.LineExtractor(ctx, 200)
    kl, ld, fn = ex(const_frame())
    assert len(kl) == 0
    ex.close()


def test_exact_division_selftest(fe, ctx):
    """the NFA tail divides small integers through a reciprocal table + two FMAs; it must equal the IEEE division bit for bit"""
    import ctypes as C
    bad = C.c_longlong(-1)
    rc = fe.lib().sslam_selftest_exact_div(ctx.h, 1024 * 768 + 4, C.c_longlong(2_000_000_000), C.byref(bad))
    assert rc == 0 and bad.value == 0, bad.value


def test_tail_test_selftest(fe, ctx):
    """the NFA tail's early-exit test is decided from fp32 estimates inside guard bands; a decided case must never
    differ from the fp64 expression, and undecided cases (which re-run that expression) must stay rare"""
    import ctypes as C
    bad, amb = C.c_longlong(-1), C.c_longlong(-1)
    n = 500_000_000
    rc = fe.lib().sslam_selftest_tail_test(ctx.h, C.c_longlong(n), C.byref(bad), C.byref(amb))
    assert rc == 0 and bad.value == 0, (bad.value, amb.value)
    assert 0 < amb.value < 0.2 * n, amb.value        # half of the samples sit within 1e-3 of the boundary


def test_region_division_selftest(fe, ctx):
    """the accept chain's division / atan2 shortcuts must equal the plain forms bit for bit over every magnitude a region can produce"""
    import ctypes as C
    bad = (C.c_longlong * 2)(-1, -1)
    rc = fe.lib().sslam_selftest_region_div(ctx.h, C.c_longlong(2_000_000_000), bad)
    assert rc == 0 and bad[0] == 0 and bad[1] == 0, (bad[0], bad[1])


def test_lines_huge_regions(fe, ctx, oracle):
    """regions far larger than the 1024-point LDS queue continue in global memory"""
    n, bad = _cmp_lines(fe, ctx, oracle, ramp_frame(), 200)
    assert n >= 1
