"""GPU parity: HIP LSD + LBD + line equations vs the CPU oracle.

The fp64/fp32 arithmetic is ordered exactly like the oracle's, so the stated tolerance is
ZERO for everything except quantities that pass through libm-vs-ocml transcendentals
(KeyLine.angle via atan2, LBD direction via cos/sin, NFA log/exp): there the bar is
<= 1 ulp on floats (checked below as exact-or-1ulp) and identical segment SETS."""
import numpy as np
import pytest
import torch
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
    rc = fe.testing_lib().sslam_selftest_exact_div(ctx.h, 1024 * 768 + 4, C.c_longlong(2_000_000_000), C.byref(bad))
    assert rc == 0 and bad.value == 0, bad.value


def test_tail_test_selftest(fe, ctx):
    """the NFA tail's early-exit test is decided from fp32 estimates inside guard bands; a decided case must never
    differ from the fp64 expression, and undecided cases (which re-run that expression) must stay rare"""
    import ctypes as C
    bad, amb = C.c_longlong(-1), C.c_longlong(-1)
    n = 500_000_000
    rc = fe.testing_lib().sslam_selftest_tail_test(ctx.h, C.c_longlong(n), C.byref(bad), C.byref(amb))
    assert rc == 0 and bad.value == 0, (bad.value, amb.value)
    assert 0 < amb.value < 0.2 * n, amb.value        # half of the samples sit within 1e-3 of the boundary


def test_lsd_bin_selftest(fe, ctx):
    """k_lsd_hist_sort's fp32-guarded gradient bin against the fp64 expression for every |g|^2 of frames with several maxima (lsd_front.h, lsd_bin_wave)"""
    import ctypes as C
    for max_s in (0, 1, 7, 1023, 4096, 65025, 130050, 520200, 1040400, 2080800, 16777215):
        bad = C.c_longlong(-1)
        rc = fe.testing_lib().sslam_selftest_lsd_bin(ctx.h, max_s, C.byref(bad))
        assert rc == 0 and bad.value == 0, (max_s, rc, bad.value)


def test_lbd_rounding_selftest(fe, ctx):
    """k_lbd<true> rounds a walk coordinate with v_cvt_rpi_i32_f32 + a median where the reference has (short)round(x) under a clamp: every float bit pattern of the coordinate
    range, against the previous sequence under the clamps and against roundf for x >= 0 (lbd.h, k_selftest_lbd_round)"""
    import ctypes as C
    bad = (C.c_longlong * 2)(-1, -1)
    rc = fe.testing_lib().sslam_selftest_lbd_round(ctx.h, bad)
    assert rc == 0 and bad[0] == 0 and bad[1] == 0, (rc, bad[0], bad[1])


def test_region_division_selftest(fe, ctx):
    """the accept chain's division / atan2 shortcuts must equal the plain forms bit for bit over every magnitude a region can produce"""
    import ctypes as C
    bad = (C.c_longlong * 2)(-1, -1)
    rc = fe.testing_lib().sslam_selftest_region_div(ctx.h, C.c_longlong(2_000_000_000), bad)
    assert rc == 0 and bad[0] == 0 and bad[1] == 0, (bad[0], bad[1])


def test_lines_huge_regions(fe, ctx, oracle):
    """regions far larger than the 1024-point LDS queue continue in global memory"""
    n, bad = _cmp_lines(fe, ctx, oracle, ramp_frame(), 200)
    assert n >= 1


@pytest.mark.parametrize("knobs", [{"SSLAM_LSD_FUSED": "0"}, {"SSLAM_LSD_SORT_RUNS": "0"}, {"SSLAM_LSD_FUSED": "0", "SSLAM_LSD_SORT_RUNS": "0"}, {"SSLAM_LSD_SPILLFREE": "0"}, {"SSLAM_LBD_RPI": "0"}])
def test_line_prologue_forms(fe, ctx, oracle, knobs, monkeypatch):
    """Round 6 gave the line prologue new kernels that cover the common geometries only: the pre-blur evaluated inside the gradient kernel (k_lsd_grad_fused: w = 5m, sw = 4m,
    h = 5n, sh = 4n) and the counting sort on tile-sorted runs (k_lsd_hist_sort + k_lsd_scatter_runs: scaled images up to 2048 x 2048); the round-1-5 kernels remain behind them.
    LBD's walk rounds its coordinates with one conversion instruction on images of up to 16 384 pixels a side (k_lbd<true>); the previous instruction sequence remains for larger ones.
    Each older form (and the six-wave instantiation of the core for small calls) is forced here over frames of both kinds of geometry and compared with the oracle, as the default is everywhere else."""
    for k, v in knobs.items(): monkeypatch.setenv(k, v)
    for img, cap in [(synth_frame(2000), 200), (synth_frame(1235, w=1280, h=960), 400), (noise_frame(3, w=320, h=240), 200), (synth_frame(91, w=333, h=251), 200)]:
        _cmp_lines(fe, ctx, oracle, img, cap)


@pytest.mark.parametrize("flavour", ["cl", "lat", "thr"])
def test_lsd_core_flavours(fe, ctx, oracle, flavour, monkeypatch):
    """The sequential core has three launch forms (lsd_regions.h, lsd_cluster.h): cluster (main wave + helper waves on several compute units,
    results through global memory, monotonic pixel map: what calls of up to 64 frames get), the lone wave and the six-waves-per-SIMD throughput form.
    Each is forced here over frames that stress the helper protocol in different ways: long lines (helpers give up beyond their reach), 1280x960 (the cluster
    form's main wave keeps its private bitmap in global memory), noise (hundreds of one-pixel regions per chunk: result slots run out), a
    ramp (regions beyond every helper limit) and an odd size."""
    import ctypes as C
    monkeypatch.setenv("SSLAM_LSD_FLAVOUR", flavour)
    frames = [(synth_frame(2000), 200), (synth_frame(1235, w=1280, h=960), 400), (synth_frame(91, w=333, h=251), 200),
              (noise_frame(3, w=320, h=240), 200), (ramp_frame(), 200)]
    taken = 0
    for img, cap in frames:
        _cmp_lines(fe, ctx, oracle, img, cap)
        if flavour == "cl":
            ex = fe.LineExtractor(ctx, cap); ex(img)
            out = (C.c_longlong * 8)(); fe.lib().sslam_lines_debug_cycles(ex.h, 0, out); ex.close()
            taken += out[5] & 0xFFFFFFFF
            assert out[7] == 0, "the main wave gave up waiting for a helper"
    if flavour == "cl":
        assert taken > 1000, "the %s form took almost no region from its helpers: %d" % (flavour, taken)


@pytest.mark.parametrize("knobs", [{"SSLAM_CL_WGS": "2"}, {"SSLAM_CL_WGS": "16", "SSLAM_CL_WINDOW": "1600"}, {"SSLAM_CL_WINDOW": "-1"}, {"SSLAM_CL_WINDOW": "6"},
                                   {"SSLAM_CL_SMAP": "-1"}, {"SSLAM_CL_SMAP": "2", "SSLAM_CL_WGS": "3"}, {"SSLAM_CL_NO_FEEDER": "1"}])
def test_lsd_cluster_configurations(fe, ctx, oracle, knobs, monkeypatch):
    """three helpers .. forty-five, helpers far ahead of the main wave (stale views, refused results, second publications) or barely ahead, no
    helpers at all (every seed through the main wave's private growth + commit), no / coarser shared map, no feeder wave (every take through
    the global path): the schedule changes completely, the output must not"""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("SSLAM_LSD_FLAVOUR", "cl")
    for img, cap in [(synth_frame(2000), 200), (synth_frame(77, w=800, h=600), 300), (noise_frame(5, w=320, h=240), 200)]:
        _cmp_lines(fe, ctx, oracle, img, cap)


@pytest.mark.parametrize("nf", [7, 19, 41])
def test_lsd_cluster_batch(fe, ctx, oracle, nf):
    """up to 64 frames per call take the cluster form, one to eight frames per XCD (fewer helper workgroups per frame as they share its
    compute units): results per frame as for single calls"""
    frames = [synth_frame(3100 + i) for i in range(nf)]
    ex = fe.LineExtractor(ctx, 200)
    try:
        dev = torch.from_numpy(np.stack(frames)).cuda()
        nf, cap = len(frames), 256
        d_kl = torch.zeros(nf * cap * 68, dtype=torch.uint8, device="cuda"); d_ld = torch.zeros(nf * cap * 32, dtype=torch.uint8, device="cuda")
        d_fn = torch.zeros(nf * cap * 3, dtype=torch.float64, device="cuda"); d_n = torch.zeros(nf, dtype=torch.int32, device="cuda")
        ex.extract_batch_dev(dev, 640, 480, 640, 640 * 480, nf, d_kl, d_ld, d_fn, d_n, cap)
        torch.cuda.synchronize()
        import ctypes as C
        for i, f in enumerate(frames):
            okl, old, ofn, oraw = oracle.lines_extract(f, 200)
            np.testing.assert_array_equal(ex.debug_segments(i), oraw, err_msg="frame %d" % i)
            out = (C.c_longlong * 8)(); fe.lib().sslam_lines_debug_cycles(ex.h, i, out)
            assert out[7] == 0 and (out[5] & 0xFFFFFFFF) > 200, (i, out[5] & 0xFFFFFFFF, out[7])
    finally:
        ex.close()


@pytest.mark.parametrize("nf", [65, 130])
def test_lsd_lone_wave_batch(fe, ctx, oracle, nf):
    """65 .. 1023 frames per call take one lone wave per frame (round 5: the multi-wave form that covered 65 .. 256 is gone): results per frame as for single calls"""
    frames = [synth_frame(3000 + i) for i in range(13)]
    ex = fe.LineExtractor(ctx, 200)
    try:
        dev = torch.from_numpy(np.stack([frames[i % len(frames)] for i in range(nf)])).cuda()
        cap = 256
        d_kl = torch.zeros(nf * cap * 68, dtype=torch.uint8, device="cuda"); d_ld = torch.zeros(nf * cap * 32, dtype=torch.uint8, device="cuda")
        d_fn = torch.zeros(nf * cap * 3, dtype=torch.float64, device="cuda"); d_n = torch.zeros(nf, dtype=torch.int32, device="cuda")
        ex.extract_batch_dev(dev, 640, 480, 640, 640 * 480, nf, d_kl, d_ld, d_fn, d_n, cap)
        torch.cuda.synchronize()
        want = [oracle.lines_extract(f, 200)[3] for f in frames]
        for i in range(nf):
            np.testing.assert_array_equal(ex.debug_segments(i), want[i % len(frames)], err_msg="frame %d" % i)
    finally:
        ex.close()


@pytest.mark.parametrize("knob", [("SSLAM_NFA_FUSED", "0"), ("SSLAM_NFA_FUSED", "2")])
def test_nfa_stage_launch_forms(fe, ctx, oracle, knob, monkeypatch):
    """The NFA stage (rect_improve: count -> evaluate -> accept, five refinement stages) behind the core has two launch forms: 18 launches (calls of 65 .. 2047 frames)
    and one wave per frame in one launch (k_nfa_all: batches that fill the chip); calls of up to 64 frames run it NEXT TO the core (k_nfa_stream: tests/test_nfa_stream_gpu.py).
    Each against the oracle on frames whose rectangle counts differ by 10x.  SSLAM_NFA_STREAM=0 so that the single-frame calls below reach the forms under test."""
    monkeypatch.setenv("SSLAM_NFA_STREAM", "0")
    monkeypatch.setenv(*knob)
    for img, cap in [(synth_frame(2000), 200), (synth_frame(1235, w=1280, h=960), 400), (noise_frame(3, w=320, h=240), 200), (synth_frame(91, w=333, h=251), 40)]:
        _cmp_lines(fe, ctx, oracle, img, cap)


def test_lines_blur_variant_opencv_340(fe, ctx, oracle):
    """sslam_lines_set_blur_variant(1): LBD's 5x5 pre-blur with OpenCV 3.4.0's rounded taps (14 63 103 63 14).  LSD's own pre-blur has the same taps under both
    variants, so the segments and keylines are unchanged; the LBD bytes follow the oracle's variant 1 and come back when the variant is switched back."""
    for img, cap in ((synth_frame(2000), 200), (np.full((240, 320), 255, np.uint8), 40), (synth_frame(91, w=333, h=251), 40)):
        ex = fe.LineExtractor(ctx, cap)
        try:
            kl0, ld0, fn0 = ex(img)
            ex.set_blur_variant(1)
            kl1, ld1, fn1 = ex(img)
            try:
                oracle.set_gauss_variant(1)
                okl, old, ofn, oraw = oracle.lines_extract(img, cap)
            finally:
                oracle.set_gauss_variant(0)
            np.testing.assert_array_equal(ex.debug_segments(0), oraw)
            assert len(kl1) == len(okl) == len(kl0)
            same = kl1["angle"].view(np.uint32) == okl["angle"].view(np.uint32)
            np.testing.assert_array_equal(ld1[same], old[same]); np.testing.assert_array_equal(fn1, ofn)
            if len(kl0) > 20:
                assert (ld1 != ld0).any()
            ex.set_blur_variant(0)
            kl2, ld2, fn2 = ex(img)
            np.testing.assert_array_equal(ld2, ld0)
        finally:
            ex.close()
