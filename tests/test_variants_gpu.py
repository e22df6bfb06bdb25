"""GPU parity of the SELECTABLE decisions of the line path: the library under a variant equals the oracle under the same variant.

The arithmetic of LSD / LBD is OpenCV's (called at reference src/ExtractLineSegment.cpp:38-40,53) and this image holds no OpenCV, so where two restatements of a leaf are
plausible both exist, in the oracle (orc_set_*) AND in the library (sslam_lines_set_*), with the size of what each moves in oracle/ref_pin/pin_report_stub.json:
                           default            |  alternative
  D11 nfa()'s first term   (double(n) + 1)    |  log_gamma(n + 1)
  D12 LBD bit order        0x80 >> i          |  1 << i
  D7  LSD's 0.8x rescale   INTER_LINEAR_EXACT |  INTER_LINEAR
  D2  seed order in a bin  raster (stable)    |  the host's std::sort
Same bar as tests/test_lines_gpu.py: segments, keylines, line equations bit-equal; KeyLine.angle within 1 ulp; LBD bytes equal wherever the angle is."""
import numpy as np
import pytest
from synth import synth_frame, noise_frame

pytestmark = pytest.mark.gpu

VARIANTS = {          # library setter, oracle setter, default value
    "nfa": ("set_nfa_variant", "orc_set_lsd_nfa_variant", 1),
    "lbd_bits": ("set_lbd_bit_order", "orc_set_lbd_bit_order", 1),
    "resize": ("set_resize_variant", "orc_set_lsd_resize", 0),
    "seed_order": ("set_seed_order", "orc_set_lsd_seed_sort", 0),
}
FRAMES = [(lambda: synth_frame(2000), 200), (lambda: synth_frame(1235, w=1280, h=960), 400), (lambda: noise_frame(3, w=320, h=240), 200),
          (lambda: synth_frame(91, w=333, h=251), 40), (lambda: np.full((240, 320), 255, np.uint8), 40)]


def _ulp_diff(a, b):
    ai = a.view(np.int32).astype(np.int64); bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai); bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)


def _equal_to_oracle(ex, oracle, img, cap):
    kl, ld, fn = ex(img)
    okl, old, ofn, oraw = oracle.lines_extract(img, cap)
    np.testing.assert_array_equal(ex.debug_segments(0), oraw, err_msg="LSD segments (before top-N)")
    assert len(kl) == len(okl)
    for f in kl.dtype.names:
        if f == "angle": assert _ulp_diff(kl[f], okl[f]).max(initial=0) <= 1, "KeyLine.angle"
        else: np.testing.assert_array_equal(kl[f], okl[f], err_msg=f)
    same = kl["angle"].view(np.uint32) == okl["angle"].view(np.uint32)
    np.testing.assert_array_equal(ld[same], old[same])
    assert np.unpackbits(ld ^ old, axis=1).sum(axis=1).max(initial=0) <= 8
    np.testing.assert_array_equal(fn, ofn)
    return kl, ld, oraw


@pytest.mark.parametrize("which", sorted(VARIANTS))
def test_line_variant_equals_oracle_variant(fe, ctx, oracle, which):
    lib_setter, orc_setter, dflt = VARIANTS[which]
    alt = 1 - dflt
    moved = 0
    for make, cap in FRAMES:
        img = make()
        ex = fe.LineExtractor(ctx, cap)
        try:
            kl0, ld0, raw0 = _equal_to_oracle(ex, oracle, img, cap)                       # both on their defaults
            getattr(ex, lib_setter)(alt)
            assert getattr(oracle.L, orc_setter)(alt) == dflt, "oracle and library disagree on the default of " + which
            try: kl1, ld1, raw1 = _equal_to_oracle(ex, oracle, img, cap)
            finally: getattr(oracle.L, orc_setter)(dflt)
            if which == "lbd_bits":
                rev = np.array([int("{:08b}".format(i)[::-1], 2) for i in range(256)], np.uint8)
                np.testing.assert_array_equal(raw1, raw0); np.testing.assert_array_equal(ld1, rev[ld0])      # every byte bit-reversed, nothing else
                moved += int((ld1 != ld0).sum())
            else:
                moved += int(raw1.shape != raw0.shape or (raw1 != raw0).any())
            if which == "nfa" and len(raw1) > 20: assert len(raw0) > 1.3 * len(raw1), (len(raw0), len(raw1))        # nearly every rectangle passes under variant 1 (the default)
            getattr(ex, lib_setter)(dflt)
            _, ld2, raw2 = _equal_to_oracle(ex, oracle, img, cap)
            np.testing.assert_array_equal(raw2, raw0); np.testing.assert_array_equal(ld2, ld0)                # and back
        finally:
            ex.close()
    assert moved > 0, "the variant moved nothing on any frame: the switch is not connected"


def test_line_variants_combined_batch(fe, ctx, oracle):
    """the three alternatives that run on the device together (nfa 0, bit order 0, resize 1), through the batch entry (frames resident on the device), 9 frames in one call"""
    import torch
    frames = [synth_frame(3000 + i) for i in range(9)]
    ex = fe.LineExtractor(ctx, 200)
    try:
        ex.set_nfa_variant(0); ex.set_lbd_bit_order(0); ex.set_resize_variant(1)
        olds = [(s, getattr(oracle.L, s)(v)) for s, v in (("orc_set_lsd_nfa_variant", 0), ("orc_set_lbd_bit_order", 0), ("orc_set_lsd_resize", 1))]
        try:
            want = [oracle.lines_extract(f, 200) for f in frames]
        finally:
            for s, v in olds: getattr(oracle.L, s)(v)
        dev = torch.from_numpy(np.stack(frames)).cuda()
        n = len(frames)
        d_kl = torch.zeros((n, 200, 68), dtype=torch.uint8, device="cuda"); d_ld = torch.zeros((n, 200, 32), dtype=torch.uint8, device="cuda")
        d_fn = torch.zeros((n, 200, 3), dtype=torch.float64, device="cuda"); d_cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
        ex.extract_batch_dev(dev.data_ptr(), 640, 480, 640, 640 * 480, n, d_kl.data_ptr(), d_ld.data_ptr(), d_fn.data_ptr(), d_cnt.data_ptr(), 200)
        torch.cuda.synchronize()
        cnt = d_cnt.cpu().numpy()
        for i, (okl, old, ofn, oraw) in enumerate(want):
            assert cnt[i] == len(okl)
            np.testing.assert_array_equal(ex.debug_segments(i), oraw)
            kl = d_kl[i, :cnt[i]].cpu().numpy().reshape(-1).view(fe.KL_DTYPE)
            same = kl["angle"].view(np.uint32) == okl["angle"].view(np.uint32)
            np.testing.assert_array_equal(d_ld[i, :cnt[i]].cpu().numpy()[same], old[same])
            np.testing.assert_array_equal(d_fn[i, :cnt[i]].cpu().numpy(), ofn)
    finally:
        ex.close()
