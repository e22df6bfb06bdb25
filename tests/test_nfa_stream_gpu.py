"""The NFA stage next to the cluster form of the core (k_nfa_stream, the default for calls of up to 64 frames since round 5; SSLAM_NFA_STREAM=0 puts it back behind the
core): consumer counts, patience settings, a core without helpers, small batches.  (Round 4 kept these behind SSLAM_TEST_EXPERIMENTAL; round 5's first GPU call ran them and the
whole suite with the knob exported -- profiles/r05a_pytest_gpu_SSLAM_NFA_STREAM_1.txt -- before the default was flipped.)"""
import os, sys
import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from synth import synth_frame, noise_frame
from test_lines_gpu import _cmp_lines            # the suite's comparison of one extraction with the oracle (segments, keylines, LBD bytes)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("knobs", [{"SSLAM_NFA_STREAM": "0"}, {"SSLAM_NFA_STREAM": "1"}, {"SSLAM_NFA_STREAM": "3"}, {"SSLAM_NFA_STREAM": "48"},
                                   {"SSLAM_NFA_STREAM": "1", "SSLAM_NFA_STREAM_TICKS": "0"},          # every consumer gives up at once: the launch behind the core does all of it
                                   {"SSLAM_NFA_STREAM": "1", "SSLAM_NFA_STREAM_TICKS": "20000"},      # 0.2 ms of patience: some of each
                                   {"SSLAM_NFA_STREAM": "1", "SSLAM_CL_WINDOW": "-1"}])               # no helpers: a slow core, consumers mostly waiting
def test_nfa_stage_next_to_the_core(fe, ctx, oracle, knobs, monkeypatch):
    """SSLAM_NFA_STREAM: the NFA stage on a second stream, on the rectangles the cluster form's main wave has published so far (csrc/lsd_nfa.h k_nfa_stream;
    the protocol as a thread model: tests/test_nfa_stream_proto_cpu.py).  Frames with 10x different rectangle counts, one with none, one larger than the LDS bitmap."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("SSLAM_LSD_FLAVOUR", "cl")
    for img, cap in [(synth_frame(2000), 200), (synth_frame(1235, w=1280, h=960), 400), (noise_frame(3, w=320, h=240), 200), (synth_frame(91, w=333, h=251), 40),
                     (np.full((240, 320), 255, np.uint8), 40)]:
        for rep in range(3):
            _cmp_lines(fe, ctx, oracle, img, cap)


def test_nfa_stage_next_to_the_core_small_batch(fe, ctx, oracle, monkeypatch):
    monkeypatch.setenv("SSLAM_NFA_STREAM", "1")
    frames = [synth_frame(3100 + i) for i in range(19)]
    ex = fe.LineExtractor(ctx, 200)
    try:
        dev = torch.from_numpy(np.stack(frames)).cuda()
        nf, cap = len(frames), 256
        d_kl = torch.zeros(nf * cap * 68, dtype=torch.uint8, device="cuda"); d_ld = torch.zeros(nf * cap * 32, dtype=torch.uint8, device="cuda")
        d_fn = torch.zeros(nf * cap * 3, dtype=torch.float64, device="cuda"); d_n = torch.zeros(nf, dtype=torch.int32, device="cuda")
        for rep in range(2):
            ex.extract_batch_dev(dev, 640, 480, 640, 640 * 480, nf, d_kl, d_ld, d_fn, d_n, cap)
            torch.cuda.synchronize()
            for i, f in enumerate(frames):
                okl, old, ofn, oraw = oracle.lines_extract(f, 200)
                np.testing.assert_array_equal(ex.debug_segments(i), oraw, err_msg="frame %d" % i)
    finally:
        ex.close()
