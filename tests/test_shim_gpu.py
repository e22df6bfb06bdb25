"""GPU: the C++ drop-in surface (StructureSLAM::ORBextractor, LineSegment::ExtractLineSegment and the
matcher bodies) called the way Frame.cc / Tracking.cc call them, compared with the oracle."""
import os, subprocess, tempfile
import numpy as np
import pytest
import pkg
from synth import synth_frame, warp_prev, synthetic_vocab, write_vocab_text

pytestmark = pytest.mark.gpu


def test_cpp_shim_end_to_end(oracle):
    exe = pkg.builder().build_shim(force=False, verbose=False)
    cur = synth_frame(1234); prev = warp_prev(cur)
    with tempfile.TemporaryDirectory() as d:
        cur.tofile(os.path.join(d, "cur.raw")); prev.tofile(os.path.join(d, "prev.raw"))
        out = os.path.join(d, "o")
        rng = np.random.default_rng(21)
        L, ptr, ch, nd, word, weight = synthetic_vocab(rng, k=8, L=3)
        vpath = os.path.join(d, "voc.txt"); write_vocab_text(vpath, 8, L, ptr, ch, nd, weight, weight_fmt="%.6g")
        ov = oracle.vocab_load_text(vpath)
        r = subprocess.run([exe, os.path.join(d, "cur.raw"), "640", "480", os.path.join(d, "prev.raw"), out, "1000", "40", vpath], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        rd = lambda n, dt: np.fromfile(out + "_" + n + ".bin", dtype=dt)
        meta = rd("meta", np.int32)
        kp = rd("kp", np.uint8).reshape(-1, 28); desc = rd("desc", np.uint8).reshape(-1, 32)
        kl = rd("kl", np.uint8).reshape(-1, 68); ldesc = rd("ldesc", np.uint8).reshape(-1, 32); fn = rd("fn", np.float64).reshape(-1, 3)
        m12 = rd("m12", np.int32); lm = rd("lm", np.int32).reshape(-1, 2)
        bow = rd("bow", np.float64).reshape(-1, 2); fvflat = rd("fv", np.int32)
    okp, odesc = oracle.orb_extract(cur, 1000)
    okl, old, ofn, _ = oracle.lines_extract(cur, 40)          # reference cap 40 (src/ExtractLineSegment.cpp:42)
    assert meta[0] == len(okp) and meta[1] == 40 and meta[4] == 1 and meta[5] == 8
    np.testing.assert_array_equal(kp, okp.view(np.uint8).reshape(-1, 28))
    np.testing.assert_array_equal(desc, odesc)
    ka = kl.copy().view(np.uint8); kb = okl.view(np.uint8).reshape(-1, 68).copy()
    ka[:, 0:4] = 0; kb[:, 0:4] = 0                               # KeyLine.angle: atan2 (<= 1 ulp), compared in test_lines_gpu
    np.testing.assert_array_equal(ka, kb)
    assert np.unpackbits(ldesc ^ old, axis=1).sum(axis=1).max() <= 8
    np.testing.assert_array_equal(fn, ofn)
    kp1, d1 = oracle.orb_extract(prev, 1000)
    pm = np.stack([kp1["x"], kp1["y"]], axis=1).astype(np.float32)
    om12, _, on = oracle.search_for_initialization(kp1, d1, okp, odesc, pm, 100, 0.9, True)
    assert meta[2] == on
    np.testing.assert_array_equal(m12, om12)
    l1 = oracle.lines_extract(prev, 40)
    opairs, _, _ = oracle.line_match(l1[1], old, 0.5, False)
    assert meta[3] == len(opairs)
    np.testing.assert_array_equal(lm, opairs)
    nobs = min(len(odesc), 37)                                   # MapPoint::ComputeDistinctiveDescriptors through the shim
    assert meta[6] == oracle.distinctive(odesc[:nobs], np.array([0, nobs], np.int32))[0]

    # Frame::ComputeBoW through sslam_shim::ORBVocabulary (text file -> device tree -> BowVector / FeatureVector)
    assert meta[7] == ov["nwords"]
    bw, bv, fn_, fp, ff = oracle.compute_bow(ov["levels"], ov["child_ptr"], ov["children"], ov["node_desc"], ov["word_id"], ov["weight"], odesc, 2)
    np.testing.assert_array_equal(bow[:, 0], bw.astype(np.float64)); np.testing.assert_array_equal(bow[:, 1], bv)
    exp = []
    for j in range(len(fn_)):
        exp += [int(fn_[j]), int(fp[j + 1] - fp[j])] + ff[fp[j]:fp[j + 1]].tolist()
    assert fvflat.tolist() == exp and len(bw) > 100
