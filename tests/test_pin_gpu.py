"""GPU parity against the REFERENCE ITSELF (not only the restatement): oracle/_ref/ref_orb_stub is /root/reference/src/ORBextractor.cc compiled
unmodified in the build container (oracle/ref_pin, stub cv:: layer over oracle/cvleaf.h, decisions D1 + D4 + D5 as the oracle defines them) and
oracle/_ref/libref_slices.so holds the reference's own SearchForInitialization / GetFeaturesInArea / SerachForInitialize bodies; both are
plain x86-64 binaries that travel with the snapshot.  Here the HIP library's outputs are compared with THEIR outputs, byte for byte.
Skipped (not failed) when the binaries did not travel: tests/test_pin_cpu.py is what must pass in the build container."""
import ctypes as C, os, subprocess, tempfile
import numpy as np
import pytest
import pkg
from oracle_lib import _p, KP_DTYPE
from synth import synth_frame, warp_prev

pytestmark = pytest.mark.gpu
REFDIR = os.path.join(pkg.ROOT, "oracle", "_ref")
BIN = os.path.join(REFDIR, "ref_orb_stub")
SLICES = os.path.join(REFDIR, "libref_slices.so")


def _ref_orb(img, nfeat, scale=1.2, nlevels=8, ini=20, mn=7):
    with tempfile.TemporaryDirectory() as d:
        pgm = os.path.join(d, "f.pgm")
        with open(pgm, "wb") as f:
            f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0])); f.write(np.ascontiguousarray(img, np.uint8).tobytes())
        subprocess.run([BIN, pgm, os.path.join(d, "o"), str(nfeat), repr(scale), str(nlevels), str(ini), str(mn)], check=True, capture_output=True)
        return np.fromfile(os.path.join(d, "o_kp.bin"), dtype=KP_DTYPE), np.fromfile(os.path.join(d, "o_desc.bin"), dtype=np.uint8).reshape(-1, 32)


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/ref_orb_stub did not travel (built by __graft_entry__.build() where /root/reference exists)")
@pytest.mark.parametrize("seed,w,h,nfeat,scale,nlevels,ini,mn", [(1234, 640, 480, 1000, 1.2, 8, 20, 7), (2003, 640, 480, 2000, 1.2, 8, 20, 7),
                                                                   (1235, 1280, 960, 2000, 1.2, 8, 20, 7), (99, 480, 360, 700, 1.5, 4, 20, 7),
                                                                   (4242, 640, 480, 1200, 1.2, 8, 35, 12)])
def test_hip_orb_equals_reference_compiled(fe, ctx, seed, w, h, nfeat, scale, nlevels, ini, mn):
    img = synth_frame(seed, w=w, h=h)
    rkp, rdesc = _ref_orb(img, nfeat, scale, nlevels, ini, mn)
    ex = fe.OrbExtractor(ctx, nfeat, scale, nlevels, ini, mn)
    try:
        kp, desc = ex(img)
    finally:
        ex.close()
    assert len(kp) == len(rkp) > 300
    np.testing.assert_array_equal(kp.view(np.uint8), rkp.view(np.uint8))
    np.testing.assert_array_equal(desc, rdesc)


@pytest.mark.skipif(not os.path.exists(SLICES), reason="oracle/_ref/libref_slices.so did not travel")
def test_hip_matchers_equal_reference_slices(fe, ctx):
    R = C.CDLL(SLICES)
    img = synth_frame(2000); prev = warp_prev(img)
    ex = fe.OrbExtractor(ctx, 2000, 1.2, 8, 20, 7); lx = fe.LineExtractor(ctx, 200)
    try:
        kp1, d1 = ex(prev); kp2, d2 = ex(img)
        _, l1, _ = lx(prev); _, l2, _ = lx(img)
    finally:
        ex.close(); lx.close()
    bb = np.array((0.0, 640.0, 0.0, 480.0), np.float32)
    for window, ratio, ori in [(100, 0.9, True), (30, 0.7, False)]:
        pm0 = np.stack([kp1["x"], kp1["y"]], 1).astype(np.float32)
        pm = pm0.copy(); m12 = np.full(len(kp1), -7, np.int32)
        n = R.ref_search_for_initialization(_p(kp1), _p(d1), len(kp1), _p(kp2), _p(d2), len(kp2), _p(pm), _p(m12), window, C.c_float(ratio), int(ori), _p(bb))
        gm, gpm, gn = ctx.search_for_initialization(kp1, d1, kp2, d2, pm0, window, ratio, ori)
        assert gn == n > 50
        np.testing.assert_array_equal(gm, m12); np.testing.assert_array_equal(gpm.view(np.uint32), pm.view(np.uint32))
    pr = np.zeros((len(l1) + 1, 2), np.int32); mad = C.c_double(); mad12 = C.c_double()
    nr = R.ref_line_search_for_initialize(_p(l1), len(l1), _p(l2), len(l2), _p(pr), len(l1) + 1, C.byref(mad), C.byref(mad12))
    gp = ctx.line_match(l1, l2, 0.5, False)
    pairs = gp[0]
    assert nr == len(pairs) > 20
    np.testing.assert_array_equal(pairs, pr[:nr])


DBOW2 = os.path.join(REFDIR, "libref_dbow2.so")


@pytest.mark.skipif(not os.path.exists(DBOW2), reason="oracle/_ref/libref_dbow2.so did not travel")
@pytest.mark.parametrize("k,L,weighting,scoring,levelsup", [(10, 4, 0, 0, 4), (9, 3, 1, 1, 2), (5, 5, 2, 3, 3)])
def test_hip_compute_bow_equals_vendored_dbow2(fe, ctx, oracle, tmp_path, k, L, weighting, scoring, levelsup):
    """Frame::ComputeBoW: the HIP vocabulary (sslam_vocab_load_text + sslam_compute_bow) against the reference's own vendored DBoW2 compiled whole
    (ORBVocabulary::loadFromTextFile + transform).  Files without a trailing newline and trees whose words are at least L - levelsup deep:
    outside that the reference reads uninitialised locals."""
    from synth import synthetic_vocab, write_vocab_text
    D2 = C.CDLL(DBOW2)
    rng = np.random.default_rng(7 * k + L)
    Lx, ptr, ch, nd, word, weight = synthetic_vocab(rng, k=k, L=L)
    path = tmp_path / "voc.txt"
    write_vocab_text(path, k, Lx, ptr, ch, nd, weight, scoring=scoring, weighting=weighting, weight_fmt="%r", trailing_newline=False)
    kp, d = oracle.orb_extract(synth_frame(1234), 1000)
    feat = np.ascontiguousarray(np.concatenate([d, nd[rng.integers(1, len(nd), 100)]])); n = len(feat)
    info = np.zeros(5, np.int32); bw = np.zeros(n, np.int32); bv = np.zeros(n, np.float64); fn = np.zeros(n, np.int32); fp = np.zeros(n + 1, np.int32); ff = np.zeros(n, np.int32)
    nb = C.c_int32(); nf = C.c_int32()
    assert D2.ref_vocab_compute_bow(str(path).encode(), _p(feat), n, levelsup, _p(info), _p(bw), _p(bv), C.byref(nb), _p(fn), _p(fp), _p(ff), C.byref(nf)) == 0
    voc = fe.Vocabulary.from_text_file(ctx, path)
    try:
        vi = voc.info()
        assert [vi["k"], vi["levels"], vi["scoring"], vi["weighting"], vi["nwords"]] == list(info)
        bow, fv = voc.compute_bow(feat, levelsup)
    finally:
        voc.close()
    assert list(bow.keys()) == bw[:nb.value].tolist() and np.array_equal(np.array(list(bow.values()), np.float64).view(np.uint64), bv[:nb.value].view(np.uint64)) and nb.value > 100
    ref_fv = {int(fn[j]): ff[fp[j]:fp[j + 1]].tolist() for j in range(nf.value)}
    assert {int(a): list(map(int, b)) for a, b in fv.items()} == ref_fv          # (every word of these trees is at least L - levelsup deep)
