"""GPU parity: Hamming matchers vs the oracle (bit-exact indices, distances, match arrays)."""
import numpy as np
import pytest
from synth import synth_frame, warp_prev

pytestmark = pytest.mark.gpu


def _rand_desc(rng, n, flip_from=None, flips=20):
    if flip_from is None:
        return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    d = flip_from[rng.integers(0, len(flip_from), size=n)].copy()
    bits = np.unpackbits(d, axis=1)
    for i in range(n):
        bits[i, rng.choice(256, size=rng.integers(0, flips), replace=False)] ^= 1
    return np.packbits(bits, axis=1)


@pytest.mark.parametrize("nq,nt", [(1, 2), (7, 1), (200, 200), (1000, 1003), (64, 4097)])
def test_knn2(ctx, oracle, nq, nt):
    rng = np.random.default_rng(nq * 7919 + nt)
    t = _rand_desc(rng, nt)
    q = _rand_desc(rng, nq, flip_from=t)
    idx, dist = ctx.hamming_knn2(q, t)
    oidx, odist = oracle.knn2(q, t)
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_array_equal(dist, odist)


def test_knn2_ties_lowest_index(ctx, oracle):
    t = np.zeros((130, 32), np.uint8)          # all identical -> every distance ties
    q = np.full((3, 32), 255, np.uint8)
    idx, dist = ctx.hamming_knn2(q, t)
    assert (idx == [[0, 1]] * 3).all() and (dist == 256).all()


def test_hamming_matrix(ctx, oracle):
    rng = np.random.default_rng(5)
    q = _rand_desc(rng, 300); t = _rand_desc(rng, 517)
    np.testing.assert_array_equal(ctx.hamming_matrix(q, t), oracle.hamming_matrix(q, t))
    z = np.zeros((1, 32), np.uint8); o = np.full((1, 32), 255, np.uint8)
    assert ctx.hamming_matrix(z, o)[0, 0] == 256


@pytest.mark.parametrize("seed,nfeat,ori", [(1234, 1000, True), (1240, 2000, True), (1241, 1000, False)])
def test_search_for_initialization(fe, ctx, oracle, seed, nfeat, ori):
    cur = synth_frame(seed)
    prev = warp_prev(cur)
    kp1, d1 = oracle.orb_extract(prev, nfeat)
    kp2, d2 = oracle.orb_extract(cur, nfeat)
    pm = np.stack([kp1["x"], kp1["y"]], axis=1).astype(np.float32)      # vbPrevMatched = F1 keypoint positions (Tracking.cc:340-342)
    m12, pmo, n = ctx.search_for_initialization(kp1, d1, kp2, d2, pm, 100, 0.9, ori)
    om12, opm, on = oracle.search_for_initialization(kp1, d1, kp2, d2, pm, 100, 0.9, ori)
    assert on > 20
    assert n == on
    np.testing.assert_array_equal(m12, om12)
    np.testing.assert_array_equal(pmo, opm)


@pytest.mark.parametrize("n1,n2,ratio", [(40, 40, False), (200, 187, False), (400, 400, True), (1, 2, False), (5, 1, False), (0, 9, False)])
def test_line_match(ctx, oracle, n1, n2, ratio):
    rng = np.random.default_rng(n1 * 31 + n2)
    t = _rand_desc(rng, n2)
    q = _rand_desc(rng, n1, flip_from=t, flips=60) if n1 and n2 else _rand_desc(rng, n1)
    pairs, mad, mad12 = ctx.line_match(q, t, 0.5, ratio)
    opairs, omad, omad12 = oracle.line_match(q, t, 0.5, ratio)
    np.testing.assert_array_equal(pairs, opairs)
    assert mad == omad and mad12 == omad12


def _proj_queries(fe, rng, feats1, kind, mode, scales):
    n = len(feats1)
    q = np.zeros(n, fe.PQ_DTYPE)
    if kind == 0:
        q["u"] = feats1["x"] + 3 + rng.normal(0, 1.5, n); q["v"] = feats1["y"] - 2 + rng.normal(0, 1.5, n)
        o = feats1["octave"]
        if mode == 1:      # TrackWithMotionModel: th 15, levels [o-1, o+1]  (Tracking.cc:1227)
            q["radius"] = 15 * scales[o]; q["min_level"] = o - 1; q["max_level"] = o + 1
            third = rng.integers(0, 3, n)                      # exercise the forward / backward level windows too
            q["max_level"] = np.where(third == 1, -1, q["max_level"]); q["min_level"] = np.where(third == 1, o, q["min_level"])
            q["min_level"] = np.where(third == 2, 0, q["min_level"]); q["max_level"] = np.where(third == 2, o, q["max_level"])
        else:              # SearchLocalPoints: r = 4 (or 2.5) * scale, levels [p-1, p]  (ORBmatcher.cc:66-71)
            q["radius"] = np.where(rng.random(n) < 0.3, 2.5, 4.0) * 3 * scales[o]; q["min_level"] = o - 1; q["max_level"] = o
        q["angle"] = feats1["angle"]
    else:
        q["u"] = feats1["startPointX"] + 3; q["v"] = feats1["startPointY"] - 2
        q["u2"] = feats1["endPointX"] + 3; q["v2"] = feats1["endPointY"] - 2
        q["radius"] = np.where(rng.random(n) < 0.5, 5.0, 8.0) * 3; q["min_level"] = -1; q["max_level"] = 0
    q["valid"] = rng.random(n) < 0.95
    q["obs_positive"] = rng.random(n) < 0.9
    return q


@pytest.mark.parametrize("mode,seed", [(0, 1234), (1, 1234), (0, 2003), (1, 2003)])
def test_orb_search_by_projection(fe, ctx, oracle, mode, seed):
    rng = np.random.default_rng(seed + mode)
    cur = synth_frame(seed); prev = warp_prev(cur)
    kp1, d1 = oracle.orb_extract(prev, 1000); kp2, d2 = oracle.orb_extract(cur, 1000)
    scales = oracle.orb_params()[0]
    q = _proj_queries(fe, rng, kp1, 0, mode, scales)
    occ = (rng.random(len(kp2)) < 0.05).astype(np.uint8)
    for uright in (None, np.where(rng.random(len(kp2)) < 0.3, kp2["x"] - rng.uniform(0, 40, len(kp2)), -1).astype(np.float32)):
        if uright is not None:
            q["ur"] = q["u"] - rng.uniform(0, 40, len(q))
        a, n = ctx.search_by_projection(0, mode, kp2, d2, q, d1, occ, uright, 0.8 if mode == 0 else 0.9, 100, True)
        oa, on = oracle.search_by_projection(0, mode, kp2, d2, q, d1, occ, uright, 0.8 if mode == 0 else 0.9, 100, True)
        assert on > 50 and n == on
        np.testing.assert_array_equal(a, oa)


def test_line_search_by_projection(fe, ctx, oracle):
    rng = np.random.default_rng(5)
    cur = synth_frame(2000); prev = warp_prev(cur)
    kl1, ld1, _, _ = oracle.lines_extract(prev, 200); kl2, ld2, _, _ = oracle.lines_extract(cur, 200)
    q = _proj_queries(fe, rng, kl1, 1, 0, None)
    occ = (rng.random(len(kl2)) < 0.05).astype(np.uint8)
    a, n = ctx.search_by_projection(1, 0, kl2, ld2, q, ld1, occ, None, 0.6, 100, True)
    oa, on = oracle.search_by_projection(1, 0, kl2, ld2, q, ld1, occ, None, 0.6, 100, True)
    assert on > 10 and n == on
    np.testing.assert_array_equal(a, oa)


def _pseudo_feature_vectors(d1, d2, nbits=5):
    """Stand-in for DBoW2's FeatureVector (the vocabulary file is not in the reference tree): node id = a few
    descriptor bits, so related descriptors mostly share a node.  Returns CSR lists over the shared nodes."""
    def node(d):
        return (d[:, 0].astype(np.int32) >> (8 - nbits)) | ((d[:, 7].astype(np.int32) >> 6) << nbits)
    n1, n2 = node(d1), node(d2)
    shared = sorted(set(n1.tolist()) & set(n2.tolist()))
    pk, pf, ik, jf = [0], [0], [], []
    for nd in shared:
        a = np.nonzero(n1 == nd)[0]; b = np.nonzero(n2 == nd)[0]      # ascending feature index, as DBoW2 fills them
        ik += a.tolist(); jf += b.tolist(); pk.append(len(ik)); pf.append(len(jf))
    return np.array(pk, np.int32), np.array(pf, np.int32), np.array(ik, np.int32), np.array(jf, np.int32)


@pytest.mark.parametrize("seed,ori,ratio", [(1234, True, 0.9), (2003, False, 0.75), (2004, True, 0.7)])
def test_search_by_bow(fe, ctx, oracle, seed, ori, ratio):
    rng = np.random.default_rng(seed)
    cur = synth_frame(seed); prev = warp_prev(cur)
    kp1, d1 = oracle.orb_extract(prev, 1000); kp2, d2 = oracle.orb_extract(cur, 1000)
    pk, pf, ik, jf = _pseudo_feature_vectors(d1, d2)
    valid = (rng.random(len(kp1)) < 0.9).astype(np.uint8)
    a, n = ctx.search_by_bow(kp1, d1, valid, kp2, d2, pk, pf, ik, jf, ratio, ori)
    oa, on = oracle.search_by_bow(kp1, d1, valid, kp2, d2, pk, pf, ik, jf, ratio, ori)
    assert on > 30 and n == on
    np.testing.assert_array_equal(a, oa)
