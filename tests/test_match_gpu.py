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
