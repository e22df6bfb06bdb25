"""GPU parity: Hamming matchers vs the oracle (bit-exact indices, distances, match arrays)."""
import numpy as np
import pytest
from synth import synth_frame, warp_prev, synthetic_vocab, write_vocab_text

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


@pytest.mark.parametrize("cap", [1000, 96, 33])
def test_knn2_batch_dev_matrix_core_form(fe, ctx, oracle, cap):
    """sslam_hamming_knn2_batch_dev (the matrix-core form, match_knn.h): ragged row counts per frame (0, 1, 2, tile edges, the capacity),
    heavy ties (duplicated and all-equal rows), rows past a frame's count filled with junk -- against the oracle's knn2, frame by frame."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(cap)
    counts = [(0, 5), (5, 0), (1, 1), (3, 2), (31, 31), (32, 32), (33, 33), (64, 65), (cap, cap), (cap, 1), (7, cap), (min(cap, 40), min(cap, 95))]
    counts = [(min(a, cap), min(b, cap)) for a, b in counts]
    B = len(counts)
    q = rng.integers(0, 256, size=(B, cap, 32), dtype=np.uint8); t = rng.integers(0, 256, size=(B, cap, 32), dtype=np.uint8)      # junk beyond the counts
    for f, (a, b) in enumerate(counts):
        if b:
            t[f, :b] = _rand_desc(rng, b)
            if f % 3 == 0: t[f, :b] = t[f, rng.integers(0, max(1, b // 4), size=b)]      # many duplicated train rows: ties resolved by index
            if f % 4 == 1: t[f, :b] = 0
            if a: q[f, :a] = _rand_desc(rng, a, flip_from=t[f, :b], flips=6)
    dq = torch.from_numpy(q).cuda(); dt = torch.from_numpy(t).cuda()
    nq = torch.tensor([a for a, _ in counts], dtype=torch.int32).cuda(); nt = torch.tensor([b for _, b in counts], dtype=torch.int32).cuda()
    idx = torch.full((B, cap, 2), -7, dtype=torch.int32).cuda(); dist = torch.full((B, cap, 2), -7, dtype=torch.int32).cuda()
    _p = lambda x: C.c_void_p(x.data_ptr())
    rc = fe.lib().sslam_hamming_knn2_batch_dev(ctx.h, _p(dq), _p(nq), _p(dt), _p(nt), cap, B, _p(idx), _p(dist), None)
    assert rc == 0, fe.lib().sslam_last_error()
    torch.cuda.synchronize()
    idx = idx.cpu().numpy(); dist = dist.cpu().numpy()
    for f, (a, b) in enumerate(counts):
        if a == 0: continue
        if b == 0:
            assert (idx[f, :a] == -1).all() and (dist[f, :a] == -1).all()
            continue
        oi, od = oracle.knn2(q[f, :a], t[f, :b])
        np.testing.assert_array_equal(idx[f, :a], oi, err_msg="frame %d (%d x %d)" % (f, a, b))
        np.testing.assert_array_equal(dist[f, :a], od, err_msg="frame %d (%d x %d)" % (f, a, b))
        assert (idx[f, a:] == -7).all()          # rows past the frame's queries are not written


@pytest.mark.parametrize("cap,knob", [(4200, None), (1000, "popc"), (96, "mfma1")])
def test_knn2_batch_dev_other_forms(fe, ctx, oracle, cap, knob, monkeypatch):
    """the xor + popcount form (the only one beyond 4096 rows per frame; SSLAM_KNN2_BATCH=popc forces it below) and the 32-queries-per-wave matrix-core form
    (mfma1) against the oracle -- the knob is read on every call since round 5, so one process can compare the forms"""
    import ctypes as C
    import torch
    if knob: monkeypatch.setenv("SSLAM_KNN2_BATCH", knob)
    rng = np.random.default_rng(cap + 1)
    counts = [(cap, cap), (37, cap), (cap, 2), (0, 9), (65, 64)]
    B = len(counts)
    q = rng.integers(0, 256, size=(B, cap, 32), dtype=np.uint8); t = rng.integers(0, 256, size=(B, cap, 32), dtype=np.uint8)
    for f, (a, b) in enumerate(counts):
        t[f, :b] = _rand_desc(rng, b)
        if f == 0: t[f, :b] = t[f, rng.integers(0, max(1, b // 4), size=b)]
        if a: q[f, :a] = _rand_desc(rng, a, flip_from=t[f, :b], flips=6)
    dq = torch.from_numpy(q).cuda(); dt = torch.from_numpy(t).cuda()
    nq = torch.tensor([a for a, _ in counts], dtype=torch.int32).cuda(); nt = torch.tensor([b for _, b in counts], dtype=torch.int32).cuda()
    idx = torch.full((B, cap, 2), -7, dtype=torch.int32).cuda(); dist = torch.full((B, cap, 2), -7, dtype=torch.int32).cuda()
    _p = lambda x: C.c_void_p(x.data_ptr())
    # two calls on two streams of the one context: the second one's expansion must wait for the first one's search (the context's expand buffer)
    s2 = torch.cuda.Stream()
    idx2 = torch.full((B, cap, 2), -7, dtype=torch.int32).cuda(); dist2 = torch.full((B, cap, 2), -7, dtype=torch.int32).cuda()
    torch.cuda.synchronize()
    assert fe.lib().sslam_hamming_knn2_batch_dev(ctx.h, _p(dq), _p(nq), _p(dt), _p(nt), cap, B, _p(idx), _p(dist), None) == 0, fe.lib().sslam_last_error()
    assert fe.lib().sslam_hamming_knn2_batch_dev(ctx.h, _p(dt), _p(nt), _p(dq), _p(nq), cap, B, _p(idx2), _p(dist2), C.c_void_p(s2.cuda_stream)) == 0, fe.lib().sslam_last_error()
    ctx.synchronize(); torch.cuda.synchronize()
    idx = idx.cpu().numpy(); dist = dist.cpu().numpy(); idx2 = idx2.cpu().numpy(); dist2 = dist2.cpu().numpy()
    for f, (a, b) in enumerate(counts):
        for (na, nb, qq, tt, ii, dd) in ((a, b, q, t, idx, dist), (b, a, t, q, idx2, dist2)):
            if na == 0: continue
            if nb == 0:
                assert (ii[f, :na] == -1).all() and (dd[f, :na] == -1).all()
                continue
            oi, od = oracle.knn2(qq[f, :na], tt[f, :nb])
            np.testing.assert_array_equal(ii[f, :na], oi, err_msg="frame %d (%d x %d)" % (f, na, nb))
            np.testing.assert_array_equal(dd[f, :na], od, err_msg="frame %d (%d x %d)" % (f, na, nb))


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


@pytest.mark.parametrize("form", ["spec", "lds", "global"])
@pytest.mark.parametrize("seed", [3, 4])
def test_search_for_initialization_contention(fe, ctx, oracle, form, seed, monkeypatch):
    """The single call has three kernels (csrc/match_ordered.h: sixteen speculative waves -- the default --, one wave in LDS, one wave on global
    memory).  Contended input: every F2 keypoint is wanted by several F1 keypoints in a row (near-duplicate descriptors at nearly the same
    place), so that matches are taken over by later, closer keypoints (the un-match path) and keypoints of one speculative round collide on
    their best / second-best candidate (the serial replay with re-evaluation).  All three forms must equal the oracle's restatement of
    src/ORBmatcher.cc:408-523."""
    if form != "spec": monkeypatch.setenv("SSLAM_SFI_FORM", form)
    rng = np.random.default_rng(seed)
    n2 = 260
    kp2 = np.zeros(n2, fe.KP_DTYPE)
    kp2["x"] = rng.uniform(20, 620, n2).astype(np.float32); kp2["y"] = rng.uniform(20, 460, n2).astype(np.float32)
    kp2["octave"] = (rng.random(n2) < 0.15).astype(np.int32); kp2["angle"] = rng.uniform(0, 360, n2).astype(np.float32); kp2["size"] = 31
    d2 = _rand_desc(rng, n2)
    reps = 4
    src = np.repeat(np.arange(n2), reps); rng.shuffle(src[: len(src) // 2])          # half in contended runs, half scattered
    n1 = len(src)
    kp1 = np.zeros(n1, fe.KP_DTYPE)
    kp1["x"] = kp2["x"][src] + rng.uniform(-3, 3, n1).astype(np.float32); kp1["y"] = kp2["y"][src] + rng.uniform(-3, 3, n1).astype(np.float32)
    kp1["octave"] = (rng.random(n1) < 0.1).astype(np.int32); kp1["angle"] = (kp2["angle"][src] + rng.normal(0, 8, n1)).astype(np.float32) % np.float32(360); kp1["size"] = 31
    d1 = d2[src].copy()
    for i in range(n1):                                                            # 0..24 flipped bits: ties, takeovers and ratio-test failures
        for b in rng.integers(0, 256, int(rng.integers(0, 25))): d1[i, b >> 3] ^= np.uint8(1 << (b & 7))
    pm = np.stack([kp1["x"], kp1["y"]], axis=1).astype(np.float32)
    m12, pmo, n = ctx.search_for_initialization(kp1, d1, kp2, d2, pm, 100, 0.9, True)
    om12, opm, on = oracle.search_for_initialization(kp1, d1, kp2, d2, pm, 100, 0.9, True)
    assert on > 50 and n == on
    np.testing.assert_array_equal(m12, om12)
    np.testing.assert_array_equal(pmo, opm)


@pytest.mark.parametrize("form", ["lds", "global"])
def test_search_for_initialization_batch_forms(fe, ctx, oracle, form, monkeypatch):
    """sslam_orb_search_for_initialization_batch_dev, more than eight pairs per call: one wave per pair with the pair's level-0 features staged
    in LDS (round 4; capacity 3/8 of the rows), pairs with more level-0 features than that falling back to the global-memory body inside the
    same launch, and the global-memory kernel of rounds 1-3 (SSLAM_SFI_BATCH=global).  Every pair against the oracle."""
    import ctypes as C, torch
    if form == "global": monkeypatch.setenv("SSLAM_SFI_BATCH", "global")
    rng = np.random.default_rng(99)
    P, cap = 13, 700
    kp1 = np.zeros((P, cap), fe.KP_DTYPE); kp2 = np.zeros((P, cap), fe.KP_DTYPE)
    d1 = np.zeros((P, cap, 32), np.uint8); d2 = np.zeros((P, cap, 32), np.uint8)
    n1 = np.zeros(P, np.int32); n2 = np.zeros(P, np.int32)
    for p in range(P):
        a, b = int(rng.integers(50, cap + 1)), int(rng.integers(50, cap + 1))
        if p == 3: a = b = 0                                    # an empty pair
        if p == 5: a = cap                                      # rows full
        lvl0 = [0.2, 0.2, 0.2, 0.2, 0.9, 1.0, 0.2, 0.45, 0.2, 0.2, 1.0, 0.2, 0.05][p]      # pairs 4, 5, 7, 10: more level-0 features than the LDS capacity (262)
        kp2[p, :b]["x"] = rng.uniform(5, 635, b); kp2[p, :b]["y"] = rng.uniform(5, 475, b); kp2[p, :b]["octave"] = (rng.random(b) >= lvl0).astype(np.int32) * rng.integers(1, 8, b)
        kp2[p, :b]["angle"] = rng.uniform(0, 360, b); d2[p, :b] = _rand_desc(rng, b)
        src = rng.integers(0, max(b, 1), a)
        kp1[p, :a]["x"] = kp2[p, src]["x"] + rng.uniform(-4, 4, a) if b else 0; kp1[p, :a]["y"] = kp2[p, src]["y"] + rng.uniform(-4, 4, a) if b else 0
        kp1[p, :a]["octave"] = (rng.random(a) >= lvl0).astype(np.int32) * rng.integers(1, 8, a)
        kp1[p, :a]["angle"] = (kp2[p, src]["angle"] + rng.normal(0, 10, a)) % 360 if b else 0
        if b:
            d1[p, :a] = d2[p, src]
            flips = rng.integers(0, 256, (a, 20)); on = rng.random((a, 20)) < 0.6
            for i in range(a):
                for bit in flips[i][on[i]]: d1[p, i, bit >> 3] ^= np.uint8(1 << (bit & 7))
        n1[p], n2[p] = a, b
    pm = np.stack([kp1["x"], kp1["y"]], axis=2).astype(np.float32)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    g = dict(kp1=t(kp1.view(np.uint8)), d1=t(d1), n1=t(n1), kp2=t(kp2.view(np.uint8)), d2=t(d2), n2=t(n2), pm=t(pm))
    m12 = torch.full((P, cap), -7, dtype=torch.int32, device="cuda"); nm = torch.zeros(P, dtype=torch.int32, device="cuda")
    _p = lambda x: C.c_void_p(x.data_ptr())
    bounds = (C.c_float * 4)(0.0, 640.0, 0.0, 480.0)
    rc = fe.lib().sslam_orb_search_for_initialization_batch_dev(ctx.h, _p(g["kp1"]), _p(g["d1"]), _p(g["n1"]), _p(g["kp2"]), _p(g["d2"]), _p(g["n2"]), cap, P, _p(g["pm"]), _p(m12), _p(nm),
                                                                100, C.c_float(0.9), 1, bounds, C.c_void_p(0))
    assert rc == 0, fe.lib().sslam_last_error()
    ctx.synchronize()
    m12 = m12.cpu().numpy(); nm = nm.cpu().numpy(); pmo = g["pm"].cpu().numpy()
    total = 0
    for p in range(P):
        a, b = int(n1[p]), int(n2[p])
        om12, opm, on = oracle.search_for_initialization(kp1[p, :a], d1[p, :a], kp2[p, :b], d2[p, :b], pm[p, :a], 100, 0.9, True)
        assert nm[p] == on, (p, nm[p], on)
        np.testing.assert_array_equal(m12[p, :a], om12, err_msg="pair %d" % p); np.testing.assert_array_equal(pmo[p, :a], opm, err_msg="pair %d" % p)
        total += on
    assert total > 300


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


@pytest.mark.parametrize("form", ["two-kernel", "lds", "wave"])
@pytest.mark.parametrize("mode", [0, 1])
def test_orb_search_by_projection_contention(fe, ctx, oracle, form, mode, monkeypatch):
    """The window matcher has three device forms (csrc/match_ordered.h: per-query top-4 lists + ordered commit with parallel prefixes -- the
    default --, sixteen speculative waves in LDS, one wave).  Contended input: every query is repeated several times in a row with small
    jitter (so that consecutive queries want the same keypoint, lists run dry and the commit has to re-scan), a third of the map points has
    no observations (takes a keypoint without blocking it), a fifth of the keypoints is occupied from the start.  All forms must equal the
    oracle's restatement of src/ORBmatcher.cc:45-129 / :1331-1473."""
    if form != "two-kernel": monkeypatch.setenv("SSLAM_PROJ_FORM", form)
    rng = np.random.default_rng(77 + mode)
    cur = synth_frame(1234); prev = warp_prev(cur)
    kp1, d1 = oracle.orb_extract(prev, 500); kp2, d2 = oracle.orb_extract(cur, 1000)
    scales = oracle.orb_params()[0]
    sel = np.repeat(rng.permutation(len(kp1))[:180], 7)                       # runs of seven near-identical queries
    q = _proj_queries(fe, rng, kp1[sel], 0, mode, scales)
    q["radius"] *= 1.5
    q["obs_positive"] = rng.random(len(q)) < 0.67
    qd = d1[sel].copy()
    for i in range(len(qd)):
        for b in rng.integers(0, 256, int(rng.integers(0, 12))): qd[i, b >> 3] ^= np.uint8(1 << (b & 7))
    occ = (rng.random(len(kp2)) < 0.2).astype(np.uint8)
    a, n = ctx.search_by_projection(0, mode, kp2, d2, q, qd, occ, None, 0.8 if mode == 0 else 0.9, 100, True)
    oa, on = oracle.search_by_projection(0, mode, kp2, d2, q, qd, occ, None, 0.8 if mode == 0 else 0.9, 100, True)
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


# ---- the remaining SearchByProjection overloads are parameterisations of the ordered window matcher (mode 1: best only, a feature that
# holds a map point is skipped); each is checked against its own line-by-line restatement in the oracle
@pytest.mark.parametrize("seed,ori,orb_dist", [(1234, True, 100), (2003, False, 64), (2004, True, 50)])
def test_search_by_projection_relocalisation(fe, ctx, oracle, seed, ori, orb_dist):
    """ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist), src/ORBmatcher.cc:1475-1602 (Tracking::Relocalization)"""
    rng = np.random.default_rng(seed)
    cur = synth_frame(seed); prev = warp_prev(cur)
    kp1, d1 = oracle.orb_extract(prev, 1000); kp2, d2 = oracle.orb_extract(cur, 1000)
    scales = oracle.orb_params()[0]
    q = np.zeros(len(kp1), fe.PQ_DTYPE)
    q["u"] = kp1["x"] + 3 + rng.normal(0, 2.0, len(kp1)); q["v"] = kp1["y"] - 2 + rng.normal(0, 2.0, len(kp1))
    pred = np.clip(kp1["octave"] + rng.integers(-1, 2, len(kp1)), 0, 7)
    q["radius"] = 10 * scales[pred]; q["min_level"] = pred - 1; q["max_level"] = pred + 1        # th = 10 (Tracking.cc Relocalization), GetFeaturesInArea(u, v, radius, pred-1, pred+1)
    q["angle"] = kp1["angle"]; q["valid"] = rng.random(len(kp1)) < 0.9; q["obs_positive"] = 1      # a matched feature always blocks later queries (:1545-1546)
    has_mp = (rng.random(len(kp2)) < 0.3).astype(np.uint8)                                         # CurrentFrame.mvpMapPoints[i2] != NULL
    qd = q.copy(); qd["max_level"] = pred                                                           # the restatement takes the predicted level
    oa, on = oracle.search_by_projection_reloc(kp2, d2, qd, d1, has_mp, orb_dist, ori)
    a, n = ctx.search_by_projection(0, 1, kp2, d2, q, d1, has_mp, None, 0.0, orb_dist, ori)
    assert on > 50 and n == on, (n, on)
    np.testing.assert_array_equal(a, oa)
    assert (has_mp[a >= 0] == 0).all()


@pytest.mark.parametrize("kind,seed", [(0, 1234), (0, 2003), (1, 1234), (1, 2003)])
def test_search_by_projection_loop_closing(fe, ctx, oracle, kind, seed):
    """ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th), src/ORBmatcher.cc:293-406, and the line twin src/LSDmatcher.cpp:558-683"""
    rng = np.random.default_rng(seed + kind)
    cur = synth_frame(seed); prev = warp_prev(cur)
    if kind == 0:
        f1, d1 = oracle.orb_extract(prev, 1000); f2, d2 = oracle.orb_extract(cur, 1000)
        scales = oracle.orb_params()[0]
        q = np.zeros(len(f1), fe.PQ_DTYPE)
        q["u"] = f1["x"] + 3 + rng.normal(0, 2.0, len(f1)); q["v"] = f1["y"] - 2 + rng.normal(0, 2.0, len(f1))
        pred = np.clip(f1["octave"] + rng.integers(0, 2, len(f1)), 0, 7)
        q["radius"] = 10 * scales[pred]
    else:
        f1, d1, _, _ = oracle.lines_extract(prev, 200); f2, d2, _, _ = oracle.lines_extract(cur, 200)
        q = np.zeros(len(f1), fe.PQ_DTYPE)
        q["u"] = f1["startPointX"] + 3; q["v"] = f1["startPointY"] - 2; q["u2"] = f1["endPointX"] + 3; q["v2"] = f1["endPointY"] - 2
        pred = rng.integers(0, 3, len(f1))                                                          # level 2 leaves no octave-0 keyline in [pred-1, pred]
        q["radius"] = 10 * 1.2 ** pred
    q["min_level"] = pred - 1; q["max_level"] = pred; q["valid"] = rng.random(len(f1)) < 0.9; q["obs_positive"] = 1
    matched = (rng.random(len(f2)) < 0.25).astype(np.uint8)                                         # vpMatched[idx] != NULL on entry
    oa, on = oracle.search_by_projection_sim3(kind, f2, d2, q, d1, matched)
    a, n = ctx.search_by_projection(kind, 1, f2, d2, q, d1, matched, None, 0.0, 50, False)          # TH_LOW, no rotation check
    assert on > (50 if kind == 0 else 10) and n == on, (n, on)
    np.testing.assert_array_equal(a, oa)


@pytest.mark.parametrize("seed,ori,ratio", [(1234, True, 0.8), (2003, False, 0.75), (2005, True, 0.9)])
def test_search_by_bow_keyframes(fe, ctx, oracle, seed, ori, ratio):
    """ORBmatcher::SearchByBoW(KF1, KF2, vpMatches12), src/ORBmatcher.cc:525-658 (loop closing): map-point masks on both sides, strict TH_LOW"""
    rng = np.random.default_rng(seed)
    cur = synth_frame(seed); prev = warp_prev(cur)
    kp1, d1 = oracle.orb_extract(prev, 1000); kp2, d2 = oracle.orb_extract(cur, 1000)
    d2 = d2.copy()
    near = np.argmin(np.unpackbits(d1[:200, None, :] ^ d2[None, :, :], axis=2).sum(axis=2), axis=1)
    for i in range(0, 200, 7):                                             # plant matches at exactly TH_LOW = 50 bits (accepted by `<=`, rejected by `<`)
        row = d1[i].copy(); bits = np.unpackbits(row); flip = rng.choice(256, 50, replace=False); bits[flip] ^= 1; d2[near[i]] = np.packbits(bits)
    pk, pf, ik, jf = _pseudo_feature_vectors(d1, d2, nbits=3)
    v1 = (rng.random(len(kp1)) < 0.85).astype(np.uint8); v2 = (rng.random(len(kp2)) < 0.8).astype(np.uint8)
    m, n = ctx.search_by_bow_keyframes(kp1, d1, v1, kp2, d2, v2, pk, pf, ik, jf, ratio, ori)
    om, on = oracle.search_by_bow_keyframes(kp1, d1, v1, kp2, d2, v2, pk, pf, ik, jf, ratio, ori)
    assert on > 30 and n == on, (n, on)
    np.testing.assert_array_equal(m, om)
    got = m[m >= 0]
    assert (v2[got] == 1).all() and len(np.unique(got)) == len(got) and (v1[m >= 0] == 1).all()
    e, en = ctx.search_by_bow_keyframes(kp1[:0], d1[:0], v1[:0], kp2, d2, v2, np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert en == 0 and len(e) == 0


# ---- device-resident frames: same matchers, features uploaded once or never
@pytest.mark.parametrize("mode", [0, 1])
def test_frame_handle_search_by_projection(fe, ctx, oracle, mode):
    rng = np.random.default_rng(77 + mode)
    cur = synth_frame(2003); prev = warp_prev(cur)
    kp1, d1 = oracle.orb_extract(prev, 1000)
    scales = oracle.orb_params()[0]
    ex = fe.OrbExtractor(ctx, 1000)
    kp2, d2 = ex(cur)                                   # ExtractORB ...
    fr = fe.Frame(ctx, orb=ex)                          # ... its features stay on the device
    up = ctx.frame_upload(0, kp2, d2)                   # the same features uploaded once from the host
    assert len(fr) == len(kp2) == len(up)
    q = _proj_queries(fe, rng, kp1, 0, mode, scales)
    occ = (rng.random(len(kp2)) < 0.05).astype(np.uint8)
    oa, on = oracle.search_by_projection(0, mode, kp2, d2, q, d1, occ, None, 0.8, 100, True)
    for f in (fr, up):
        a, n = f.search_by_projection(mode, q, d1, occ, 0.8, 100, True)
        assert on > 50 and n == on
        np.testing.assert_array_equal(a, oa)
    ex(prev)                                            # a later extraction must not disturb the snapshot
    a, n = fr.search_by_projection(mode, q, d1, occ, 0.8, 100, True)
    np.testing.assert_array_equal(a, oa)
    fr.close(); up.close(); ex.close()


def test_frame_handle_lines_and_knn2(fe, ctx, oracle):
    rng = np.random.default_rng(9)
    cur = synth_frame(2000); prev = warp_prev(cur)
    lx = fe.LineExtractor(ctx, 200)
    kl1, ld1, _ = lx(prev); f1 = fe.Frame(ctx, lines=lx)
    kl2, ld2, _ = lx(cur); f2 = fe.Frame(ctx, lines=lx)
    assert len(f1) == len(kl1) and len(f2) == len(kl2)
    idx, dist = f1.knn2(f2)                             # BFMatcher::knnMatch(d1, d2, m, 2) on resident descriptors
    oi, od = oracle.knn2(ld1, ld2)
    np.testing.assert_array_equal(idx, oi); np.testing.assert_array_equal(dist, od)
    q = _proj_queries(fe, rng, kl1, 1, 0, None)
    occ = (rng.random(len(kl2)) < 0.05).astype(np.uint8)
    a, n = f2.search_by_projection(0, q, ld1, occ, 0.6, 100, True)
    oa, on = oracle.search_by_projection(1, 0, kl2, ld2, q, ld1, occ, None, 0.6, 100, True)
    assert on > 10 and n == on
    np.testing.assert_array_equal(a, oa)
    f1.close(); f2.close(); lx.close()


# ---- MapPoint / MapLine ::ComputeDistinctiveDescriptors
def test_distinctive_descriptors(ctx, oracle):
    rng = np.random.default_rng(31)
    sizes = [1, 2, 3, 4, 5, 17, 64, 65, 130, 0, 257, 2, 1024]
    sets = []
    for n in sizes:
        base = rng.integers(0, 256, (1, 32), dtype=np.uint8)
        d = np.repeat(base, n, axis=0)
        if n:
            flips = rng.random((n, 256)) < rng.uniform(0.02, 0.3, (n, 1))      # observations of one point: noisy copies
            d = d ^ np.packbits(flips, axis=1)
        sets.append(d)
    sets.append(np.repeat(rng.integers(0, 256, (1, 32), dtype=np.uint8), 9, axis=0))     # all equal: first row wins
    desc = np.concatenate(sets); ptr = np.concatenate([[0], np.cumsum([len(x) for x in sets])]).astype(np.int32)
    got = ctx.distinctive_descriptors(desc, ptr)
    want = oracle.distinctive(desc, ptr)
    np.testing.assert_array_equal(got, want)
    assert got[sizes.index(0)] == -1 and got[-1] == 0


def test_distinctive_descriptors_limits(fe, ctx):
    import ctypes as C
    d = np.zeros((1025, 32), np.uint8); ptr = np.array([0, 1025], np.int32); best = np.zeros(1, np.int32)
    rc = fe.lib().sslam_distinctive_descriptors(ctx.h, d.ctypes.data_as(C.c_void_p), ptr.ctypes.data_as(C.c_void_p), 1, best.ctypes.data_as(C.c_void_p))
    assert rc != 0 and b"1024" in fe.lib().sslam_last_error()


# ---- Fuse candidate search: ORBmatcher::Fuse x2, LSDmatcher::Fuse
@pytest.mark.parametrize("chi2,seed", [(1, 1234), (0, 1234), (1, 2003)])
def test_orb_fuse_search(fe, ctx, oracle, chi2, seed):
    rng = np.random.default_rng(seed + chi2)
    cur = synth_frame(seed); prev = warp_prev(cur)
    kp1, d1 = oracle.orb_extract(prev, 1000); kp2, d2 = oracle.orb_extract(cur, 1000)
    scales, _, _, inv_sigma2 = oracle.orb_params()[0], None, None, None
    sc = oracle.orb_params()[0].astype(np.float32)
    inv_sigma2 = (1.0 / (sc * sc)).astype(np.float32)           # KeyFrame::mvInvLevelSigma2
    q = _proj_queries(fe, rng, kp1, 0, 0, sc)
    q["radius"] = 3.0 * sc[kp1["octave"]]                        # Fuse: th = 3 (LocalMapping::SearchInNeighbors)
    q["ur"] = q["u"] - rng.uniform(0, 40, len(q))
    uright = np.where(rng.random(len(kp2)) < 0.4, kp2["x"] - rng.uniform(0, 40, len(kp2)), -1).astype(np.float32)
    kf = ctx.frame_upload(0, kp2, d2, uright)
    bi, bd = kf.fuse_search(q, d1, chi2, inv_sigma2 if chi2 else None)
    oi, od = oracle.fuse_search(0, chi2, kp2, d2, q, d1, uright, inv_sigma2)
    assert (oi >= 0).sum() > 100
    np.testing.assert_array_equal(bi, oi); np.testing.assert_array_equal(bd, od)
    kf.close()


def test_line_fuse_search(fe, ctx, oracle):
    rng = np.random.default_rng(11)
    cur = synth_frame(2000); prev = warp_prev(cur)
    kl1, ld1, _, _ = oracle.lines_extract(prev, 200); kl2, ld2, _, _ = oracle.lines_extract(cur, 200)
    q = _proj_queries(fe, rng, kl1, 1, 0, None)
    kf = ctx.frame_upload(1, kl2, ld2)
    bi, bd = kf.fuse_search(q, ld1, 0)
    oi, od = oracle.fuse_search(1, 0, kl2, ld2, q, ld1)
    assert (oi >= 0).sum() > 20
    np.testing.assert_array_equal(bi, oi); np.testing.assert_array_equal(bd, od)
    kf.close()


# ---- ORBmatcher::SearchForTriangulation
@pytest.mark.parametrize("seed,only_stereo,ori", [(1234, False, True), (2003, False, False), (2004, True, True)])
def test_search_for_triangulation(fe, ctx, oracle, seed, only_stereo, ori):
    rng = np.random.default_rng(seed)
    cur = synth_frame(seed); prev = warp_prev(cur)
    kp1, d1 = oracle.orb_extract(prev, 1000); kp2, d2 = oracle.orb_extract(cur, 1000)
    # the warp prev -> cur is a rigid image motion H; F12 = ([e2]x H)^T makes the epipolar line of kp1 pass through H kp1
    a = np.deg2rad(1.5); cx, cy = 319.5, 239.5
    H = np.array([[np.cos(a), np.sin(a), cx - 3.0 - cx * np.cos(a) - cy * np.sin(a)],
                  [-np.sin(a), np.cos(a), cy + 2.0 + cx * np.sin(a) - cy * np.cos(a)], [0, 0, 1.0]])
    ex, ey = -2000.0, 300.0
    E = np.array([[0, -1.0, ey], [1.0, 0, -ex], [-ey, ex, 0]])
    F12 = (E @ H).T
    F12 = (F12 / np.abs(F12).max()).astype(np.float32)
    sc = oracle.orb_params()[0].astype(np.float32); sg = (sc * sc).astype(np.float32)
    pk, pf, ik, jf = _pseudo_feature_vectors(d1, d2)
    free1 = (rng.random(len(kp1)) < 0.9).astype(np.uint8); free2 = (rng.random(len(kp2)) < 0.9).astype(np.uint8)
    p_st = 0.7 if only_stereo else 0.3
    ur1 = np.where(rng.random(len(kp1)) < p_st, kp1["x"] - 5, -1).astype(np.float32)
    ur2 = np.where(rng.random(len(kp2)) < p_st, kp2["x"] - 5, -1).astype(np.float32)
    f1 = ctx.frame_upload(0, kp1, d1, ur1); f2 = ctx.frame_upload(0, kp2, d2, ur2)
    m, n = f1.search_for_triangulation(f2, free1, free2, pk, pf, ik, jf, F12, ex, ey, sc, sg, only_stereo, ori)
    om, on = oracle.search_for_triangulation(kp1, d1, ur1, free1, kp2, d2, ur2, free2, pk, pf, ik, jf, F12, ex, ey, sc, sg, only_stereo, ori)
    assert on > 30 and n == on, (n, on)
    np.testing.assert_array_equal(m, om)
    f1.close(); f2.close()


# ---- DBoW2 vocabulary descent, Frame::ComputeBoW.  The ORBvoc file is an LFS pointer in the reference
# tree, so the tree here is synthetic: k-ary, L levels, clustered node descriptors, a few stopped words (weight 0) and a ragged branch.
@pytest.mark.parametrize("levelsup", [4, 2, 0, 6])
def test_bow_transform(fe, ctx, oracle, levelsup):
    rng = np.random.default_rng(5)
    L, ptr, ch, nd, word, weight = synthetic_vocab(rng)
    kp, d = oracle.orb_extract(synth_frame(1234), 1000)
    feat = np.concatenate([d, nd[rng.integers(1, len(nd), 200)]])        # real descriptors + exact node descriptors (distance-0 ties)
    voc = fe.Vocabulary(ctx, L, ptr, ch, nd, word, weight)
    w, v, n = voc.transform(feat, levelsup)
    ow, ov, on = oracle.bow_transform(L, ptr, ch, nd, word, weight, feat, levelsup)
    np.testing.assert_array_equal(w, ow); np.testing.assert_array_equal(v, ov); np.testing.assert_array_equal(n, on)
    assert (w >= 0).all() and len(np.unique(w)) > 300
    fr = ctx.frame_upload(0, kp, d)
    w2, v2, n2 = voc.transform(fr, levelsup)                               # Frame::ComputeBoW on a resident frame
    np.testing.assert_array_equal(w2, ow[:len(d)]); np.testing.assert_array_equal(n2, on[:len(d)])
    fr.close(); voc.close()


def test_search_by_bow_overlapping_nodes(fe, ctx, oracle):
    """DBoW2 files a feature under exactly one node; lists that share a frame feature between nodes make the order of the
    nodes matter, and the matcher then replays them in order (one workgroup) instead of one workgroup per node"""
    rng = np.random.default_rng(3)
    cur = synth_frame(2003); prev = warp_prev(cur)
    kp1, d1 = oracle.orb_extract(prev, 1000); kp2, d2 = oracle.orb_extract(cur, 1000)
    pk, pf, ik, jf = _pseudo_feature_vectors(d1, d2, nbits=4)
    # append to every node a few frame features of the NEXT node (so they are candidates twice)
    nf = []; npf = [0]
    for k in range(len(pf) - 1):
        own = jf[pf[k]:pf[k + 1]].tolist()
        nxt = jf[pf[k + 1]:pf[k + 2]][:3].tolist() if k + 2 < len(pf) else []
        nf += own + nxt; npf.append(len(nf))
    jf2 = np.array(nf, np.int32); pf2 = np.array(npf, np.int32)
    valid = (rng.random(len(kp1)) < 0.9).astype(np.uint8)
    a, n = ctx.search_by_bow(kp1, d1, valid, kp2, d2, pk, pf2, ik, jf2, 0.8, True)
    oa, on = oracle.search_by_bow(kp1, d1, valid, kp2, d2, pk, pf2, ik, jf2, 0.8, True)
    assert on > 30 and n == on
    np.testing.assert_array_equal(a, oa)


def _bow_sets(bw, bv, fn, fp, ff):
    return {int(w): float(v) for w, v in zip(bw, bv)}, {int(fn[j]): ff[fp[j]:fp[j + 1]].tolist() for j in range(len(fn))}


@pytest.mark.parametrize("weighting,scoring", [(0, 0), (1, 5), (2, 1), (3, 0)])
def test_vocabulary_text_file_and_compute_bow(fe, ctx, oracle, tmp_path, weighting, scoring):
    """System.cc:64-73 + Frame::ComputeBoW end to end: ORBvoc.txt-format file -> device tree -> BowVector / FeatureVector"""
    rng = np.random.default_rng(11)
    L, ptr, ch, nd, word, weight = synthetic_vocab(rng, k=9, L=3)
    path = tmp_path / "voc.txt"
    write_vocab_text(path, 9, L, ptr, ch, nd, weight, scoring=scoring, weighting=weighting, weight_fmt="%.6g")
    ov = oracle.vocab_load_text(path)
    voc = fe.Vocabulary.from_text_file(ctx, path)
    info = voc.info()
    assert info == dict(k=9, levels=L, scoring=scoring, weighting=weighting, nnodes=len(ptr) - 1, nwords=ov["nwords"])
    kp, d = oracle.orb_extract(synth_frame(1234), 1000)
    feat = np.concatenate([d, nd[rng.integers(1, len(nd), 100)]])
    w, v, n = voc.transform(feat, 2)
    ow, ovv, on = oracle.bow_transform(ov["levels"], ov["child_ptr"], ov["children"], ov["node_desc"], ov["word_id"], ov["weight"], feat, 2)
    np.testing.assert_array_equal(w, ow); np.testing.assert_array_equal(v, ovv); np.testing.assert_array_equal(n, on)
    bow, fv = voc.compute_bow(feat, 2)
    obow, ofv = _bow_sets(*oracle.compute_bow(ov["levels"], ov["child_ptr"], ov["children"], ov["node_desc"], ov["word_id"], ov["weight"], feat, 2, weighting, scoring))
    assert list(bow.keys()) == list(obow.keys()) and list(bow.values()) == list(obow.values())        # bit-equal doubles, same key order
    assert fv == ofv and len(bow) > 100 and len(fv) > 5
    assert sorted(sum(fv.values(), [])) == [i for i in range(len(feat)) if ovv[i] > 0]                 # stopped words drop their features
    if scoring != 5:
        norm = sum(abs(x) for x in bow.values()) if scoring != 1 else sum(x * x for x in bow.values()) ** 0.5
        assert abs(norm - 1.0) < 1e-12
    fr = ctx.frame_upload(0, kp, d)
    bow2, fv2 = voc.compute_bow(fr, 2)
    obow2, ofv2 = _bow_sets(*oracle.compute_bow(ov["levels"], ov["child_ptr"], ov["children"], ov["node_desc"], ov["word_id"], ov["weight"], d, 2, weighting, scoring))
    assert bow2 == obow2 and fv2 == ofv2
    e_bow, e_fv = voc.compute_bow(np.zeros((0, 32), np.uint8))
    assert e_bow == {} and e_fv == {}
    fr.close(); voc.close()


def test_vocabulary_text_file_rejects_garbage(fe, ctx, tmp_path):
    for name, text in (("hdr", "25 6 0 0\n0 1 " + "0 " * 32 + "1.0\n"), ("short", "10 3 0 0\n0 1 1 2 3 0.5\n"), ("parent", "10 3 0 0\n5 1 " + "0 " * 32 + "1.0\n")):
        p = tmp_path / (name + ".txt"); p.write_text(text)
        with pytest.raises(Exception):
            fe.Vocabulary.from_text_file(ctx, p)
    with pytest.raises(Exception):
        fe.Vocabulary.from_text_file(ctx, tmp_path / "missing.txt")
