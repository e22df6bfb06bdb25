"""Timing of the matcher-side entry points (SURVEY.md §8 a14-a21 and the §8(f) rows) through the C ABI, host buffers in /
host buffers out, next to the CPU oracle on the same inputs.  Writes a table (stdout); run through gpurun."""
import sys, time; sys.path.insert(0, 'tests')
import numpy as np
import pkg, oracle_lib
from synth import synth_frame, warp_prev
from test_match_gpu import _proj_queries, _pseudo_feature_vectors
from synth import synthetic_vocab as _synthetic_vocab

fe = pkg.frontend(); ctx = fe.Context(0); orc = oracle_lib.Oracle()
rng = np.random.default_rng(1)
cur = synth_frame(2003); prev = warp_prev(cur)
kp1, d1 = orc.orb_extract(prev, 1000); kp2, d2 = orc.orb_extract(cur, 1000)
kl1, ld1, _, _ = orc.lines_extract(prev, 200); kl2, ld2, _, _ = orc.lines_extract(cur, 200)
sc = orc.orb_params()[0].astype(np.float32)


def t(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3


rows = []
def row(name, gpu, cpu, greps=20, creps=3):
    g = t(gpu, greps); c = t(cpu, creps)
    rows.append((name, g, c)); print("%-64s GPU %8.3f ms   CPU oracle %9.3f ms   x%.1f" % (name, g, c, c / g), flush=True)

pm = np.stack([kp1["x"], kp1["y"]], axis=1).astype(np.float32)
row("SearchForInitialization 1000x1000, window 100", lambda: ctx.search_for_initialization(kp1, d1, kp2, d2, pm.copy(), 100, 0.9, True),
    lambda: orc.search_for_initialization(kp1, d1, kp2, d2, pm.copy(), 100, 0.9, True))
row("knn-2 ORB 1000x1000", lambda: ctx.hamming_knn2(d1, d2), lambda: orc.knn2(d1, d2))
row("knn-2 + MAD gate LBD %dx%d" % (len(ld1), len(ld2)), lambda: ctx.line_match(ld1, ld2, 0.5, False), lambda: orc.line_match(ld1, ld2, 0.5, False))
for mode, nm in ((0, "SearchByProjection(F, MapPoints)"), (1, "SearchByProjection(F, LastFrame)")):
    q = _proj_queries(fe, rng, kp1, 0, mode, sc); occ = np.zeros(len(kp2), np.uint8)
    fr = ctx.frame_upload(0, kp2, d2)
    row(nm + " 1000 queries, host features", lambda: ctx.search_by_projection(0, mode, kp2, d2, q, d1, occ, None, 0.8, 100, True),
        lambda: orc.search_by_projection(0, mode, kp2, d2, q, d1, occ, None, 0.8, 100, True))
    row(nm + " 1000 queries, resident frame", lambda: fr.search_by_projection(mode, q, d1, occ, 0.8, 100, True),
        lambda: orc.search_by_projection(0, mode, kp2, d2, q, d1, occ, None, 0.8, 100, True))
    fr.close()
q = _proj_queries(fe, rng, kp1, 0, 0, sc); q["radius"] = 3.0 * sc[kp1["octave"]]; q["ur"] = q["u"] - 5
inv = (1.0 / (sc * sc)).astype(np.float32)
kf = ctx.frame_upload(0, kp2, d2, np.full(len(kp2), -1, np.float32))
row("Fuse candidate search, 1000 map points", lambda: kf.fuse_search(q, d1, 1, inv), lambda: orc.fuse_search(0, 1, kp2, d2, q, d1, np.full(len(kp2), -1, np.float32), inv))
pk, pf, ik, jf = _pseudo_feature_vectors(d1, d2)
F12 = np.array([[0, -1e-3, 0.3], [1e-3, 0, 2.0], [-0.3, -2.0, 1.0]], np.float32); free1 = np.ones(len(kp1), np.uint8); free2 = np.ones(len(kp2), np.uint8)
kf1 = ctx.frame_upload(0, kp1, d1); sg = (sc * sc).astype(np.float32)
row("SearchForTriangulation 1000x1000", lambda: kf1.search_for_triangulation(kf, free1, free2, pk, pf, ik, jf, F12, -2000.0, 300.0, sc, sg, False, True),
    lambda: orc.search_for_triangulation(kp1, d1, None, free1, kp2, d2, None, free2, pk, pf, ik, jf, F12, -2000.0, 300.0, sc, sg, False, True))
valid = np.ones(len(kp1), np.uint8)
row("SearchByBoW 1000x1000 (32-node pseudo vocabulary level)", lambda: ctx.search_by_bow(kp1, d1, valid, kp2, d2, pk, pf, ik, jf, 0.7, True),
    lambda: orc.search_by_bow(kp1, d1, valid, kp2, d2, pk, pf, ik, jf, 0.7, True))
sizes = rng.integers(3, 15, 2000); ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
dd = rng.integers(0, 256, (ptr[-1], 32), dtype=np.uint8)
row("ComputeDistinctiveDescriptors, 2000 map points x 3..14 observations", lambda: ctx.distinctive_descriptors(dd, ptr), lambda: orc.distinctive(dd, ptr))
L, vp, vc, vd, vw, vv = _synthetic_vocab(rng, 10, 5)
voc = fe.Vocabulary(ctx, L, vp, vc, vd, vw, vv)
row("ComputeBoW descent, 1000 features, 10^5-leaf vocabulary", lambda: voc.transform(kf, 4), lambda: orc.bow_transform(L, vp, vc, vd, vw, vv, d2, 4))
