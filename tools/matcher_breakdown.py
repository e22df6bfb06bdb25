"""Where a single matcher call spends its time: wall clock of the host entry point against the HIP-event time of its kernels
(sslam_profile_*): python tools/matcher_breakdown.py   (GPU)"""
import sys, os, time; sys.path.insert(0, 'tests')
import numpy as np
import pkg, oracle_lib
from synth import synth_frame, warp_prev
from test_match_gpu import _proj_queries, _pseudo_feature_vectors
fe = pkg.frontend(); ctx = fe.Context(0); orc = oracle_lib.Oracle()
pipeline = pkg._load("sslam_pipeline", os.path.join(pkg.PKG_DIR, "pipeline.py"))
rng = np.random.default_rng(1)
cur = synth_frame(2003); prev = warp_prev(cur)
kp1, d1 = orc.orb_extract(prev, 1000); kp2, d2 = orc.orb_extract(cur, 1000)
sc = orc.orb_params()[0].astype(np.float32)
pm = np.stack([kp1["x"], kp1["y"]], axis=1).astype(np.float32)
def measure(name, fn, reps=30):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    wall = (time.perf_counter() - t0) / reps * 1e3
    fe.lib().sslam_profile_enable(ctx.h, 1)
    for _ in range(reps): fn()
    fe.lib().sslam_profile_enable(ctx.h, 0)
    prof = pipeline.profile_drain(fe, ctx)
    print("%-56s wall %.3f ms | kernels %s" % (name, wall, {k: round(v[0] / reps, 3) for k, v in prof.items()}), flush=True)
measure("SearchForInitialization", lambda: ctx.search_for_initialization(kp1, d1, kp2, d2, pm.copy(), 100, 0.9, True))
for mode in (0, 1):
    q = _proj_queries(fe, rng, kp1, 0, mode, sc); occ = np.zeros(len(kp2), np.uint8)
    fr = ctx.frame_upload(0, kp2, d2)
    measure("SearchByProjection mode %d host features" % mode, lambda: ctx.search_by_projection(0, mode, kp2, d2, q, d1, occ, None, 0.8, 100, True))
    measure("SearchByProjection mode %d resident frame" % mode, lambda: fr.search_by_projection(mode, q, d1, occ, 0.8, 100, True))
    fr.close()
pk, pf, ik, jf = _pseudo_feature_vectors(d1, d2)
valid = np.ones(len(kp1), np.uint8)
measure("SearchByBoW", lambda: ctx.search_by_bow(kp1, d1, valid, kp2, d2, pk, pf, ik, jf, 0.7, True))
measure("knn2", lambda: ctx.hamming_knn2(d1, d2))
