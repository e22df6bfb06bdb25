import sys, os; sys.path.insert(0, 'tests'); os.environ["SSLAM_PROJ_STATS"] = "1"
import numpy as np
import pkg, oracle_lib
from synth import synth_frame, warp_prev
from test_match_gpu import _proj_queries
fe = pkg.frontend(); ctx = fe.Context(0); orc = oracle_lib.Oracle()
rng = np.random.default_rng(1)
cur = synth_frame(2003); prev = warp_prev(cur)
kp1, d1 = orc.orb_extract(prev, 1000); kp2, d2 = orc.orb_extract(cur, 1000)
sc = orc.orb_params()[0].astype(np.float32)
for mode in (0, 1):
    q = _proj_queries(fe, rng, kp1, 0, mode, sc); occ = np.zeros(len(kp2), np.uint8)
    print("mode", mode, "radius mean %.1f" % q["radius"].mean(), "valid", int(q["valid"].sum()))
    for _ in range(3): ctx.search_by_projection(0, mode, kp2, d2, q, d1, occ, None, 0.8, 100, True)
