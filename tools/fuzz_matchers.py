"""Randomised parity sweep of the matcher entry points (run through gpurun): random frame pairs, query jitter / radii / level
windows / occupancy / validity, duplicated queries (so that consecutive queries compete for the same feature and the
speculative projection kernel has to re-evaluate), stereo flags, rotation checks.  usage: python tools/fuzz_matchers.py [n] [seed]"""
import sys, time; sys.path.insert(0, 'tests')
import numpy as np
from synth import synth_frame, warp_prev


def main():
    import pkg, oracle_lib
    from test_match_gpu import _pseudo_feature_vectors
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
    fe = pkg.frontend(); ctx = fe.Context(0); orc = oracle_lib.Oracle()
    sc = orc.orb_params()[0].astype(np.float32)
    pairs = []
    for s in range(6):
        w, h = [(640, 480), (320, 240), (800, 300), (500, 640), (1000, 700), (256, 256)][s]
        cur = synth_frame(3000 + s, w, h); prev = warp_prev(cur)
        nf = [1000, 500, 1000, 1000, 2000, 300][s]
        kp1, d1 = orc.orb_extract(prev, nf); kp2, d2 = orc.orb_extract(cur, nf)
        kl1, ld1, _, _ = orc.lines_extract(prev, 200); kl2, ld2, _, _ = orc.lines_extract(cur, 200)
        pairs.append((w, h, kp1, d1, kp2, d2, kl1, ld1, kl2, ld2))
    bad = []; nchk = 0; voc = [None, None, None]
    t0 = time.time()
    for it in range(n_iter):
        w, h, kp1, d1, kp2, d2, kl1, ld1, kl2, ld2 = pairs[int(rng.integers(0, len(pairs)))]
        bounds = (0.0, float(w), 0.0, float(h))
        # ---- ORB projection search: queries = jittered prev keypoints, a random subset duplicated / shuffled
        sel = rng.integers(0, len(kp1), int(rng.integers(1, 2 * len(kp1))))
        if rng.random() < 0.5: sel = np.sort(sel)
        q = np.zeros(len(sel), fe.PQ_DTYPE); k = kp1[sel]
        jit = float(rng.choice([0.5, 1.5, 4.0]))
        q["u"] = k["x"] + 3 + rng.normal(0, jit, len(sel)); q["v"] = k["y"] - 2 + rng.normal(0, jit, len(sel))
        o = k["octave"]; mode = int(rng.integers(0, 2))
        rs = float(rng.choice([1.0, 2.5, 4.0, 15.0]))
        q["radius"] = rs * sc[o]
        q["min_level"] = o - 1; q["max_level"] = np.where(rng.random(len(sel)) < 0.2, -1, o + (mode == 1))
        q["angle"] = k["angle"]; q["valid"] = rng.random(len(sel)) < 0.95; q["obs_positive"] = rng.random(len(sel)) < float(rng.choice([0.0, 0.5, 1.0]))
        occ = (rng.random(len(kp2)) < float(rng.choice([0.0, 0.05, 0.5]))).astype(np.uint8)
        ur = None
        if rng.random() < 0.4:
            ur = np.where(rng.random(len(kp2)) < 0.5, kp2["x"] - rng.uniform(0, 40, len(kp2)), -1).astype(np.float32); q["ur"] = q["u"] - rng.uniform(0, 40, len(sel))
        ratio = float(rng.choice([0.6, 0.8, 0.9])); th = int(rng.choice([50, 100])); ori = bool(rng.integers(0, 2))
        qd = d1[sel]
        a, n = ctx.search_by_projection(0, mode, kp2, d2, q, qd, occ, ur, ratio, th, ori, bounds)
        oa, on = orc.search_by_projection(0, mode, kp2, d2, q, qd, occ, ur, ratio, th, ori, bounds)
        nchk += 1
        if n != on or not np.array_equal(a, oa): bad.append("proj it %d mode %d nq %d n %d/%d" % (it, mode, len(sel), n, on))
        # ---- Fuse search on a resident keyframe
        kf = ctx.frame_upload(0, kp2, d2, ur, bounds)
        chi2 = int(rng.integers(0, 2)); inv = (1.0 / (sc * sc)).astype(np.float32)
        q2 = q.copy(); q2["min_level"] = o - 1; q2["max_level"] = o
        bi, bd = kf.fuse_search(q2, qd, chi2, inv if chi2 else None)
        oi, od = orc.fuse_search(0, chi2, kp2, d2, q2, qd, ur, inv, bounds)
        if not (np.array_equal(bi, oi) and np.array_equal(bd, od)): bad.append("fuse it %d" % it)
        # ---- triangulation search
        pk, pf, ik, jf = _pseudo_feature_vectors(d1, d2, nbits=int(rng.integers(2, 7)))
        F12 = rng.normal(0, 1, (3, 3)).astype(np.float32); F12[2, 2] = 1
        free1 = (rng.random(len(kp1)) < 0.9).astype(np.uint8); free2 = (rng.random(len(kp2)) < 0.9).astype(np.uint8)
        ur1 = np.where(rng.random(len(kp1)) < 0.3, kp1["x"] - 5, -1).astype(np.float32)
        f1 = ctx.frame_upload(0, kp1, d1, ur1, bounds)
        ex, ey = float(rng.uniform(-500, w + 500)), float(rng.uniform(-500, h + 500))
        onlyst = bool(rng.random() < 0.2)
        m, nm = f1.search_for_triangulation(kf, free1, free2, pk, pf, ik, jf, F12, ex, ey, sc, (sc * sc).astype(np.float32), onlyst, ori)
        om, onm = orc.search_for_triangulation(kp1, d1, ur1, free1, kp2, d2, ur, free2, pk, pf, ik, jf, F12, ex, ey, sc, (sc * sc).astype(np.float32), onlyst, ori)
        if nm != onm or not np.array_equal(m, om): bad.append("tri it %d %d/%d" % (it, nm, onm))
        f1.close(); kf.close()
        # ---- SearchByBoW over the same pseudo vocabulary nodes
        validk = (rng.random(len(kp1)) < 0.9).astype(np.uint8); rb = float(rng.choice([0.6, 0.75, 0.9]))
        ab, nb = ctx.search_by_bow(kp1, d1, validk, kp2, d2, pk, pf, ik, jf, rb, ori)
        ob, onb = orc.search_by_bow(kp1, d1, validk, kp2, d2, pk, pf, ik, jf, rb, ori)
        if nb != onb or not np.array_equal(ab, ob): bad.append("bow it %d %d/%d" % (it, nb, onb))
        # ---- SearchByBoW(KF, KF): map-point masks on both sides, strict TH_LOW
        valid2 = (rng.random(len(kp2)) < float(rng.choice([0.5, 0.9, 1.0]))).astype(np.uint8)
        mk, nk = ctx.search_by_bow_keyframes(kp1, d1, validk, kp2, d2, valid2, pk, pf, ik, jf, rb, ori)
        omk, onk = orc.search_by_bow_keyframes(kp1, d1, validk, kp2, d2, valid2, pk, pf, ik, jf, rb, ori)
        if nk != onk or not np.array_equal(mk, omk): bad.append("bow kf-kf it %d %d/%d" % (it, nk, onk))
        # ---- relocalisation / loop-closing overloads of SearchByProjection against their own restatements
        pred = np.clip(o + rng.integers(-1, 2, len(sel)), 0, 7)
        qr = q.copy(); qr["radius"] = rs * sc[pred]; qr["min_level"] = pred - 1; qr["max_level"] = pred + 1; qr["obs_positive"] = 1
        qdm = qr.copy(); qdm["max_level"] = pred
        od_ = int(rng.choice([50, 64, 100]))
        a, n = ctx.search_by_projection(0, 1, kp2, d2, qr, qd, occ, None, 0.0, od_, ori, bounds)
        oa, on = orc.search_by_projection_reloc(kp2, d2, qdm, qd, occ, od_, ori, bounds)
        if n != on or not np.array_equal(a, oa): bad.append("reloc it %d %d/%d" % (it, n, on))
        qs = qr.copy(); qs["max_level"] = pred
        a, n = ctx.search_by_projection(0, 1, kp2, d2, qs, qd, occ, None, 0.0, 50, False, bounds)
        oa, on = orc.search_by_projection_sim3(0, kp2, d2, qdm, qd, occ, bounds)
        if n != on or not np.array_equal(a, oa): bad.append("sim3 proj it %d %d/%d" % (it, n, on))
        # ---- Frame::ComputeBoW through a text vocabulary (rebuilt now and then)
        if it % 25 == 0:
            import tempfile, os
            from synth import synthetic_vocab, write_vocab_text
            if voc[0] is not None: voc[0].close()
            L, ptr_, ch_, nd_, word_, weight_ = synthetic_vocab(rng, k=int(rng.integers(3, 11)), L=int(rng.integers(2, 5)))
            wg, scg = int(rng.integers(0, 4)), int(rng.integers(0, 6))
            with tempfile.TemporaryDirectory() as td:
                vp = os.path.join(td, "v.txt"); write_vocab_text(vp, 10, L, ptr_, ch_, nd_, weight_, scoring=scg, weighting=wg, weight_fmt="%.6g")
                voc[0] = fe.Vocabulary.from_text_file(ctx, vp); voc[1] = orc.vocab_load_text(vp); voc[2] = (wg, scg)
        ov = voc[1]; lu = int(rng.integers(0, 6))
        bow, fv = voc[0].compute_bow(d2, lu)
        bw, bv, fn, fp, ff = orc.compute_bow(ov["levels"], ov["child_ptr"], ov["children"], ov["node_desc"], ov["word_id"], ov["weight"], d2, lu, voc[2][0], voc[2][1])
        if list(bow.keys()) != bw.tolist() or list(bow.values()) != bv.tolist() or list(fv.keys()) != fn.tolist() or \
                any(fv[int(fn[j])] != ff[fp[j]:fp[j + 1]].tolist() for j in range(len(fn))): bad.append("compute_bow it %d" % it)
        # ---- lines: projection + knn + MAD gate
        if len(kl1) > 2 and len(kl2) > 2:
            ql = np.zeros(len(kl1), fe.PQ_DTYPE)
            ql["u"] = kl1["startPointX"] + 3; ql["v"] = kl1["startPointY"] - 2; ql["u2"] = kl1["endPointX"] + 3; ql["v2"] = kl1["endPointY"] - 2
            ql["radius"] = float(rng.choice([5.0, 15.0, 24.0])); ql["min_level"] = -1; ql["max_level"] = 0
            ql["valid"] = rng.random(len(kl1)) < 0.95; ql["obs_positive"] = rng.random(len(kl1)) < 0.8
            occl = (rng.random(len(kl2)) < 0.1).astype(np.uint8)
            a, n = ctx.search_by_projection(1, 0, kl2, ld2, ql, ld1, occl, None, 0.6, 100, True, bounds)
            oa, on = orc.search_by_projection(1, 0, kl2, ld2, ql, ld1, occl, None, 0.6, 100, True, bounds)
            if n != on or not np.array_equal(a, oa): bad.append("line proj it %d" % it)
            predl = rng.integers(0, 3, len(kl1)); qm = ql.copy(); qm["min_level"] = predl - 1; qm["max_level"] = predl; qm["obs_positive"] = 1
            a, n = ctx.search_by_projection(1, 1, kl2, ld2, qm, ld1, occl, None, 0.0, 50, False, bounds)
            oa, on = orc.search_by_projection_sim3(1, kl2, ld2, qm, ld1, occl, bounds)
            if n != on or not np.array_equal(a, oa): bad.append("line sim3 proj it %d %d/%d" % (it, n, on))
            gs = float(rng.choice([0.5, 0.1])); rm = bool(rng.integers(0, 2))
            p1, _, _ = ctx.line_match(ld1, ld2, gs, rm); p2, _, _ = orc.line_match(ld1, ld2, gs, rm)
            if not np.array_equal(p1, p2): bad.append("line match it %d" % it)
        # ---- distinctive descriptors on random observation sets
        sizes = rng.integers(0, 40, 50); ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        dd = d2[rng.integers(0, len(d2), ptr[-1])] ^ np.packbits(rng.random((ptr[-1], 256)) < 0.05, axis=1)
        if not np.array_equal(ctx.distinctive_descriptors(dd, ptr), orc.distinctive(dd, ptr)): bad.append("distinctive it %d" % it)
    print("fuzz_matchers: %d iterations in %.1f s, %d mismatches" % (n_iter, time.time() - t0, len(bad)))
    for b in bad[:30]: print("  MISMATCH", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
