#!/usr/bin/env python3
"""`make -C oracle/ref_pin pin-stub`, second half: the reference's own matcher / grid function bodies -- cut by line range out of
src/ORBmatcher.cc, src/LSDmatcher.cpp and src/Frame.cc into oracle/_ref/libref_slices.so (stub_slam/stub_slam.h supplies the class
declarations) -- against the CPU oracle's restatements, on the same arrays.

    compare_slices.py <oracle/_ref> <report.json>

Covered: ORBmatcher::DescriptorDistance (:1650-1666) = LSDmatcher::DescriptorDistance (src/LSDmatcher.cpp:364-380), ComputeThreeMaxima
(:1604-1645) and SearchForInitialization (:408-523) over Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea
(src/Frame.cc:133-148, 462-472, 368-421); Frame::GetLinesInArea (:423-460); LSDmatcher::SerachForInitialize (src/LSDmatcher.cpp:257-284)
over Frame::lineDescriptorMAD (src/Frame.cc:190-215) and a stand-in cv::BFMatcher::knnMatch (the one OpenCV leaf left in this half)."""
import ctypes as C, json, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
import oracle_lib
from oracle_lib import _p
from synth import synth_frame, warp_prev, noise_frame

ref_dir, report_path = sys.argv[1], sys.argv[2]
R = C.CDLL(os.path.join(ref_dir, "libref_slices.so"))
orc = oracle_lib.Oracle()
O = orc.L
rng = np.random.Generator(np.random.PCG64(int(os.environ.get("SSLAM_PIN_SEED", "20260926"))))      # (SSLAM_PIN_SEED: other draws of the random cases; the committed report is the default seed's)
report = {"what": "reference function bodies (line-range slices, stub_slam.h declarations) vs the CPU oracle", "cases": {}, "all_equal": True}


def note(name, ok, **kw):
    report["cases"][name] = dict(equal=bool(ok), **kw); report["all_equal"] &= bool(ok)
    print("%-44s %s %s" % (name, "equal" if ok else "DIFFERENT", kw))


# --- DescriptorDistance -----------------------------------------------------------------------------------------------------------------
a = rng.integers(0, 256, (4000, 32), dtype=np.uint8); b = rng.integers(0, 256, (4000, 32), dtype=np.uint8)
b[:500] = a[:500]; b[500:1000] = a[500:1000] ^ 0xFF; b[1000:1500, 7:] = a[1000:1500, 7:]
dr = np.array([R.ref_descriptor_distance(_p(a[i]), _p(b[i])) for i in range(len(a))])
do = np.array([O.orc_descriptor_distance(_p(a[i]), _p(b[i])) for i in range(len(a))])
note("DescriptorDistance (ORB == LSD body)", np.array_equal(dr, do) and dr.min() == 0 and dr.max() == 256, pairs=len(a))

# --- frames ------------------------------------------------------------------------------------------------------------------------------
frames = {}
for name, img, nf in [("synth1234", synth_frame(1234), 1000), ("synth2000", synth_frame(2000), 2000), ("noise7", noise_frame(7), 1000),
                      ("big1235", synth_frame(1235, w=1280, h=960), 2000)]:
    prev = warp_prev(img)
    frames[name] = (img, orc.orb_extract(prev, nf), orc.orb_extract(img, nf))

# --- GetFeaturesInArea (every level-filter form the reference uses: (-1,-1), (l,l), (l-1,l+1), (0,l), (l,-1)) ------------------------------
nq = 0; ok = True
for name, (img, (kp1, d1), (kp2, d2)) in frames.items():
    h, w = img.shape
    for bounds in [(0.0, float(w), 0.0, float(h)), (-7.25, w + 3.5, -2.0, h + 11.75)]:
        bb = np.array(bounds, np.float32); out_r = np.zeros(len(kp2) + 1, np.int32); out_o = np.zeros(len(kp2) + 1, np.int32)
        for _ in range(400):
            x, y = rng.uniform(-60, w + 60), rng.uniform(-60, h + 60); r = float(rng.choice([1.0, 7.5, 15.0, 40.0, 100.0, 333.0]))
            lv = int(rng.integers(0, 8)); mn, mx = [(-1, -1), (lv, lv), (lv - 1, lv + 1), (0, lv), (lv, -1)][int(rng.integers(0, 5))]
            nr = R.ref_features_in_area(_p(kp2), len(kp2), _p(bb), C.c_float(x), C.c_float(y), C.c_float(r), mn, mx, _p(out_r))
            no = O.orc_features_in_area(_p(kp2), len(kp2), _p(bb), C.c_float(x), C.c_float(y), C.c_float(r), mn, mx, _p(out_o))
            ok &= nr == no and np.array_equal(out_r[:nr], out_o[:no]); nq += 1
note("Frame::GetFeaturesInArea + grid assignment", ok, queries=nq)

# --- SearchForInitialization ---------------------------------------------------------------------------------------------------------------
for name, (img, (kp1, d1), (kp2, d2)) in frames.items():
    h, w = img.shape; bb = np.array((0.0, float(w), 0.0, float(h)), np.float32)
    for window, ratio, ori in [(100, 0.9, 1), (100, 0.9, 0), (30, 0.7, 1), (250, 1.0, 1)]:
        pm0 = np.stack([kp1["x"], kp1["y"]], 1).astype(np.float32)
        res = []
        for L, fn in ((R, "ref_search_for_initialization"), (O, "orc_search_for_initialization")):
            pm = pm0.copy(); m12 = np.full(len(kp1), -7, np.int32)
            n = getattr(L, fn)(_p(kp1), _p(d1), len(kp1), _p(kp2), _p(d2), len(kp2), _p(pm), _p(m12), window, C.c_float(ratio), ori, _p(bb))
            # a second call on the updated vbPrevMatched, as Tracking::MonocularInitialization does frame after frame
            m12b = np.full(len(kp1), -7, np.int32)
            n2 = getattr(L, fn)(_p(kp1), _p(d1), len(kp1), _p(kp2), _p(d2), len(kp2), _p(pm), _p(m12b), window, C.c_float(ratio), ori, _p(bb))
            res.append((n, m12, pm.copy(), n2, m12b))
        (n_r, m_r, p_r, n2_r, mb_r), (n_o, m_o, p_o, n2_o, mb_o) = res
        eq = n_r == n_o and np.array_equal(m_r, m_o) and np.array_equal(p_r.view(np.uint32), p_o.view(np.uint32)) and n2_r == n2_o and np.array_equal(mb_r, mb_o)
        note("SearchForInitialization %s w=%d r=%.1f ori=%d" % (name, window, ratio, ori), eq, matches=int(n_r), level0=int((kp1["octave"] == 0).sum()))

# --- lines: GetLinesInArea, SerachForInitialize + lineDescriptorMAD -----------------------------------------------------------------------
for name in ("synth1234", "synth2000", "big1235"):
    img = frames[name][0]; prev = warp_prev(img)
    kl1, ld1, _, _ = orc.lines_extract(prev, 400); kl2, ld2, _, _ = orc.lines_extract(img, 400)
    out_r = np.zeros(len(kl2) + 1, np.int32); out_o = np.zeros(len(kl2) + 1, np.int32); ok = True; h, w = img.shape
    for _ in range(600):
        x1, y1, x2, y2 = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(0, w), rng.uniform(0, h); r = float(rng.choice([5.0, 20.0, 60.0, 200.0]))
        if rng.random() < 0.3: k = kl2[int(rng.integers(0, len(kl2)))]; x1, y1, x2, y2 = k["startPointX"] + 1, k["startPointY"] - 1, k["endPointX"] + 2, k["endPointY"]
        mn, mx = [(-1, -1), (0, 0), (0, 1), (1, -1)][int(rng.integers(0, 4))]
        a_ = [C.c_float(v) for v in (x1, y1, x2, y2, r)]
        nr = R.ref_lines_in_area(_p(kl2), len(kl2), *a_, mn, mx, _p(out_r)); no = O.orc_lines_in_area(_p(kl2), len(kl2), *a_, mn, mx, _p(out_o))
        ok &= nr == no and np.array_equal(out_r[:nr], out_o[:no])
    note("Frame::GetLinesInArea %s" % name, ok, queries=600, lines=len(kl2))
    for n1, n2 in [(len(ld1), len(ld2)), (40, 40), (7, 3), (1, 2)]:
        q, t = ld1[:n1], ld2[:n2]
        pr = np.zeros((n1 + 1, 2), np.int32); mad = C.c_double(); mad12 = C.c_double()
        nr = R.ref_line_search_for_initialize(_p(q), n1, _p(t), n2, _p(pr), n1 + 1, C.byref(mad), C.byref(mad12))
        po, omad, omad12 = orc.line_match(q, t, 0.5, False)
        eq = nr == len(po) and np.array_equal(pr[:nr], po) and mad.value == omad and mad12.value == omad12
        note("LSDmatcher::SerachForInitialize %s %dx%d" % (name, n1, n2), eq, pairs=int(nr), nn_mad=mad.value, nn12_mad=mad12.value)

# --- the tracking thread's projection matchers (src/ORBmatcher.cc:45-137, 1331-1473; src/LSDmatcher.cpp:22-141, 185-255) ---------------------
# The reference bodies run on stand-in Frame / MapPoint / MapLine objects (stub_slam.h); the oracle's restatement takes the QUERIES the reference's
# projection loops form, so they are formed here the way shim/ORBmatcher.h / LSDmatcher.h form them (float32, the reference's operation order).
import importlib.util
spec = importlib.util.spec_from_file_location("sslam_frontend_types", os.path.join(HERE, "..", "..", "structure-slam-pointline_amd", "frontend.py"))
fe = importlib.util.module_from_spec(spec); spec.loader.exec_module(fe)      # dtypes only (PQ_DTYPE): nothing of the HIP library is loaded
f32 = np.float32
MP = np.dtype([("inView", "<i4"), ("bad", "<i4"), ("level", "<i4"), ("nObs", "<i4"), ("viewCos", "<f4"), ("projX", "<f4"), ("projY", "<f4"), ("projXR", "<f4")])
ML = np.dtype([("inView", "<i4"), ("bad", "<i4"), ("level", "<i4"), ("nObs", "<i4"), ("viewCos", "<f4"), ("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4")])
sc = [f32(1.0)]
for _ in range(7): sc.append(f32(sc[-1] * f32(1.2)))
scale8 = np.array(sc, np.float32)


def decode(assigned, owner, state):
    """oracle `assigned` (query index, -1 untouched, -2 matched then pruned) -> the reference side's encoding"""
    return np.array([owner[a] if a >= 0 else -1 if a == -2 else (-2 if state[i] == 1 else -3 if state[i] == 2 else -1) for i, a in enumerate(assigned)], np.int32)


for name in ("synth1234", "synth2000", "big1235"):
    img, (kp1, d1), (kp2, d2) = frames[name]
    h, w = img.shape; bounds = (0.0, float(w), 0.0, float(h)); bb = np.array(bounds, np.float32)
    n1, n2 = len(kp1), len(kp2)
    state2 = rng.choice([0, 0, 0, 1, 2], n2).astype(np.uint8)
    for stereo in (False, True):
        ur2 = np.where(rng.random(n2) < 0.5, kp2["x"] - rng.uniform(2, 40, n2), -1).astype(np.float32) if stereo else np.full(n2, -1, np.float32)
        # -- SearchByProjection(F, vpMapPoints, th): Tracking::SearchLocalPoints (th = 1, 3 or 5)
        for th in (1.0, 3.0, 5.0):
            mp = np.zeros(n1, MP)
            mp["inView"] = rng.random(n1) < 0.85; mp["bad"] = rng.random(n1) < 0.05; mp["level"] = kp1["octave"]; mp["nObs"] = rng.integers(0, 3, n1)
            mp["viewCos"] = np.where(rng.random(n1) < 0.5, 0.9995, 0.97); mp["projX"] = kp1["x"] + rng.uniform(-3, 3, n1); mp["projY"] = kp1["y"] + rng.uniform(-3, 3, n1)
            mp["projXR"] = mp["projX"] - rng.uniform(2, 40, n1)
            out_r = np.zeros(n2, np.int32)
            nr = R.ref_search_by_projection_mappoints(_p(kp2), _p(d2), n2, _p(bb), _p(scale8), _p(ur2), _p(state2), _p(mp), _p(d1), n1, C.c_float(th), C.c_float(0.8), _p(out_r))
            q = np.zeros(n1, fe.PQ_DTYPE)
            for i in range(n1):
                r0 = f32(2.5) if mp["viewCos"][i] > f32(0.998) else f32(4.0)
                if th != 1.0: r0 = f32(r0 * f32(th))
                lv = int(mp["level"][i])
                q[i] = (mp["projX"][i], mp["projY"][i], 0, 0, f32(r0 * sc[lv]), lv - 1, lv, 0.0, mp["projXR"][i], int(mp["inView"][i] and not mp["bad"][i]), int(mp["nObs"][i] > 0))
            a, no = orc.search_by_projection(0, 0, kp2, d2, q, d1, occupied=(state2 == 1).astype(np.uint8), uright=ur2, nnratio=0.8, th_dist=100, check_orientation=False, bounds=bounds)[:2]
            note("ORBmatcher::SearchByProjection(F, MapPoints) %s th=%g stereo=%d" % (name, th, stereo), nr == no and np.array_equal(out_r, decode(a, list(range(n1)), state2)), matches=int(nr))
    # -- SearchByProjection(CurrentFrame, LastFrame, th, bMono): poses, intrinsics, world points behind the last frame's keypoints
    cz, sz = f32(0.99995), f32(0.0099998)
    TL = np.eye(4, dtype=np.float32)
    for pose, (tz, mb) in enumerate([(-0.05, 0.01), (0.06, 0.01), (0.002, 0.05)]):       # forward / backward / neither (when not mono)
        TC = np.array([[cz, -sz, 0, 0.012], [sz, cz, 0, -0.007], [0, 0, 1, tz], [0, 0, 0, 1]], np.float32)
        cam = np.array([520.9 * w / 640, 521.0 * h / 480, 325.1 * w / 640, 249.7 * h / 480, 40.0, mb], np.float32)
        fx, fy, cx, cy, mbf, mbv = (f32(v) for v in cam)
        z = (2.0 + 0.35 * (np.arange(n1) % 7)).astype(np.float32)
        wp = np.stack([(kp1["x"] - cx) / fx * z, (kp1["y"] - cy) / fy * z, np.where(np.arange(n1) % 61 == 0, -z, z)], 1).astype(np.float32)
        has1 = (rng.random(n1) < 0.75).astype(np.uint8); out1 = (rng.random(n1) < 0.1).astype(np.uint8); nobs1 = rng.integers(0, 3, n1).astype(np.int32)

        def rx_plus_t(T, x):
            return [f32(f32(f32(f32(T[r, 0] * x[0]) + f32(T[r, 1] * x[1])) + f32(T[r, 2] * x[2])) + T[r, 3]) for r in range(3)]
        twc = [f32(f32(f32(f32(-TC[0, r]) * TC[0, 3]) + f32(f32(-TC[1, r]) * TC[1, 3])) + f32(f32(-TC[2, r]) * TC[2, 3])) for r in range(3)]
        tlc = rx_plus_t(TL, twc)
        for bMono in (True, False):
            fwd = bool(tlc[2] > mbv) and not bMono; bwd = bool(-tlc[2] > mbv) and not bMono
            lev = lambda o: (o, -1) if fwd else (0, o) if bwd else (o - 1, o + 1)
            ur2 = np.full(n2, -1, np.float32) if bMono else np.where(rng.random(n2) < 0.5, kp2["x"] - rng.uniform(2, 40, n2), -1).astype(np.float32)
            th = 15.0 if bMono else 7.0
            out_r = np.zeros(n2, np.int32)
            nr = R.ref_track_points(_p(kp1), _p(d1), n1, _p(has1), _p(out1), _p(nobs1), _p(wp), _p(kp2), _p(d2), n2, _p(state2), _p(ur2), _p(bb), _p(scale8),
                                    _p(cam), _p(TL), _p(TC), C.c_float(th), int(bMono), C.c_float(0.9), _p(out_r))
            q = []; qd = []; owner = []
            for i in range(n1):
                if not has1[i] or out1[i]: continue
                xc, yc, zc = rx_plus_t(TC, wp[i])
                invzc = f32(1.0 / np.float64(zc))
                if invzc < 0: continue
                u = f32(f32(f32(fx * xc) * invzc) + cx); v = f32(f32(f32(fy * yc) * invzc) + cy)
                if u < 0 or u > f32(w) or v < 0 or v > f32(h): continue
                o = int(kp1["octave"][i]); lo, hi = lev(o)
                q.append((u, v, 0, 0, f32(f32(th) * sc[o]), lo, hi, f32(kp1["angle"][i]), f32(u - f32(mbf * invzc)), 1, int(nobs1[i] > 0))); qd.append(d1[i]); owner.append(i)
            a, no = orc.search_by_projection(0, 1, kp2, d2, np.array(q, fe.PQ_DTYPE), np.stack(qd), occupied=(state2 == 1).astype(np.uint8), uright=ur2, nnratio=0.9, th_dist=100,
                                             check_orientation=True, bounds=bounds)[:2]
            note("ORBmatcher::SearchByProjection(Cur, Last) %s pose=%d mono=%d" % (name, pose, bMono), nr == no and np.array_equal(out_r, decode(a, owner, state2)),
                 matches=int(nr), queries=len(q), pruned=int((a == -2).sum()), fwd=fwd, bwd=bwd)
            # -- the line twin (src/LSDmatcher.cpp:22-141)
            kl1, ld1 = orc.lines_extract(warp_prev(img), 200)[:2]; kl2, ld2 = orc.lines_extract(img, 200)[:2]
            nl1, nl2 = len(kl1), len(kl2)
            zl = 2.5 + 0.4 * (np.arange(nl1) % 5)
            wl = np.stack([(kl1["startPointX"] - cx) / fx * zl, (kl1["startPointY"] - cy) / fy * zl, np.where(np.arange(nl1) % 17 == 3, -zl, zl),
                           (kl1["endPointX"] - cx) / fx * zl, (kl1["endPointY"] - cy) / fy * zl, zl], 1).astype(np.float64)
            hasl = (rng.random(nl1) < 0.8).astype(np.uint8); badl = (rng.random(nl1) < 0.08).astype(np.uint8); outl = (rng.random(nl1) < 0.1).astype(np.uint8)
            nobsl = rng.integers(0, 3, nl1).astype(np.int32); statel = rng.choice([0, 0, 1, 2], nl2).astype(np.uint8)
            out_l = np.zeros(nl2, np.int32)
            nrl = R.ref_track_lines(_p(kp1), n1, _p(kl1), _p(ld1), nl1, _p(hasl), _p(badl), _p(outl), _p(nobsl), _p(wl), _p(kl2), _p(ld2), nl2, _p(statel), _p(bb), _p(scale8),
                                    _p(cam), _p(TL), _p(TC), C.c_float(th), int(bMono), C.c_float(0.6), _p(out_l))
            q = []; qd = []; owner = []
            for i in range(nl1):
                if not hasl[i] or badl[i] or outl[i]: continue
                s3 = rx_plus_t(TC, [f32(wl[i, k]) for k in range(3)]); e3 = rx_plus_t(TC, [f32(wl[i, 3 + k]) for k in range(3)])
                if s3[2] < 0 or e3[2] < 0: continue
                iz1 = f32(f32(1.0) / s3[2]); u1 = f32(f32(f32(fx * s3[0]) * iz1) + cx); v1 = f32(f32(f32(fy * s3[1]) * iz1) + cy)
                if u1 < 0 or u1 > f32(w) or v1 < 0 or v1 > f32(h): continue
                iz2 = f32(f32(1.0) / e3[2]); u2 = f32(f32(f32(fx * e3[0]) * iz2) + cx); v2 = f32(f32(f32(fy * e3[1]) * iz2) + cy)
                if u2 < 0 or u2 > f32(w) or v2 < 0 or v2 > f32(h): continue
                o = int(kp1["octave"][i]); lo, hi = lev(o)
                q.append((u1, v1, u2, v2, f32(f32(th) * sc[o]), lo, hi, 0.0, 0.0, 1, int(nobsl[i] > 0))); qd.append(ld1[i]); owner.append(i)
            if q:
                a, nol = orc.search_by_projection(1, 0, kl2, ld2, np.array(q, fe.PQ_DTYPE), np.stack(qd), occupied=(statel == 1).astype(np.uint8), uright=None, nnratio=0.6, th_dist=100,
                                                  check_orientation=False, bounds=bounds)[:2]
                exp = decode(a, owner, statel)
            else:
                nol, exp = 0, decode(np.full(nl2, -1), owner, statel)
            note("LSDmatcher::SearchByProjection(Cur, Last) %s pose=%d mono=%d" % (name, pose, bMono), nrl == nol and np.array_equal(out_l, exp), matches=int(nrl), queries=len(q))
    # -- LSDmatcher::SearchByProjection(F, vpMapLines, th)
    kl1, ld1 = orc.lines_extract(warp_prev(img), 200)[:2]; kl2, ld2 = orc.lines_extract(img, 200)[:2]
    nl1, nl2 = len(kl1), len(kl2)
    statel = rng.choice([0, 0, 1, 2], nl2).astype(np.uint8)
    for th in (1.0, 3.0):
        ml = np.zeros(nl1, ML)
        ml["inView"] = rng.random(nl1) < 0.85; ml["bad"] = rng.random(nl1) < 0.05; ml["level"] = rng.integers(0, 3, nl1); ml["nObs"] = rng.integers(0, 3, nl1)
        ml["viewCos"] = np.where(rng.random(nl1) < 0.5, 0.9995, 0.97)
        ml["x1"] = kl1["startPointX"] + 2; ml["y1"] = kl1["startPointY"] - 1; ml["x2"] = kl1["endPointX"] + 2; ml["y2"] = kl1["endPointY"] - 1
        out_l = np.zeros(nl2, np.int32)
        nrl = R.ref_line_search_by_projection_maplines(_p(kl2), _p(ld2), nl2, _p(scale8), _p(statel), _p(ml), _p(ld1), nl1, C.c_float(th), C.c_float(0.6), _p(out_l))
        q = np.zeros(nl1, fe.PQ_DTYPE)
        for i in range(nl1):
            r0 = f32(5.0) if ml["viewCos"][i] > f32(0.998) else f32(8.0)
            if th != 1.0: r0 = f32(r0 * f32(th))
            lv = int(ml["level"][i])
            q[i] = (ml["x1"][i], ml["y1"][i], ml["x2"][i], ml["y2"][i], f32(r0 * sc[lv]), lv - 1, lv, 0.0, 0.0, int(ml["inView"][i] and not ml["bad"][i]), int(ml["nObs"][i] > 0))
        a, nol = orc.search_by_projection(1, 0, kl2, ld2, q, ld1, occupied=(statel == 1).astype(np.uint8), uright=None, nnratio=0.6, th_dist=100, check_orientation=False, bounds=bounds)[:2]
        note("LSDmatcher::SearchByProjection(F, MapLines) %s th=%g" % (name, th), nrl == nol and np.array_equal(out_l, decode(a, list(range(nl1)), statel)), matches=int(nrl))

# --- SearchByBoW (src/ORBmatcher.cc:159-291, 525-658): the reference walks two DBoW2::FeatureVector maps (its own vendored class, compiled here);
# the oracle takes the shared nodes as CSR lists, which is what the walk with its lower_bound jumps visits
def shared_csr(nodeA, nodeB):
    ptrA, ptrB, idxA, idxB = [0], [0], [], []
    for nd in sorted(set(nodeA.tolist()) & set(nodeB.tolist())):
        idxA += np.nonzero(nodeA == nd)[0].tolist(); idxB += np.nonzero(nodeB == nd)[0].tolist()
        ptrA.append(len(idxA)); ptrB.append(len(idxB))
    return np.array(ptrA, np.int32), np.array(ptrB, np.int32), np.array(idxA, np.int32), np.array(idxB, np.int32)


for name in ("synth1234", "synth2000", "noise7"):
    img, (kp1, d1), (kp2, d2) = frames[name]
    n1, n2 = len(kp1), len(kp2)
    # nodes: nearby keypoints share a node (a coarse position hash plays the vocabulary), plus nodes that exist on one side only
    node1 = ((kp1["x"] // 80).astype(np.int32) * 16 + (kp1["y"] // 80).astype(np.int32)) * 3; node2 = ((kp2["x"] // 80).astype(np.int32) * 16 + (kp2["y"] // 80).astype(np.int32)) * 3
    node1 = np.where(rng.random(n1) < 0.1, node1 + 1, node1).astype(np.int32); node2 = np.where(rng.random(n2) < 0.1, node2 + 2, node2).astype(np.int32)
    valid1 = rng.choice([0, 1, 1, 1, 2], n1).astype(np.uint8); valid2 = rng.choice([0, 1, 1, 1, 2], n2).astype(np.uint8)
    pk, pf, ik, if_ = shared_csr(node1, node2)
    for ratio, ori in [(0.7, True), (0.9, False)]:
        out_r = np.zeros(n2, np.int32)
        nr = R.ref_search_by_bow(_p(kp1), _p(d1), n1, _p(node1), _p(valid1), _p(kp2), _p(d2), n2, _p(node2), C.c_float(ratio), int(ori), _p(out_r))
        a, no = orc.search_by_bow(kp1, d1, (valid1 == 1).astype(np.uint8), kp2, d2, pk, pf, ik, if_, ratio, ori)
        note("ORBmatcher::SearchByBoW(KF, F) %s r=%.1f ori=%d" % (name, ratio, ori), nr == no and np.array_equal(out_r, a), matches=int(nr), shared_nodes=len(pk) - 1)
        m_r = np.zeros(n1, np.int32)
        nr = R.ref_search_by_bow_keyframes(_p(kp1), _p(d1), n1, _p(node1), _p(valid1), _p(kp2), _p(d2), n2, _p(node2), _p(valid2), C.c_float(ratio), int(ori), _p(m_r))
        m_o, no = orc.search_by_bow_keyframes(kp1, d1, (valid1 == 1).astype(np.uint8), kp2, d2, (valid2 == 1).astype(np.uint8), pk, pf, ik, if_, ratio, ori)
        note("ORBmatcher::SearchByBoW(KF, KF) %s r=%.1f ori=%d" % (name, ratio, ori), nr == no and np.array_equal(m_r, m_o), matches=int(nr))

# --- LineSegment::ExtractLineSegment (src/ExtractLineSegment.cpp:18-69) over stand-in LSDDetector / BinaryDescriptor classes that forward to the oracle's
# LSD / LBD: the reference's own orchestration -- cap of 40, std::sort by response (UNSTABLE: decision D3 of the oracle is a stable sort), class ids,
# Eigen cross product for the line equations.  Lines are compared as records; an order that differs only among equal responses is D3's error bar.
from oracle_lib import KL_DTYPE
d3 = {"fixtures": 0, "lines": 0, "fixtures_with_another_order": 0, "lines_at_another_rank": 0, "fixtures_with_another_set": 0}
for name, img in [("synth1234", frames["synth1234"][0]), ("synth2000", frames["synth2000"][0]), ("big1235", frames["big1235"][0]), ("icl", np.load(os.path.join(HERE, "..", "..", "tests", "golden", "icl_input_gray.npz"))["gray"]),
                  ("noise3", noise_frame(3, w=320, h=240)), ("synth91", synth_frame(91, w=333, h=251)), ("synth2001", synth_frame(2001)), ("synth2002", synth_frame(2002)), ("few", synth_frame(5, nshapes=3, nstrokes=2))]:
    img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
    kl = np.zeros(64, KL_DTYPE); ld = np.zeros((64, 32), np.uint8); fn = np.zeros((64, 3), np.float64)
    n = R.ref_extract_line_segment(_p(img), w, h, _p(kl), _p(ld), _p(fn), 64)
    okl, old, ofn, _ = orc.lines_extract(img, 40)
    kl, ld, fn = kl[:n], ld[:n], fn[:n]
    same = n == len(okl) and np.array_equal(kl.view(np.uint8), okl.view(np.uint8)) and np.array_equal(ld, old) and np.array_equal(fn, ofn)
    # up to the order among equal responses: compare as sets of (record without class_id, descriptor, equation)
    key = lambda K, D, F: sorted((bytes(np.concatenate([K[i:i + 1].view(np.uint8)[:4], K[i:i + 1].view(np.uint8)[8:]])) + D[i].tobytes() + F[i].tobytes()) for i in range(len(K)))
    same_set = n == len(okl) and key(kl, ld, fn) == key(okl, old, ofn)
    moved = int(sum(1 for i in range(min(n, len(okl))) if kl[i:i + 1].view(np.uint8)[8:].tobytes() != okl[i:i + 1].view(np.uint8)[8:].tobytes()))
    d3["fixtures"] += 1; d3["lines"] += int(n); d3["fixtures_with_another_order"] += int(not same and same_set); d3["lines_at_another_rank"] += 0 if same else moved
    d3["fixtures_with_another_set"] += int(not same_set)
    note("LineSegment::ExtractLineSegment %s" % name, same_set, lines=int(n), identical_order=bool(same), lines_at_another_rank=0 if same else moved)
report["d3_error_bar"] = d3
print("D3 (std::sort vs stable sort of the 40 strongest lines):", d3)

# --- the keyframe-side line matchers (src/LSDmatcher.cpp:143-183, 286-362, 382-415 over KeyFrame::lineDescriptorMAD, src/KeyFrame.cc:820-845)
for name in ("synth1234", "synth2000", "big1235"):
    img = frames[name][0]
    l1 = orc.lines_extract(warp_prev(img), 200)[1]; l2 = orc.lines_extract(img, 200)[1]
    n1, n2 = len(l1), len(l2)
    has1 = (rng.random(n1) < 0.6).astype(np.uint8); has2 = (rng.random(n2) < 0.6).astype(np.uint8)
    pr, _, _ = orc.line_match(l1, l2, 0.5, True)          # ratio gate
    p05, _, _ = orc.line_match(l1, l2, 0.5, False)        # 0.5 x MAD gate
    p01, _, _ = orc.line_match(l1, l2, 0.1, False)        # 0.1 x MAD gate
    for which, label in [(0, "SearchByProjection(KF, F)"), (1, "SearchByDescriptor(KF, F)")]:
        out = np.full(n2, -9, np.int32); r = R.ref_line_keyframe_match(which, _p(l1), n1, _p(has1), _p(l2), n2, _p(has2), _p(out), n2)
        e = np.full(n2, -1, np.int32); cnt = 0
        for a, b in pr:
            if has1[a]: e[b] = a; cnt += 1
        note("LSDmatcher::%s %s" % (label, name), r == cnt and np.array_equal(out, e), matches=int(r))
    out = np.full(n1, -9, np.int32); r = R.ref_line_keyframe_match(2, _p(l1), n1, _p(has1), _p(l2), n2, _p(has2), _p(out), n1)
    e = np.full(n1, -1, np.int32); cnt = 0
    for a, b in p05:
        if has2[b]: e[a] = b; cnt += 1
    note("LSDmatcher::SearchByDescriptor(KF, KF) %s" % name, r == cnt and np.array_equal(out, e), matches=int(r))
    out = np.full(2 * n1 + 2, -9, np.int32); r = R.ref_line_keyframe_match(3, _p(l1), n1, _p(has1), _p(l2), n2, _p(has2), _p(out), 2 * n1 + 2)
    e = np.array([p for p in p01 if not (has1[p[0]] or has2[p[1]])], np.int32).reshape(-1, 2)
    note("LSDmatcher::SearchForTriangulation %s" % name, r == len(e) and np.array_equal(out[:2 * r].reshape(-1, 2), e), pairs=int(r))

# --- ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:660-826) + CheckDistEpipolarLine (:140-157): the epipole comes out of the reference's own
# pose algebra (C2 = R2w * Cw + t2w, float32 row order of the stand-in gemm); the oracle takes it as (ex, ey)
for name in ("synth1234", "synth2000", "noise7"):
    img, (kp1, d1), (kp2, d2) = frames[name]
    n1, n2 = len(kp1), len(kp2); h, w = img.shape
    node1 = ((kp1["x"] // 64).astype(np.int32) * 32 + (kp1["y"] // 64).astype(np.int32)); node2 = ((kp2["x"] // 64).astype(np.int32) * 32 + (kp2["y"] // 64).astype(np.int32))
    node1 = np.where(rng.random(n1) < 0.08, node1 + 4000, node1).astype(np.int32)
    a = np.deg2rad(1.5); cxh, cyh = (w - 1) / 2.0, (h - 1) / 2.0
    Hm = np.array([[np.cos(a), np.sin(a), cxh - 3.0 - cxh * np.cos(a) - cyh * np.sin(a)], [-np.sin(a), np.cos(a), cyh + 2.0 + cxh * np.sin(a) - cyh * np.cos(a)], [0, 0, 1.0]])
    for only_stereo, ori, C1 in [(False, True, (-4.0, 0.3, 0.9)), (True, True, (-4.0, 0.3, 0.9)), (False, False, (0.02, -0.01, 2.0))]:
        T2 = np.array([[0.99995, -0.0099998, 0, 0.012], [0.0099998, 0.99995, 0, -0.007], [0, 0, 1, 0.3], [0, 0, 0, 1]], np.float32)
        cam2 = np.array([520.9, 521.0, 325.1, 249.7], np.float32); C1a = np.array(C1, np.float32)
        c2 = [f32(f32(f32(f32(T2[r, 0] * C1a[0]) + f32(T2[r, 1] * C1a[1])) + f32(T2[r, 2] * C1a[2])) + T2[r, 3]) for r in range(3)]
        invz = f32(f32(1.0) / c2[2]); ex = f32(f32(f32(cam2[0] * c2[0]) * invz) + cam2[2]); ey = f32(f32(f32(cam2[1] * c2[1]) * invz) + cam2[3])
        E = np.array([[0, -1.0, float(ey)], [1.0, 0, -float(ex)], [-float(ey), float(ex), 0]])
        F12 = (E @ Hm).T; F12 = np.ascontiguousarray(F12 / np.abs(F12).max(), np.float32)      # (C order: the slice takes the raw buffer)
        sg8 = (scale8 * scale8).astype(np.float32)
        free1 = (rng.random(n1) < 0.9).astype(np.uint8); free2 = (rng.random(n2) < 0.9).astype(np.uint8)
        pst = 0.7 if only_stereo else 0.3
        ur1 = np.where(rng.random(n1) < pst, kp1["x"] - 5, -1).astype(np.float32); ur2 = np.where(rng.random(n2) < pst, kp2["x"] - 5, -1).astype(np.float32)
        m_r = np.zeros(n1, np.int32)
        nr = R.ref_search_for_triangulation(_p(kp1), _p(d1), n1, _p(node1), _p(free1), _p(ur1), _p(kp2), _p(d2), n2, _p(node2), _p(free2), _p(ur2), _p(F12), _p(T2), _p(C1a), _p(cam2),
                                            _p(scale8), _p(sg8), int(only_stereo), int(ori), _p(m_r))
        pk, pf, ik, if_ = shared_csr(node1, node2)
        m_o, no = orc.search_for_triangulation(kp1, d1, ur1, free1, kp2, d2, ur2, free2, pk, pf, ik, if_, F12, float(ex), float(ey), scale8, sg8, only_stereo, ori)
        note("ORBmatcher::SearchForTriangulation %s stereo=%d ori=%d" % (name, only_stereo, ori), nr == no and np.array_equal(m_r, m_o), pairs=int(nr), epipole=(round(float(ex), 1), round(float(ey), 1)))

# --- ORBmatcher::Fuse(KeyFrame*, vector<MapPoint*>, th) (src/ORBmatcher.cc:828-978) over KeyFrame::GetFeaturesInArea / IsInImage (src/KeyFrame.cc:610-649,
# 686-689) and MapPoint::PredictScale / Get*DistanceInvariance (src/MapPoint.cc:378-405): LocalMapping::SearchInNeighbors' call (th = 3).  The oracle restates the
# search per map point (fuse_search: the window, the level gate, the chi-square gates, the best descriptor); the windows are the ones the reference's own
# projection block forms (ref_fuse_queries, same stand-ins and leaves).  Compared: which keyframe feature every map point is fused to, and the count.
FMP = np.dtype([("wp", "<f4", 3), ("nrm", "<f4", 3), ("minDist", "<f4"), ("maxDist", "<f4"), ("nObs", "<i4"), ("bad", "<i4"), ("inKF", "<i4")])
FQ = np.dtype([("u", "<f4"), ("v", "<f4"), ("ur", "<f4"), ("radius", "<f4"), ("level", "<i4"), ("valid", "<i4")])
inv_sigma2 = np.array([f32(1.0) / (s_ * s_) for s_ in scale8], np.float32); log_sf = float(np.log(f32(1.2)))
for name in ("synth1234", "synth2000", "big1235"):
    img, (kp1, d1), (kp2, d2) = frames[name]
    h, w = img.shape; bb = np.array((0.0, float(w), 0.0, float(h)), np.float32); n2 = len(kp2)
    fx = fy = f32(0.9 * w); cxx, cyy = f32(w / 2 - 3.25), f32(h / 2 + 1.5)
    for stereo, th, same in ((False, 3.0, False), (True, 3.0, False), (True, 5.0, False), (True, 3.0, True), (False, 4.0, True)):
        skp, sd = (kp2, d2) if same else (kp1, d1)      # where the map points come from: this keyframe's own features (most of them fuse) or the other view's
        ur2 = np.where(rng.random(n2) < 0.5, kp2["x"] - rng.uniform(2, 40, n2), -1).astype(np.float32) if stereo else np.full(n2, -1, np.float32)
        cam = np.array([fx, fy, cxx, cyy, 40.0], np.float32)
        # a pose (small rotation about y and x, a translation) and its camera centre Ow = -R^T t, all float32
        ay, ax = 0.07, -0.04
        Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]]); Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
        Rm = (Ry @ Rx).astype(np.float32); t = np.array([0.3, -0.2, 0.5], np.float32)
        Tcw = np.eye(4, dtype=np.float32); Tcw[:3, :3] = Rm; Tcw[:3, 3] = t
        Ow = (-(Rm.astype(np.float64).T @ t.astype(np.float64))).astype(np.float32)
        # map points: features of the other view unprojected at random depths (so that they land near features of this keyframe), some pushed
        # out of the image / behind the camera / out of their distance range / seen from the side
        nmp = len(skp); depth = rng.uniform(2.0, 12.0, nmp)
        uu = skp["x"].astype(np.float64) + rng.uniform(-4, 4, nmp); vv = skp["y"].astype(np.float64) + rng.uniform(-4, 4, nmp)
        pc = np.stack([(uu - float(cxx)) / float(fx) * depth, (vv - float(cyy)) / float(fy) * depth, depth], 1)
        kind = rng.choice(6, nmp, p=[0.7, 0.06, 0.06, 0.06, 0.06, 0.06])
        pc[kind == 1, 2] *= -1                                        # behind the camera
        pc[kind == 2, 0] += 3 * depth[kind == 2]                      # outside the image
        pw = (Rm.astype(np.float64).T @ (pc - t.astype(np.float64)).T).T
        mp = np.zeros(nmp, FMP); mp["wp"] = pw.astype(np.float32)
        po = pw - Ow.astype(np.float64); dist = np.linalg.norm(po, axis=1)
        nrm = po / dist[:, None]; nrm[kind == 3] = -nrm[kind == 3]    # seen from behind
        side = kind == 4; nrm[side] = np.cross(nrm[side], [0.0, 1.0, 0.0]) * 0.9 + nrm[side] * 0.45      # a viewing angle around the 60-degree gate
        mp["nrm"] = nrm.astype(np.float32)
        lvl = skp["octave"].astype(np.int64) if same else rng.integers(0, 8, nmp); mp["maxDist"] = (dist * 1.2 ** lvl * rng.uniform(0.9, 1.3, nmp)).astype(np.float32)
        mp["minDist"] = (mp["maxDist"] / f32(1.2 ** 7)).astype(np.float32)
        far = kind == 5; mp["maxDist"][far] = (dist[far] * 0.5).astype(np.float32)          # outside the scale range
        mp["nObs"] = rng.integers(0, 4, nmp); mp["bad"] = rng.random(nmp) < 0.04; mp["inKF"] = rng.random(nmp) < 0.04
        mpd = sd.copy()
        flip = rng.random(nmp) < 0.5
        for i in np.nonzero(flip)[0]: mpd[i, rng.integers(0, 32, 6)] ^= rng.integers(1, 256, 6).astype(np.uint8)      # some descriptors drift past TH_LOW
        state = rng.choice([0, 0, 1, 1, 2], n2).astype(np.uint8); sobs = rng.integers(0, 4, n2).astype(np.int32)
        fused = np.zeros(nmp, np.int32); act = np.zeros(nmp, np.int32)
        nr = R.ref_fuse(_p(kp2), _p(d2), n2, _p(bb), _p(scale8), _p(inv_sigma2), C.c_float(log_sf), _p(ur2), _p(state), _p(sobs), _p(cam), _p(Tcw), _p(Ow),
                        _p(mp), _p(mpd), nmp, C.c_float(th), _p(fused), _p(act))
        fq = np.zeros(nmp, FQ)
        R.ref_fuse_queries(_p(bb), _p(scale8), C.c_float(log_sf), _p(cam), _p(Tcw), _p(Ow), _p(mp), nmp, C.c_float(th), _p(fq))
        q = np.zeros(nmp, fe.PQ_DTYPE)
        q["u"] = fq["u"]; q["v"] = fq["v"]; q["ur"] = fq["ur"]; q["radius"] = fq["radius"]; q["min_level"] = fq["level"] - 1; q["max_level"] = fq["level"]; q["valid"] = fq["valid"]
        bi, bd = orc.fuse_search(0, 1, kp2, d2, q, mpd, uright=ur2, inv_level_sigma2=inv_sigma2, bounds=tuple(bb))
        want = np.where((fq["valid"] == 1) & (bi >= 0) & (bd <= 50), bi, -1)
        eq = nr == int((want >= 0).sum()) and np.array_equal(fused, want)
        note("Fuse(KeyFrame, MapPoints) %s %s th=%g %s" % (name, "stereo" if stereo else "mono", th, "own features" if same else "other view"), eq, map_points=int(nmp), projected=int(fq["valid"].sum()), fused=int(nr),
             added=int((act == 1).sum()), kf_point_kept=int((act == 2).sum()), kf_point_replaced=int((act == 3).sum()))

# --- ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:1475-1602) over MapPoint::PredictScale(dist, Frame*)
# (src/MapPoint.cc:407-422): Tracking::Relocalization's calls (th = 10 / ORBdist = 100, then 3 / 64).  The oracle's restatement takes the windows; they are the ones
# the reference's own projection block forms (ref_reloc_queries).  Compared: CurrentFrame.mvpMapPoints after the call and the count.
for name in ("synth1234", "synth2000", "big1235"):
    img, (kp1, d1), (kp2, d2) = frames[name]
    h, w = img.shape; bb = np.array((0.0, float(w), 0.0, float(h)), np.float32); n2 = len(kp2); nkf = len(kp1)
    fx = fy = f32(0.9 * w); cxx, cyy = f32(w / 2 - 3.25), f32(h / 2 + 1.5); cam = np.array([fx, fy, cxx, cyy, 40.0], np.float32)
    ay, ax = -0.05, 0.03
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]]); Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Rm = (Ry @ Rx).astype(np.float32); t = np.array([-0.2, 0.1, 0.4], np.float32)
    Tcw = np.eye(4, dtype=np.float32); Tcw[:3, :3] = Rm; Tcw[:3, 3] = t
    Ow = (-(Rm.astype(np.float64).T @ t.astype(np.float64)))
    for th, orbdist, ori, same in ((10.0, 100, 1, False), (3.0, 64, 1, False), (10.0, 100, 0, False), (10.0, 100, 1, True), (3.0, 64, 1, True)):
        # the keyframe: the other view's features, or (same) the current frame's own -- then most map points find their feature again
        kkp, kd = (kp2, d2) if same else (kp1, d1); nkf = len(kkp)
        depth = rng.uniform(2.0, 12.0, nkf)
        uu = kkp["x"].astype(np.float64) + rng.uniform(-6, 6, nkf); vv = kkp["y"].astype(np.float64) + rng.uniform(-6, 6, nkf)
        pc = np.stack([(uu - float(cxx)) / float(fx) * depth, (vv - float(cyy)) / float(fy) * depth, depth], 1)
        kind = rng.choice(4, nkf, p=[0.85, 0.05, 0.05, 0.05])
        pc[kind == 1, 0] += 3 * depth[kind == 1]                      # outside the image
        pw = (Rm.astype(np.float64).T @ (pc - t.astype(np.float64)).T).T
        mp = np.zeros(nkf, FMP); mp["wp"] = pw.astype(np.float32)
        dist = np.linalg.norm(pw - Ow, axis=1)
        lvl = kkp["octave"].astype(np.int64); mp["maxDist"] = (dist * 1.2 ** lvl * rng.uniform(0.9, 1.3, nkf)).astype(np.float32)
        mp["minDist"] = (mp["maxDist"] / f32(1.2 ** 7)).astype(np.float32)
        far = kind == 2; mp["maxDist"][far] = (dist[far] * 0.5).astype(np.float32)
        mp["bad"] = kind == 3
        present = rng.choice([0, 1, 1, 1, 1, 2], nkf).astype(np.uint8)
        mpd = kd.copy()
        for i in np.nonzero(rng.random(nkf) < 0.5)[0]: mpd[i, rng.integers(0, 32, 5)] ^= rng.integers(1, 256, 5).astype(np.uint8)
        state = (rng.random(n2) < 0.3).astype(np.uint8)
        out_r = np.zeros(n2, np.int32)
        nr = R.ref_search_by_projection_reloc(_p(kp2), _p(d2), n2, _p(bb), _p(scale8), C.c_float(log_sf), _p(state), _p(cam), _p(Tcw), _p(kkp), _p(present), _p(mp), _p(mpd), nkf,
                                              C.c_float(th), orbdist, ori, _p(out_r))
        fq = np.zeros(nkf, FQ)
        R.ref_reloc_queries(_p(bb), _p(scale8), C.c_float(log_sf), _p(cam), _p(Tcw), _p(present), _p(mp), nkf, C.c_float(th), _p(fq))
        q = np.zeros(nkf, fe.PQ_DTYPE)
        q["u"] = fq["u"]; q["v"] = fq["v"]; q["radius"] = fq["radius"]; q["max_level"] = fq["level"]; q["min_level"] = fq["level"] - 1; q["angle"] = kkp["angle"]; q["valid"] = fq["valid"]
        a, no = orc.search_by_projection_reloc(kp2, d2, q, mpd, state, orb_dist=orbdist, check_orientation=bool(ori), bounds=tuple(bb))
        want = np.array([x if x >= 0 else (-2 if state[i] else -1) for i, x in enumerate(a)], np.int32)      # (pruned by the rotation histogram: NULL again, i.e. -1)
        eq = nr == no and np.array_equal(out_r, want)
        note("SearchByProjection(Cur, KF, found) %s th=%g dist=%d ori=%d %s" % (name, th, orbdist, ori, "own features" if same else "other view"), eq,
             keyframe_points=int((present == 1).sum()), projected=int(fq["valid"].sum()), matches=int(nr), pruned=int((a == -2).sum()))

# --- ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th) (src/ORBmatcher.cc:293-406): LoopClosing::ComputeSim3's call (th = 10).  Scw = s [R | t]:
# the decomposition (scale from the first row, R = sR / s, t / s) is the reference's own, on the stand-in Mat algebra; the oracle takes the windows it leads to.
for name in ("synth1234", "synth2000", "big1235"):
    img, (kp1, d1), (kp2, d2) = frames[name]
    h, w = img.shape; bb = np.array((0.0, float(w), 0.0, float(h)), np.float32); n2 = len(kp2)
    fx = fy = f32(0.9 * w); cxx, cyy = f32(w / 2 - 3.25), f32(h / 2 + 1.5); cam = np.array([fx, fy, cxx, cyy, 40.0], np.float32)
    ay, ax = 0.04, 0.06
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]]); Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Rm64 = Ry @ Rx; t64 = np.array([0.25, 0.15, -0.3])
    for scale, th, same in ((1.0, 10, False), (1.37, 10, False), (0.81, 10, True), (1.37, 4, True)):
        Scw = np.eye(4, dtype=np.float32); Scw[:3, :3] = (scale * Rm64).astype(np.float32); Scw[:3, 3] = (scale * t64).astype(np.float32)
        Ow = -(Rm64.T @ t64)
        skp, sd = (kp2, d2) if same else (kp1, d1); nmp = len(skp)
        depth = rng.uniform(2.0, 12.0, nmp)
        uu = skp["x"].astype(np.float64) + rng.uniform(-6, 6, nmp); vv = skp["y"].astype(np.float64) + rng.uniform(-6, 6, nmp)
        pc = np.stack([(uu - float(cxx)) / float(fx) * depth, (vv - float(cyy)) / float(fy) * depth, depth], 1)
        kind = rng.choice(5, nmp, p=[0.8, 0.05, 0.05, 0.05, 0.05])
        pc[kind == 1, 2] *= -1; pc[kind == 2, 0] += 3 * depth[kind == 2]
        pw = (Rm64.T @ (pc - t64).T).T
        mp = np.zeros(nmp, FMP); mp["wp"] = pw.astype(np.float32)
        po = pw - Ow; dist = np.linalg.norm(po, axis=1); nrm = po / dist[:, None]; nrm[kind == 3] = -nrm[kind == 3]; mp["nrm"] = nrm.astype(np.float32)
        lvl = skp["octave"].astype(np.int64) if same else rng.integers(0, 8, nmp)
        mp["maxDist"] = (dist * 1.2 ** lvl * rng.uniform(0.9, 1.3, nmp)).astype(np.float32); mp["minDist"] = (mp["maxDist"] / f32(1.2 ** 7)).astype(np.float32)
        mp["bad"] = kind == 4
        mpd = sd.copy()
        for i in np.nonzero(rng.random(nmp) < 0.5)[0]: mpd[i, rng.integers(0, 32, 5)] ^= rng.integers(1, 256, 5).astype(np.uint8)
        # vpMatched on entry: empty slots, slots holding some other point, slots holding one of the candidates (those candidates are "already found")
        m_in = np.full(n2, -1, np.int32); r_ = rng.random(n2); m_in[r_ < 0.15] = -2
        pick = np.nonzero((r_ >= 0.15) & (r_ < 0.25))[0]; m_in[pick] = rng.permutation(nmp)[:len(pick)]
        found = np.zeros(nmp, np.uint8); found[m_in[m_in >= 0]] = 1
        m_ref = m_in.copy()
        nr = R.ref_search_by_projection_sim3(_p(kp2), _p(d2), n2, _p(bb), _p(scale8), C.c_float(log_sf), _p(cam), _p(Scw), _p(mp), _p(mpd), nmp, th, _p(m_ref))
        fq = np.zeros(nmp, FQ)
        R.ref_sim3_queries(_p(bb), _p(scale8), C.c_float(log_sf), _p(cam), _p(Scw), _p(found), _p(mp), nmp, C.c_float(th), 0, _p(fq))
        q = np.zeros(nmp, fe.PQ_DTYPE)
        q["u"] = fq["u"]; q["v"] = fq["v"]; q["radius"] = fq["radius"]; q["max_level"] = fq["level"]; q["min_level"] = fq["level"] - 1; q["valid"] = fq["valid"]
        a, no = orc.search_by_projection_sim3(0, kp2, d2, q, mpd, (m_in != -1).astype(np.uint8), bounds=tuple(bb))
        want = np.where(a >= 0, a, m_in).astype(np.int32)
        eq = nr == no and np.array_equal(m_ref, want)
        note("SearchByProjection(KF, Scw, points) %s s=%g th=%d %s" % (name, scale, th, "own features" if same else "other view"), eq,
             candidates=int(nmp), already_found=int(found.sum()), projected=int(fq["valid"].sum()), matches=int(nr))

# --- ORBmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, vpPoints, th, vpReplacePoint) (src/ORBmatcher.cc:980-1103): LoopClosing::SearchAndFuse's call (th = 4).  The oracle's
# fuse_search without the chi-square gates, on the windows of the reference's own Sim3 projection block; candidates the keyframe already observes are skipped.
for name in ("synth1234", "synth2000", "big1235"):
    img, (kp1, d1), (kp2, d2) = frames[name]
    h, w = img.shape; bb = np.array((0.0, float(w), 0.0, float(h)), np.float32); n2 = len(kp2)
    fx = fy = f32(0.9 * w); cxx, cyy = f32(w / 2 - 3.25), f32(h / 2 + 1.5); cam = np.array([fx, fy, cxx, cyy, 40.0], np.float32)
    ay, ax = -0.03, 0.05
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]]); Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Rm64 = Ry @ Rx; t64 = np.array([-0.15, 0.2, 0.35])
    for scale, th, same in ((1.0, 4.0, False), (1.21, 4.0, True), (0.9, 6.0, True)):
        Scw = np.eye(4, dtype=np.float32); Scw[:3, :3] = (scale * Rm64).astype(np.float32); Scw[:3, 3] = (scale * t64).astype(np.float32)
        Ow = -(Rm64.T @ t64)
        skp, sd = (kp2, d2) if same else (kp1, d1); nmp = len(skp)
        depth = rng.uniform(2.0, 12.0, nmp)
        uu = skp["x"].astype(np.float64) + rng.uniform(-3, 3, nmp); vv = skp["y"].astype(np.float64) + rng.uniform(-3, 3, nmp)
        pc = np.stack([(uu - float(cxx)) / float(fx) * depth, (vv - float(cyy)) / float(fy) * depth, depth], 1)
        kind = rng.choice(5, nmp, p=[0.8, 0.05, 0.05, 0.05, 0.05])
        pc[kind == 1, 2] *= -1; pc[kind == 2, 0] += 3 * depth[kind == 2]
        pw = (Rm64.T @ (pc - t64).T).T
        mp = np.zeros(nmp, FMP); mp["wp"] = pw.astype(np.float32)
        po = pw - Ow; dist = np.linalg.norm(po, axis=1); nrm = po / dist[:, None]; nrm[kind == 3] = -nrm[kind == 3]; mp["nrm"] = nrm.astype(np.float32)
        lvl = skp["octave"].astype(np.int64) if same else rng.integers(0, 8, nmp)
        mp["maxDist"] = (dist * 1.2 ** lvl * rng.uniform(0.9, 1.3, nmp)).astype(np.float32); mp["minDist"] = (mp["maxDist"] / f32(1.2 ** 7)).astype(np.float32)
        mp["bad"] = kind == 4
        mpd = sd.copy()
        for i in np.nonzero(rng.random(nmp) < 0.5)[0]: mpd[i, rng.integers(0, 32, 5)] ^= rng.integers(1, 256, 5).astype(np.uint8)
        state = rng.choice([0, 0, 1, 1, 2], n2).astype(np.uint8)
        slot = np.full(nmp, -1, np.int32); own = rng.permutation(nmp)[:nmp // 12]; slot[own] = rng.permutation(n2)[:len(own)]      # candidates the keyframe already holds
        found = ((slot >= 0) & (mp["bad"] == 0)).astype(np.uint8)                      # GetMapPoints(): good points only
        fused = np.zeros(nmp, np.int32); act = np.zeros(nmp, np.int32)
        nr = R.ref_fuse_sim3(_p(kp2), _p(d2), n2, _p(bb), _p(scale8), C.c_float(log_sf), _p(state), _p(slot), _p(cam), _p(Scw), _p(mp), _p(mpd), nmp, C.c_float(th), _p(fused), _p(act))
        fq = np.zeros(nmp, FQ)
        R.ref_sim3_queries(_p(bb), _p(scale8), C.c_float(log_sf), _p(cam), _p(Scw), _p(found), _p(mp), nmp, C.c_float(th), 1, _p(fq))
        q = np.zeros(nmp, fe.PQ_DTYPE)
        q["u"] = fq["u"]; q["v"] = fq["v"]; q["radius"] = fq["radius"]; q["min_level"] = fq["level"] - 1; q["max_level"] = fq["level"]; q["valid"] = fq["valid"]
        bi, bd = orc.fuse_search(0, 0, kp2, d2, q, mpd, bounds=tuple(bb))
        want = np.where((fq["valid"] == 1) & (bi >= 0) & (bd <= 50), bi, -1)
        eq = nr == int((want >= 0).sum()) and np.array_equal(fused, want)
        note("Fuse(KeyFrame, Scw, points) %s s=%g th=%g %s" % (name, scale, th, "own features" if same else "other view"), eq, candidates=int(nmp), already_in_keyframe=int(found.sum()),
             projected=int(fq["valid"].sum()), fused=int(nr), added=int((act == 1).sum()), to_replace=int((act == 2).sum()))

# --- ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1105-1329): LoopClosing::ComputeSim3's call (th = 7.5).  Both directions are the fuse_search of the oracle without the
# chi-square gates and with TH_HIGH, on the windows of the reference's own two projection blocks; the agreement test is restated here in three lines.
for name in ("synth1234", "synth2000", "big1235"):
    img, (kp1, d1), (kp2, d2) = frames[name]
    h, w = img.shape; bb = np.array((0.0, float(w), 0.0, float(h)), np.float32); n = len(kp2)
    fx = fy = f32(0.9 * w); cxx, cyy = f32(w / 2 - 3.25), f32(h / 2 + 1.5); cam = np.array([fx, fy, cxx, cyy, 40.0], np.float32)
    ay, ax = 0.05, -0.02
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]]); Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Rw = Ry @ Rx; tw = np.array([0.1, -0.25, 0.3])
    T1w = np.eye(4, dtype=np.float32); T1w[:3, :3] = Rw.astype(np.float32); T1w[:3, 3] = tw.astype(np.float32); T2w = T1w.copy()      # both keyframes at one pose, the Sim3 slightly off identity
    for s12, ang, th in ((1.0, 0.0, 7.5), (1.03, 0.004, 7.5), (0.97, -0.006, 4.0)):
        Rz = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]]); R12 = Rz.astype(np.float32); t12 = np.array([0.02, -0.01, 0.015 * (s12 != 1.0)], np.float32)

        def points(kp):
            m = len(kp); depth = rng.uniform(2.0, 12.0, m)
            uu = kp["x"].astype(np.float64) + rng.uniform(-2, 2, m); vv = kp["y"].astype(np.float64) + rng.uniform(-2, 2, m)
            pc = np.stack([(uu - float(cxx)) / float(fx) * depth, (vv - float(cyy)) / float(fy) * depth, depth], 1)
            pw = (Rw.T @ (pc - tw).T).T
            mp = np.zeros(m, FMP); mp["wp"] = pw.astype(np.float32)
            lvl = kp["octave"].astype(np.int64)
            mp["maxDist"] = (depth * np.sqrt(1 + ((uu - float(cxx)) / float(fx)) ** 2 + ((vv - float(cyy)) / float(fy)) ** 2) * 1.2 ** lvl * rng.uniform(0.9, 1.3, m)).astype(np.float32)
            mp["minDist"] = (mp["maxDist"] / f32(1.2 ** 7)).astype(np.float32); mp["bad"] = rng.random(m) < 0.04
            return mp
        mp1, mp2 = points(kp2), points(kp2)
        mpd1, mpd2 = d2.copy(), d2.copy()
        for dd in (mpd1, mpd2):
            for i in np.nonzero(rng.random(n) < 0.5)[0]: dd[i, rng.integers(0, 32, 6)] ^= rng.integers(1, 256, 6).astype(np.uint8)
        present1 = (rng.random(n) < 0.85).astype(np.uint8); present2 = (rng.random(n) < 0.85).astype(np.uint8)
        m_in = np.full(n, -1, np.int32); r_ = rng.random(n); m_in[r_ < 0.06] = -2
        pick = np.nonzero((r_ >= 0.06) & (r_ < 0.16))[0]; m_in[pick] = rng.choice(np.nonzero(present2)[0], len(pick), replace=False)
        m_ref = m_in.copy()
        nr = R.ref_search_by_sim3(_p(kp2), _p(d2), n, _p(kp2), _p(d2), n, _p(bb), _p(scale8), C.c_float(log_sf), _p(cam), _p(T1w), _p(T2w), C.c_float(s12), _p(R12), _p(t12),
                                  _p(present1), _p(mp1), _p(mpd1), _p(present2), _p(mp2), _p(mpd2), C.c_float(th), _p(m_ref))
        already1 = m_in != -1; already2 = np.zeros(n, bool); already2[m_in[m_in >= 0]] = True
        vn = []
        for direction, present, already, mp, mpd in ((0, present1, already1, mp1, mpd1), (1, present2, already2, mp2, mpd2)):
            skip = ((present == 0) | already).astype(np.uint8); fq = np.zeros(n, FQ)
            R.ref_sim3_pair_queries(direction, _p(bb), _p(scale8), C.c_float(log_sf), _p(cam), _p(T1w), _p(T2w), C.c_float(s12), _p(R12), _p(t12), _p(skip), _p(mp), n, C.c_float(th), _p(fq))
            q = np.zeros(n, fe.PQ_DTYPE)
            q["u"] = fq["u"]; q["v"] = fq["v"]; q["radius"] = fq["radius"]; q["min_level"] = fq["level"] - 1; q["max_level"] = fq["level"]; q["valid"] = fq["valid"]
            bi, bd = orc.fuse_search(0, 0, kp2, d2, q, mpd, bounds=tuple(bb))
            vn.append(np.where((fq["valid"] == 1) & (bi >= 0) & (bd <= 100), bi, -1))
        want = m_in.copy(); found = 0
        for i1 in range(n):
            i2 = vn[0][i1]
            if i2 >= 0 and vn[1][i2] == i1: want[i1] = i2 if present2[i2] else -1; found += 1
        eq = nr == found and np.array_equal(m_ref, want)
        note("SearchBySim3 %s s12=%g rot=%g th=%g" % (name, s12, ang, th), eq, points1=int(present1.sum()), points2=int(present2.sum()), already_matched=int(already1.sum()),
             one_way=int((vn[0] >= 0).sum()), other_way=int((vn[1] >= 0).sum()), agreed=int(found))

# --- LSDmatcher::Fuse(KeyFrame*, vector<MapLine*>, th) (src/LSDmatcher.cpp:417-548) over KeyFrame::GetLinesInArea (src/KeyFrame.cc:651-684) and MapLine::PredictScale /
# Get*DistanceInvariance (src/MapLine.cpp:374-395): LocalMapping::SearchInNeighbors' line call.  Oracle: fuse_search, kind 1, on the windows of the reference's projection block.
# (MapLine::PredictScale does not clamp its level, and the reference indexes mvScaleFactors with it: the cases keep every predicted level inside [0, 7].)
FML = np.dtype([("wp", "<f8", 6), ("nrm", "<f8", 3), ("minDist", "<f4"), ("maxDist", "<f4"), ("nObs", "<i4"), ("bad", "<i4")])
FQL = np.dtype([("u1", "<f4"), ("v1", "<f4"), ("u2", "<f4"), ("v2", "<f4"), ("radius", "<f4"), ("level", "<i4"), ("valid", "<i4")])
for name in ("synth1234", "synth2000", "big1235"):
    img = frames[name][0]; h, w = img.shape; bb = np.array((0.0, float(w), 0.0, float(h)), np.float32)
    kl1, ld1 = orc.lines_extract(warp_prev(img), 200)[:2]; kl2, ld2 = orc.lines_extract(img, 200)[:2]
    fx = fy = f32(0.9 * w); cxx, cyy = f32(w / 2 - 3.25), f32(h / 2 + 1.5); cam = np.array([fx, fy, cxx, cyy, 40.0], np.float32)
    ay, ax = 0.03, -0.05
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]]); Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Rm = (Ry @ Rx).astype(np.float32); t = np.array([0.2, 0.1, 0.3], np.float32)
    Tcw = np.eye(4, dtype=np.float32); Tcw[:3, :3] = Rm; Tcw[:3, 3] = t
    Ow32 = (-(Rm.astype(np.float64).T @ t.astype(np.float64))).astype(np.float32)
    R64 = Rm.astype(np.float64); t64 = t.astype(np.float64)
    for th, same in ((3.0, False), (3.0, True), (6.0, True)):
        skl, sld = (kl2, ld2) if same else (kl1, ld1); nml = len(skl); nl2 = len(kl2)
        z = rng.uniform(2.0, 9.0, nml); dz = rng.uniform(-0.4, 0.4, nml)
        def unproj(x, y, zz):
            pc = np.stack([(x.astype(np.float64) + rng.uniform(-2, 2, nml) - float(cxx)) / float(fx) * zz, (y.astype(np.float64) + rng.uniform(-2, 2, nml) - float(cyy)) / float(fy) * zz, zz], 1)
            return (R64.T @ (pc - t64).T).T
        kind = rng.choice(5, nml, p=[0.8, 0.05, 0.05, 0.05, 0.05])
        zs = z.copy(); zs[kind == 1] *= -1                                     # one end behind the camera
        sp = unproj(skl["startPointX"], skl["startPointY"], zs); ep = unproj(skl["endPointX"], skl["endPointY"], z + dz)
        sp[kind == 2, 0] += 40.0                                               # projects outside the image
        ml = np.zeros(nml, FML); ml["wp"] = np.concatenate([sp, ep], 1).astype(np.float32).astype(np.float64)      # (Vector6d of values a float holds exactly)
        mid = 0.5 * (sp + ep); om = mid - Ow32.astype(np.float64); dist = np.linalg.norm(om, axis=1); nrm = om / dist[:, None]; nrm[kind == 3] = -nrm[kind == 3]
        ml["nrm"] = nrm.astype(np.float32).astype(np.float64)
        lvl = rng.choice([0, 0, 0, 1, 1, 2, 4], nml); ml["maxDist"] = (dist * 1.2 ** lvl * rng.uniform(0.86, 0.99, nml)).astype(np.float32)      # ceil(log(max / dist) / log 1.2) = lvl
        ml["minDist"] = (ml["maxDist"] / f32(1.2 ** 7)).astype(np.float32); ml["bad"] = kind == 4; ml["nObs"] = rng.integers(0, 4, nml)
        mld = sld.copy()
        for i in np.nonzero(rng.random(nml) < 0.4)[0]: mld[i, rng.integers(0, 32, 4)] ^= rng.integers(1, 256, 4).astype(np.uint8)
        state = rng.choice([0, 0, 1, 1, 2], nl2).astype(np.uint8); sobs = rng.integers(0, 4, nl2).astype(np.int32)
        fql = np.zeros(nml, FQL)
        R.ref_line_fuse_queries(_p(bb), _p(scale8), C.c_float(log_sf), _p(cam), _p(Tcw), _p(Ow32), _p(ml), nml, C.c_float(th), _p(fql))
        assert (fql["valid"] != 2).all(), "a predicted level outside [0, 7]: the reference would read out of bounds"
        fused = np.zeros(nml, np.int32); act = np.zeros(nml, np.int32)
        nr = R.ref_line_fuse(_p(kl2), _p(ld2), nl2, _p(bb), _p(scale8), C.c_float(log_sf), _p(state), _p(sobs), _p(cam), _p(Tcw), _p(Ow32), _p(ml), _p(mld), nml, C.c_float(th), _p(fused), _p(act))
        q = np.zeros(nml, fe.PQ_DTYPE)
        q["u"] = fql["u1"]; q["v"] = fql["v1"]; q["u2"] = fql["u2"]; q["v2"] = fql["v2"]; q["radius"] = fql["radius"]; q["min_level"] = fql["level"] - 1; q["max_level"] = fql["level"]; q["valid"] = fql["valid"]
        bi, bd = orc.fuse_search(1, 0, kl2, ld2, q, mld, bounds=tuple(bb))
        want = np.where((fql["valid"] == 1) & (bi >= 0) & (bd <= 50), bi, -1)
        eq = nr == int((want >= 0).sum()) and np.array_equal(fused, want)
        note("LSDmatcher::Fuse(KeyFrame, MapLines) %s th=%g %s" % (name, th, "own lines" if same else "other view"), eq, map_lines=int(nml), projected=int((fql["valid"] == 1).sum()),
             fused=int(nr), added=int((act == 1).sum()), kf_line_kept=int((act == 2).sum()), kf_line_replaced=int((act == 3).sum()))

# --- LSDmatcher::SearchByProjection(KeyFrame*, Scw, vpLines, vpMatched, th) (src/LSDmatcher.cpp:558-683) and LSDmatcher::Fuse(KeyFrame*, Scw, vpLines, th, vpReplaceLine)
# (:931-1063): the loop-closing line calls.  Oracle: search_by_projection_sim3 / fuse_search, kind 1, on the windows of the reference's (common) Sim3 projection block.
for name in ("synth1234", "synth2000", "big1235"):
    img = frames[name][0]; h, w = img.shape; bb = np.array((0.0, float(w), 0.0, float(h)), np.float32)
    kl1, ld1 = orc.lines_extract(warp_prev(img), 200)[:2]; kl2, ld2 = orc.lines_extract(img, 200)[:2]; nl2 = len(kl2)
    fx = fy = f32(0.9 * w); cxx, cyy = f32(w / 2 - 3.25), f32(h / 2 + 1.5); cam = np.array([fx, fy, cxx, cyy, 40.0], np.float32)
    ay, ax = -0.04, 0.02
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]]); Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    R64 = Ry @ Rx; t64 = np.array([0.1, -0.15, 0.25]); Ow64 = -(R64.T @ t64)
    for scale, th, same in ((1.0, 10.0, False), (1.19, 10.0, True), (0.87, 4.0, True)):
        Scw = np.eye(4, dtype=np.float32); Scw[:3, :3] = (scale * R64).astype(np.float32); Scw[:3, 3] = (scale * t64).astype(np.float32)
        skl, sld = (kl2, ld2) if same else (kl1, ld1); nml = len(skl)
        z = rng.uniform(2.0, 9.0, nml); dz = rng.uniform(-0.4, 0.4, nml)
        def unproj(x, y, zz):
            pc = np.stack([(x.astype(np.float64) + rng.uniform(-2, 2, nml) - float(cxx)) / float(fx) * zz, (y.astype(np.float64) + rng.uniform(-2, 2, nml) - float(cyy)) / float(fy) * zz, zz], 1)
            return (R64.T @ (pc - t64).T).T
        kind = rng.choice(5, nml, p=[0.8, 0.05, 0.05, 0.05, 0.05])
        zs = z.copy(); zs[kind == 1] *= -1
        sp = unproj(skl["startPointX"], skl["startPointY"], zs); ep = unproj(skl["endPointX"], skl["endPointY"], z + dz); sp[kind == 2, 0] += 40.0
        ml = np.zeros(nml, FML); ml["wp"] = np.concatenate([sp, ep], 1).astype(np.float32).astype(np.float64)
        om = 0.5 * (sp + ep) - Ow64; dist = np.linalg.norm(om, axis=1); nrm = om / dist[:, None]; nrm[kind == 3] = -nrm[kind == 3]; ml["nrm"] = nrm.astype(np.float32).astype(np.float64)
        lvl = rng.choice([0, 0, 0, 1, 1, 2, 4], nml); ml["maxDist"] = (dist * 1.2 ** lvl * rng.uniform(0.86, 0.99, nml)).astype(np.float32)
        ml["minDist"] = (ml["maxDist"] / f32(1.2 ** 7)).astype(np.float32); ml["bad"] = kind == 4
        mld = sld.copy()
        for i in np.nonzero(rng.random(nml) < 0.4)[0]: mld[i, rng.integers(0, 32, 4)] ^= rng.integers(1, 256, 4).astype(np.uint8)
        # -- SearchByProjection(KF, Scw, lines, matched, th)
        m_in = np.full(nl2, -1, np.int32); r_ = rng.random(nl2); m_in[r_ < 0.15] = -2
        pick = np.nonzero((r_ >= 0.15) & (r_ < 0.25))[0]; m_in[pick] = rng.permutation(nml)[:len(pick)]
        found = np.zeros(nml, np.uint8); found[m_in[m_in >= 0]] = 1
        fql = np.zeros(nml, FQL)
        R.ref_line_sim3_queries(_p(bb), _p(scale8), C.c_float(log_sf), _p(cam), _p(Scw), _p(found), _p(ml), nml, C.c_float(th), _p(fql))
        assert (fql["valid"] != 2).all()
        m_ref = m_in.copy()
        nr = R.ref_line_search_by_projection_sim3(_p(kl2), _p(ld2), nl2, _p(bb), _p(scale8), C.c_float(log_sf), _p(cam), _p(Scw), _p(ml), _p(mld), nml, int(th), _p(m_ref))
        q = np.zeros(nml, fe.PQ_DTYPE)
        q["u"] = fql["u1"]; q["v"] = fql["v1"]; q["u2"] = fql["u2"]; q["v2"] = fql["v2"]; q["radius"] = fql["radius"]; q["min_level"] = fql["level"] - 1; q["max_level"] = fql["level"]; q["valid"] = fql["valid"]
        a, no = orc.search_by_projection_sim3(1, kl2, ld2, q, mld, (m_in != -1).astype(np.uint8), bounds=tuple(bb))
        want = np.where(a >= 0, a, m_in).astype(np.int32)
        note("LSDmatcher::SearchByProjection(KF, Scw, lines) %s s=%g th=%g %s" % (name, scale, th, "own lines" if same else "other view"), nr == no and np.array_equal(m_ref, want),
             candidates=int(nml), already_found=int(found.sum()), projected=int((fql["valid"] == 1).sum()), matches=int(nr))
        # -- Fuse(KF, Scw, lines, th, replace)
        state = rng.choice([0, 0, 1, 1, 2], nl2).astype(np.uint8)
        slot = np.full(nml, -1, np.int32); own = rng.permutation(nml)[:nml // 10]; slot[own] = rng.permutation(nl2)[:len(own)]
        inkf = ((slot >= 0) & (ml["bad"] == 0)).astype(np.uint8)
        R.ref_line_sim3_queries(_p(bb), _p(scale8), C.c_float(log_sf), _p(cam), _p(Scw), _p(inkf), _p(ml), nml, C.c_float(th), _p(fql))
        fused = np.zeros(nml, np.int32); act = np.zeros(nml, np.int32)
        nr = R.ref_line_fuse_sim3(_p(kl2), _p(ld2), nl2, _p(bb), _p(scale8), C.c_float(log_sf), _p(state), _p(slot), _p(cam), _p(Scw), _p(ml), _p(mld), nml, C.c_float(th), _p(fused), _p(act))
        q["u"] = fql["u1"]; q["v"] = fql["v1"]; q["u2"] = fql["u2"]; q["v2"] = fql["v2"]; q["radius"] = fql["radius"]; q["min_level"] = fql["level"] - 1; q["max_level"] = fql["level"]; q["valid"] = fql["valid"]
        bi, bd = orc.fuse_search(1, 0, kl2, ld2, q, mld, bounds=tuple(bb))
        want = np.where((fql["valid"] == 1) & (bi >= 0) & (bd <= 50), bi, -1)
        note("LSDmatcher::Fuse(KeyFrame, Scw, lines) %s s=%g th=%g %s" % (name, scale, th, "own lines" if same else "other view"), nr == int((want >= 0).sum()) and np.array_equal(fused, want),
             candidates=int(nml), already_in_keyframe=int(inkf.sum()), projected=int((fql["valid"] == 1).sum()), fused=int(nr), added=int((act == 1).sum()), to_replace=int((act == 2).sum()))

# --- LSDmatcher::SearchBySim3 (src/LSDmatcher.cpp:685-929): both directions = fuse_search, kind 1, TH_HIGH, on the windows of the reference's two projection blocks; agreement restated here.
for name in ("synth1234", "synth2000", "big1235"):
    img = frames[name][0]; h, w = img.shape; bb = np.array((0.0, float(w), 0.0, float(h)), np.float32)
    kl2, ld2 = orc.lines_extract(img, 200)[:2]; n = len(kl2)
    fx = fy = f32(0.9 * w); cxx, cyy = f32(w / 2 - 3.25), f32(h / 2 + 1.5); cam = np.array([fx, fy, cxx, cyy, 40.0], np.float32)
    ay, ax = 0.02, 0.04
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]]); Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Rw = Ry @ Rx; tw = np.array([-0.1, 0.2, 0.3])
    T1w = np.eye(4, dtype=np.float32); T1w[:3, :3] = Rw.astype(np.float32); T1w[:3, 3] = tw.astype(np.float32); T2w = T1w.copy()
    for s12, ang, th in ((1.0, 0.0, 7.5), (1.02, 0.003, 7.5), (0.98, -0.004, 4.0)):
        Rz = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]]); R12 = Rz.astype(np.float32); t12 = np.array([0.01, -0.02, 0.01 * (s12 != 1.0)], np.float32)

        def lines3d():
            z = rng.uniform(2.0, 9.0, n); dz = rng.uniform(-0.3, 0.3, n)
            def unproj(x, y, zz):
                pc = np.stack([(x.astype(np.float64) + rng.uniform(-1.5, 1.5, n) - float(cxx)) / float(fx) * zz, (y.astype(np.float64) + rng.uniform(-1.5, 1.5, n) - float(cyy)) / float(fy) * zz, zz], 1)
                return pc, (Rw.T @ (pc - tw).T).T
            pcs, sp = unproj(kl2["startPointX"], kl2["startPointY"], z); pce, ep = unproj(kl2["endPointX"], kl2["endPointY"], z + dz)
            ml = np.zeros(n, FML); ml["wp"] = np.concatenate([sp, ep], 1).astype(np.float32).astype(np.float64)
            dist = np.linalg.norm(0.5 * (pcs + pce), axis=1)
            lvl = rng.choice([0, 0, 0, 1, 1, 2], n); ml["maxDist"] = (dist * 1.2 ** lvl * rng.uniform(0.88, 0.97, n)).astype(np.float32)
            ml["minDist"] = (ml["maxDist"] / f32(1.2 ** 7)).astype(np.float32); ml["bad"] = rng.random(n) < 0.04
            return ml
        ml1, ml2 = lines3d(), lines3d()
        mld1, mld2 = ld2.copy(), ld2.copy()
        for dd in (mld1, mld2):
            for i in np.nonzero(rng.random(n) < 0.4)[0]: dd[i, rng.integers(0, 32, 5)] ^= rng.integers(1, 256, 5).astype(np.uint8)
        present1 = (rng.random(n) < 0.85).astype(np.uint8); present2 = (rng.random(n) < 0.85).astype(np.uint8)
        m_in = np.full(n, -1, np.int32); r_ = rng.random(n); m_in[r_ < 0.06] = -2
        pick = np.nonzero((r_ >= 0.06) & (r_ < 0.16))[0]; m_in[pick] = rng.choice(np.nonzero(present2)[0], len(pick), replace=False)
        m_ref = m_in.copy()
        nr = R.ref_line_search_by_sim3(_p(kl2), _p(ld2), n, _p(kl2), _p(ld2), n, _p(bb), _p(scale8), C.c_float(log_sf), _p(cam), _p(T1w), _p(T2w), C.c_float(s12), _p(R12), _p(t12),
                                       _p(present1), _p(ml1), _p(mld1), _p(present2), _p(ml2), _p(mld2), C.c_float(th), _p(m_ref))
        already1 = m_in != -1; already2 = np.zeros(n, bool); already2[m_in[m_in >= 0]] = True
        vn = []
        for direction, present, already, ml, mld in ((0, present1, already1, ml1, mld1), (1, present2, already2, ml2, mld2)):
            skip = ((present == 0) | already).astype(np.uint8); fql = np.zeros(n, FQL)
            R.ref_line_sim3_pair_queries(direction, _p(bb), _p(scale8), C.c_float(log_sf), _p(cam), _p(T1w), _p(T2w), C.c_float(s12), _p(R12), _p(t12), _p(skip), _p(ml), n, C.c_float(th), _p(fql))
            assert (fql["valid"] != 2).all()
            q = np.zeros(n, fe.PQ_DTYPE)
            q["u"] = fql["u1"]; q["v"] = fql["v1"]; q["u2"] = fql["u2"]; q["v2"] = fql["v2"]; q["radius"] = fql["radius"]; q["min_level"] = fql["level"] - 1; q["max_level"] = fql["level"]; q["valid"] = fql["valid"]
            bi, bd = orc.fuse_search(1, 0, kl2, ld2, q, mld, bounds=tuple(bb))
            vn.append(np.where((fql["valid"] == 1) & (bi >= 0) & (bd <= 100), bi, -1))
        want = m_in.copy(); found = 0
        for i1 in range(n):
            i2 = vn[0][i1]
            if i2 >= 0 and vn[1][i2] == i1: want[i1] = i2 if present2[i2] else -1; found += 1
        note("LSDmatcher::SearchBySim3 %s s12=%g rot=%g th=%g" % (name, s12, ang, th), nr == found and np.array_equal(m_ref, want), lines1=int(present1.sum()), lines2=int(present2.sum()),
             already_matched=int(already1.sum()), one_way=int((vn[0] >= 0).sum()), other_way=int((vn[1] >= 0).sum()), agreed=int(found))

# --- MapPoint / MapLine::ComputeDistinctiveDescriptors (src/MapPoint.cc:247-312, src/MapLine.cpp:246-317): least median Hamming distance to the others
ok_all = True; nsets = 0
base = frames["synth2000"][2][1]
for trial in range(400):
    n = int(rng.integers(1, 40)) if trial % 7 else int(rng.integers(40, 120))
    seed_row = base[int(rng.integers(0, len(base)))]
    desc = np.tile(seed_row, (n, 1)).copy()
    for i in range(n):                                           # observations of one point: near-duplicates (ties are the interesting case), a few outliers
        for b_ in rng.integers(0, 256, int(rng.integers(0, 12)) if rng.random() < 0.85 else 90): desc[i, b_ >> 3] ^= np.uint8(1 << (b_ & 7))
    if trial % 5 == 0: desc[1:] = desc[0]                          # all identical
    bad = (rng.random(n) < 0.15).astype(np.uint8) if trial % 3 == 0 else np.zeros(n, np.uint8)
    good = np.ascontiguousarray(desc[bad == 0])
    want = int(orc.distinctive(good, np.array([0, len(good)], np.int32))[0]) if len(good) else -1
    for lines in (0, 1):
        got = R.ref_distinctive(lines, _p(np.ascontiguousarray(desc)), n, _p(bad))
        # identical descriptors: the reference returns a CLONE, so any row that holds the chosen bytes is the same answer
        same = got == want or (got >= 0 and want >= 0 and np.array_equal(good[got], good[want]))
        ok_all &= bool(same); nsets += 1
note("MapPoint / MapLine::ComputeDistinctiveDescriptors", ok_all, observation_sets=nsets)

# --- Frame::ComputeBoW (src/Frame.cc:474-481) over the reference's own vendored DBoW2, compiled unmodified (oracle/_ref/libref_dbow2.so): ORBVocabulary::
# loadFromTextFile (TemplatedVocabulary.h:1338-1423: what System.cc:64-73 calls on ORBvoc.txt) + transform(features, BowVector&, FeatureVector&, levelsup)
# (:1126-1208, tree descent :1216-1259, FORB::distance) against the oracle's restatement of the loader and the transform.  ORBvoc.txt itself is a git-LFS
# pointer in the reference tree, so the vocabularies are synthetic trees written in its text format (ragged branches, stopped words, all weightings / scorings).
import tempfile
from synth import synthetic_vocab, write_vocab_text
D2 = C.CDLL(os.path.join(ref_dir, "libref_dbow2.so"))
kpv, dv = frames["synth1234"][2]
d9 = {"files_with_trailing_newline": 0, "phantom_words": 0, "bow_entries_differing": 0, "bow_entries": 0}
with tempfile.TemporaryDirectory() as td:
    for vi, (k, Lv, weighting, scoring, levelsup, fmt) in enumerate([(10, 4, 0, 0, 4, "%r"), (9, 3, 1, 1, 2, "%.6g"), (9, 3, 2, 2, 0, "%.6g"), (10, 4, 3, 5, 6, "%r"), (5, 5, 0, 3, 3, "%r"), (10, 3, 0, 4, 1, "%r")]):
        Lx, ptr, ch, nd, word, weight = synthetic_vocab(np.random.default_rng(100 + vi), k=k, L=Lv)
        feat = np.ascontiguousarray(np.concatenate([dv, nd[np.random.default_rng(vi).integers(1, len(nd), 150)]]))
        n = len(feat)
        for nl in (False, True):
            path = os.path.join(td, "voc%d_%d.txt" % (vi, nl))
            write_vocab_text(path, k, Lx, ptr, ch, nd, weight, scoring=scoring, weighting=weighting, weight_fmt=fmt, trailing_newline=nl)
            info = np.zeros(5, np.int32); bw = np.zeros(n, np.int32); bv = np.zeros(n, np.float64); fnn = np.zeros(n, np.int32); fpp = np.zeros(n + 1, np.int32); fff = np.zeros(n, np.int32)
            nb = C.c_int32(); nf = C.c_int32()
            rc = D2.ref_vocab_compute_bow(path.encode(), _p(feat), n, levelsup, _p(info), _p(bw), _p(bv), C.byref(nb), _p(fnn), _p(fpp), _p(fff), C.byref(nf))
            ov = orc.vocab_load_text(path)
            obw, obv, ofn, ofp, off = orc.compute_bow(ov["levels"], ov["child_ptr"], ov["children"], ov["node_desc"], ov["word_id"], ov["weight"], feat, levelsup, ov["weighting"], ov["scoring"])
            if not nl:
                # D10.  A word that sits SHALLOWER than level L - levelsup never reaches the line that stores its FeatureVector node (TemplatedVocabulary.h:1250-1251),
                # and the caller's `NodeId nid;` is uninitialised (:1150): the reference files such a feature under whatever the previous feature left on the stack.
                # The oracle (and the library) file it under the root, node 0.  Such features are compared on the BowVector only and counted.
                cp, chd = ov["child_ptr"], ov["children"]; depth = np.zeros(len(cp) - 1, np.int32)
                for nd_ in range(len(cp) - 1):
                    for c_ in chd[cp[nd_]:cp[nd_ + 1]]: depth[c_] = depth[nd_] + 1          # (DBoW2 numbers children after their parents)
                leaf_of_word = {int(wd): i_ for i_, wd in enumerate(ov["word_id"]) if cp[i_ + 1] == cp[i_] and i_ > 0}
                fw, fwt, fnode = orc.bow_transform(ov["levels"], cp, chd, ov["node_desc"], ov["word_id"], ov["weight"], feat, levelsup)
                undefined = set(i_ for i_ in range(n) if fwt[i_] > 0 and ov["levels"] - levelsup > 0 and depth[leaf_of_word[int(fw[i_])]] < ov["levels"] - levelsup)
                def fv_pairs(nodes, ptr_, feats):
                    return sorted((int(nodes[j]), int(f_)) for j in range(len(nodes)) for f_ in feats[ptr_[j]:ptr_[j + 1]] if int(f_) not in undefined)
                ok = (rc == 0 and list(info) == [ov["k"], ov["levels"], ov["scoring"], ov["weighting"], ov["nwords"]] and nb.value == len(obw) and np.array_equal(bw[:nb.value], obw)
                      and np.array_equal(bv[:nb.value].view(np.uint64), obv.view(np.uint64)) and fv_pairs(fnn[:nf.value], fpp, fff) == fv_pairs(ofn, ofp, off))
                d9["features_on_words_above_the_node_level"] = d9.get("features_on_words_above_the_node_level", 0) + len(undefined)
                note("DBoW2 loadFromTextFile + transform k=%d L=%d weighting=%d scoring=%d levelsup=%d" % (k, Lv, weighting, scoring, levelsup), ok, words=int(nb.value), nodes=int(nf.value),
                     vocabulary_words=int(info[4]), features_with_undefined_node=len(undefined))
            else:
                # D9.  A file that ends with a newline (as ORBvoc.txt does) makes the reference's `while(!f.eof())` loop read one more, empty line: `ssnode >> pid` fails
                # before storing, so the phantom node's parent and leaf flag are whatever the previous iteration left on the stack (TemplatedVocabulary.h:1381-1398, undefined
                # behaviour), its descriptor is an unfilled 32-byte row.  The oracle (and the library) stop at the last real line.  Recorded, not asserted: the reference's
                # answer here depends on its own stack and heap.
                ref_set = dict(zip(bw[:nb.value].tolist(), bv[:nb.value].tolist())); orc_set = dict(zip(obw.tolist(), obv.tolist()))
                d9["files_with_trailing_newline"] += 1; d9["phantom_words"] += int(info[4]) - int(ov["nwords"])
                d9["bow_entries"] += len(orc_set); d9["bow_entries_differing"] += sum(1 for w_ in set(ref_set) | set(orc_set) if ref_set.get(w_) != orc_set.get(w_))
    bad = os.path.join(td, "missing.txt")
    note("DBoW2 loadFromTextFile on a missing file", D2.ref_vocab_compute_bow(bad.encode(), _p(feat), n, 2, _p(info), _p(bw), _p(bv), C.byref(nb), _p(fnn), _p(fpp), _p(fff), C.byref(nf)) == -1)
report["d9_trailing_newline"] = d9
print("D9 (vocabulary file with a trailing newline: the reference reads a phantom node, undefined behaviour):", d9)

# --- D2 (LSD seed order inside a gradient bin): NOT a comparison with reference code -- both sides are the oracle's LSD (UPSTREAM-RECALL), once with the stable order the
# oracle and the library define, once with upstream's std::sort as recalled.  Recorded as the size of what that decision can move.
def unmatched(a, b, tol=1.0):
    """segments of a without a counterpart in b whose two end points lie within tol pixels (either orientation)"""
    if len(a) == 0: return 0
    if len(b) == 0: return len(a)
    A = a.view(np.float32).reshape(-1, 4).astype(np.float64); B = b.view(np.float32).reshape(-1, 4).astype(np.float64)
    d1 = np.maximum(np.hypot(A[:, None, 0] - B[None, :, 0], A[:, None, 1] - B[None, :, 1]), np.hypot(A[:, None, 2] - B[None, :, 2], A[:, None, 3] - B[None, :, 3]))
    d2_ = np.maximum(np.hypot(A[:, None, 0] - B[None, :, 2], A[:, None, 1] - B[None, :, 3]), np.hypot(A[:, None, 2] - B[None, :, 0], A[:, None, 3] - B[None, :, 1]))
    return int((np.minimum(d1, d2_).min(axis=1) > tol).sum())


d2 = {"frames": 0, "segments": 0, "segments_in_one_set_only": 0, "segments_moved_more_than_1px": 0}; d7 = {"frames": 0, "segments": 0, "segments_in_one_set_only": 0, "segments_moved_more_than_1px": 0}
# D11 (nfa()'s first term: (double(n) + 1), the default | log_gamma(n + 1)) and D12 (LBD bit order: 0x80 >> i, the default | 1 << i): the same kind of row -- oracle against oracle, the size of what the
# decision moves.  Both alternatives are selectable in the library too (sslam_lines_set_nfa_variant / _lbd_bit_order; tests/test_variants_gpu.py compares library and oracle under each).
d11 = {"frames": 0, "segments": 0, "segments_variant_0": 0, "segments_in_both": 0, "segments_in_one_set_only": 0, "segments_moved_more_than_1px": 0,
       "keylines_after_cap_200": 0, "keylines_after_cap_200_variant_0": 0}
d12 = {"frames": 0, "lines": 0, "lbd_bytes": 0, "lbd_bytes_differing": 0, "lbd_bytes_not_the_bit_reversal": 0, "hamming_distances_differing": 0, "line_matcher_results_differing": 0}
_rev = np.array([int("{:08b}".format(i)[::-1], 2) for i in range(256)], np.uint8)
_prev = None
for img in [frames["synth1234"][0], frames["synth2000"][0], frames["big1235"][0], synth_frame(2001), synth_frame(2002), synth_frame(91, w=333, h=251),
            np.load(os.path.join(HERE, "..", "..", "tests", "golden", "icl_input_gray.npz"))["gray"]]:
    a = orc.lines_extract(img, 400)[3]
    sa = set(map(bytes, a.view(np.uint8).reshape(len(a), -1)))
    for setter, acc in ((O.orc_set_lsd_seed_sort, d2), (O.orc_set_lsd_resize, d7)):
        old_ = setter(1)
        try: b = orc.lines_extract(img, 400)[3]
        finally: setter(old_)
        sb = set(map(bytes, b.view(np.uint8).reshape(len(b), -1)))
        acc["frames"] += 1; acc["segments"] += len(sa); acc["segments_in_one_set_only"] += len(sa ^ sb)
        acc["segments_moved_more_than_1px"] += unmatched(a, b) + unmatched(b, a)
    # D11: variant 0 (von Gioi's binomial coefficient) against the default, variant 1
    old_ = O.orc_set_lsd_nfa_variant(0)
    try: r0 = orc.lines_extract(img, 200); a0 = orc.lines_extract(img, 400)[3]
    finally: O.orc_set_lsd_nfa_variant(old_)
    r1 = orc.lines_extract(img, 200)
    sa0 = set(map(bytes, a0.view(np.uint8).reshape(len(a0), -1)))
    d11["frames"] += 1; d11["segments_variant_0"] += len(sa0); d11["segments"] += len(sa); d11["segments_in_both"] += len(sa & sa0); d11["segments_in_one_set_only"] += len(sa ^ sa0)
    d11["segments_moved_more_than_1px"] += unmatched(a, a0) + unmatched(a0, a); d11["keylines_after_cap_200_variant_0"] += len(r0[0]); d11["keylines_after_cap_200"] += len(r1[0])
    # D12: variant 0 (LSB first) against the default
    old_ = O.orc_set_lbd_bit_order(0)
    try: q0 = orc.lines_extract(img, 200)
    finally: O.orc_set_lbd_bit_order(old_)
    d12["frames"] += 1; d12["lines"] += len(r1[0]); d12["lbd_bytes"] += r1[1].size; d12["lbd_bytes_differing"] += int((r1[1] != q0[1]).sum()); d12["lbd_bytes_not_the_bit_reversal"] += int((_rev[r1[1]] != q0[1]).sum())
    if _prev is not None and len(_prev[0]) >= 2 and len(r1[1]) >= 2:
        d12["hamming_distances_differing"] += int((orc.hamming_matrix(_prev[0], r1[1]) != orc.hamming_matrix(_prev[1], q0[1])).sum())
        for gate, ratio in ((0.5, False), (0.1, False), (0.5, True)):
            m0 = orc.line_match(_prev[0], r1[1], gate, ratio); m1 = orc.line_match(_prev[1], q0[1], gate, ratio)
            d12["line_matcher_results_differing"] += int(not (np.array_equal(m0[0], m1[0]) and m0[1:] == m1[1:]))
    _prev = (r1[1], q0[1])
report["d11_nfa_variant_error_bar_oracle_only"] = d11
print("D11 (nfa() first term log_gamma(n + 1) vs (n + 1), oracle against oracle):", d11)
report["d12_lbd_bit_order_oracle_only"] = d12
print("D12 (LBD bit order 1 << i vs 0x80 >> i, oracle against oracle):", d12)
report["d7_error_bar_oracle_only"] = d7
print("D7 (INTER_LINEAR_EXACT vs INTER_LINEAR for LSD's 0.8x scale, oracle against oracle):", d7)
report["d2_error_bar_oracle_only"] = d2
print("D2 (stable seed order vs upstream's std::sort, oracle against oracle):", d2)

json.dump(report, open(report_path, "w"), indent=1)
print("reference slices == oracle on every case:", report["all_equal"])
