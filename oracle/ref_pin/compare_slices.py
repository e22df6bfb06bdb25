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
rng = np.random.Generator(np.random.PCG64(20260926))
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

json.dump(report, open(report_path, "w"), indent=1)
print("reference slices == oracle on every case:", report["all_equal"])
