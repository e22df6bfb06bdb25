"""ctypes binding of oracle/liboracle.so — the CPU restatement used ONLY as the checker
(tests, smoke, bench cpu_baseline).  Builds the oracle with its Makefile on first use."""
import ctypes as C
import os, subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
OLIB = os.path.join(ODIR, "liboracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def build():
    subprocess.check_call(["make", "-s", "-C", ODIR, "liboracle.so"])
    return OLIB


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


def build_native():
    """-O3 -march=native build for the CPU-baseline timing leg (oracle/Makefile `native`); None if the host compiler refuses"""
    try:
        subprocess.check_call(["make", "-s", "-C", ODIR, "native"])
        return os.path.join(ODIR, "_native", "liboracle_native.so")
    except Exception:
        return None


class Oracle:
    def __init__(self, native=False):
        build()
        self.native = False
        path = OLIB
        if native:
            p = build_native()
            if p and os.path.exists(p):
                path, self.native = p, True
        path = os.environ.get("SSLAM_ORACLE_LIB", path)      # (tests/test_sanitize_cpu.py: the same sources built with ASan + UBSan)
        self.L = C.CDLL(path)
        self.L.orc_fast_atan2.restype = C.c_float
        self.L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]

    def set_gauss_variant(self, v):
        """0: decision D6 (bit-exact 8.8 taps that sum to 256); 1: OpenCV 3.4.0's rounded taps (sum 257).  Process-global; returns the previous value."""
        return int(self.L.orc_set_gauss_variant(int(v)))

    def orb_params(self, nfeatures=1000, scale=1.2, nlevels=8):
        s = np.zeros(nlevels, np.float32); p = np.zeros(nlevels, np.int32); u = np.zeros(16, np.int32)
        self.L.orc_orb_params(nfeatures, C.c_float(scale), nlevels, _p(s), _p(p), _p(u))
        return s, p, u

    def orb_extract(self, gray, nfeatures=1000, scale=1.2, nlevels=8, ini=20, mn=7):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        cap = nfeatures + 64 * nlevels
        kp = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        n = self.L.orc_orb_extract(_p(gray), w, h, gray.strides[0], nfeatures, C.c_float(scale), nlevels, ini, mn, _p(kp), _p(desc), cap)
        assert n <= cap
        return kp[:n].copy(), desc[:n].copy()

    def pyramid_level(self, gray, level, scale=1.2, nlevels=8):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        lw = C.c_int(0); lh = C.c_int(0)
        self.L.orc_orb_pyramid_level(_p(gray), w, h, gray.strides[0], C.c_float(scale), nlevels, level, C.c_void_p(0), C.byref(lw), C.byref(lh))
        out = np.zeros((lh.value, lw.value), np.uint8)
        self.L.orc_orb_pyramid_level(_p(gray), w, h, gray.strides[0], C.c_float(scale), nlevels, level, _p(out), C.byref(lw), C.byref(lh))
        return out

    def candidates(self, gray, level, nfeatures=1000, scale=1.2, nlevels=8, ini=20, mn=7):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        cap = 400000
        out = np.zeros((cap, 3), np.int32)
        n = self.L.orc_orb_candidates(_p(gray), w, h, gray.strides[0], nfeatures, C.c_float(scale), nlevels, ini, mn, level, _p(out), cap)
        return out[:n].copy()

    def blur7(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        out = np.zeros_like(img)
        self.L.orc_blur7(_p(img), img.shape[1], img.shape[0], _p(out))
        return out

    def gauss_taps(self, n, sigma):
        t = np.zeros(n, np.int32)
        self.L.orc_gauss_taps(n, C.c_double(sigma), _p(t))
        return t

    def fast_atan2(self, y, x):
        return float(self.L.orc_fast_atan2(float(y), float(x)))

    def fast_score(self, patch7):
        patch7 = np.ascontiguousarray(patch7, np.uint8)
        return int(self.L.orc_fast_score(_p(patch7)))


def _orc_match_methods():
    def descriptor_distance(self, a, b):
        a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
        return int(self.L.orc_descriptor_distance(_p(a), _p(b)))

    def knn2(self, q, t):
        q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
        idx = np.zeros((len(q), 2), np.int32); dist = np.zeros((len(q), 2), np.int32)
        self.L.orc_knn2(_p(q), len(q), _p(t), len(t), _p(idx), _p(dist))
        return idx, dist

    def hamming_matrix(self, q, t):
        q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
        D = np.zeros((len(q), len(t)), np.uint16)
        self.L.orc_hamming_matrix(_p(q), len(q), _p(t), len(t), _p(D))
        return D

    def search_for_initialization(self, kp1, d1, kp2, d2, prev_matched, window=100, nnratio=0.9, check_orientation=True,
                                  bounds=(0.0, 640.0, 0.0, 480.0)):
        kp1 = np.ascontiguousarray(kp1); kp2 = np.ascontiguousarray(kp2)
        d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        pm = np.ascontiguousarray(prev_matched, np.float32).copy()
        m12 = np.full(len(kp1), -1, np.int32)
        b = np.array(bounds, np.float32)
        n = self.L.orc_search_for_initialization(_p(kp1), _p(d1), len(kp1), _p(kp2), _p(d2), len(kp2), _p(pm), _p(m12),
                                                 int(window), C.c_float(nnratio), int(bool(check_orientation)), _p(b))
        return m12, pm, n

    def line_match(self, l1, l2, gate_scale=0.5, ratio_mode=False):
        l1 = np.ascontiguousarray(l1, np.uint8); l2 = np.ascontiguousarray(l2, np.uint8)
        cap = max(len(l1), 1)
        pairs = np.zeros((cap, 2), np.int32)
        mad = C.c_double(0); mad12 = C.c_double(0)
        n = self.L.orc_line_match(_p(l1), len(l1), _p(l2), len(l2), C.c_double(gate_scale), int(bool(ratio_mode)), _p(pairs), cap,
                                  C.byref(mad), C.byref(mad12))
        return pairs[:n].copy(), mad.value, mad12.value

    def search_by_projection(self, kind, mode, feats, desc, queries, qdesc, occupied=None, uright=None, nnratio=0.8, th_dist=100,
                             check_orientation=True, bounds=(0.0, 640.0, 0.0, 480.0)):
        feats = np.ascontiguousarray(feats); desc = np.ascontiguousarray(desc, np.uint8)
        queries = np.ascontiguousarray(queries); qdesc = np.ascontiguousarray(qdesc, np.uint8)
        n = len(feats)
        assigned = np.full(n, -1, np.int32)
        b = np.array(bounds, np.float32)
        occ = np.zeros(n, np.uint8) if occupied is None else np.ascontiguousarray(occupied, np.uint8)
        ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
        nm = self.L.orc_search_by_projection(int(kind), int(mode), _p(feats), _p(desc), n, _p(b), _p(ur), _p(occ), _p(queries), _p(qdesc),
                                             len(queries), C.c_float(nnratio), int(th_dist), int(bool(check_orientation)), _p(assigned))
        return assigned, nm

    def search_by_projection_reloc(self, feats, desc, queries, qdesc, has_map_point, orb_dist=100, check_orientation=True, bounds=(0.0, 640.0, 0.0, 480.0)):
        """direct restatement of ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist), src/ORBmatcher.cc:1475-1602"""
        feats = np.ascontiguousarray(feats); desc = np.ascontiguousarray(desc, np.uint8)
        queries = np.ascontiguousarray(queries); qdesc = np.ascontiguousarray(qdesc, np.uint8)
        n = len(feats); assigned = np.full(n, -1, np.int32); b = np.array(bounds, np.float32)
        hm = np.ascontiguousarray(has_map_point, np.uint8)
        nm = self.L.orc_search_by_projection_reloc(_p(feats), _p(desc), n, _p(b), _p(hm), _p(queries), _p(qdesc), len(queries), int(orb_dist),
                                                   int(bool(check_orientation)), _p(assigned))
        return assigned, nm

    def search_by_projection_sim3(self, kind, feats, desc, queries, qdesc, matched, bounds=(0.0, 640.0, 0.0, 480.0)):
        """direct restatement of the loop-closing SearchByProjection(KeyFrame*, Scw, ...): src/ORBmatcher.cc:293-406, src/LSDmatcher.cpp:558-683"""
        feats = np.ascontiguousarray(feats); desc = np.ascontiguousarray(desc, np.uint8)
        queries = np.ascontiguousarray(queries); qdesc = np.ascontiguousarray(qdesc, np.uint8)
        n = len(feats); assigned = np.full(n, -1, np.int32); b = np.array(bounds, np.float32)
        mt = np.ascontiguousarray(matched, np.uint8)
        nm = self.L.orc_search_by_projection_sim3(int(kind), _p(feats), _p(desc), n, _p(b), _p(mt), _p(queries), _p(qdesc), len(queries), _p(assigned))
        return assigned, nm

    def search_by_bow(self, kf_kp, kf_desc, kf_valid, f_kp, f_desc, ptr_kf, ptr_f, idx_kf, idx_f, nnratio=0.9, check_orientation=True):
        kf_kp = np.ascontiguousarray(kf_kp); f_kp = np.ascontiguousarray(f_kp)
        kf_desc = np.ascontiguousarray(kf_desc, np.uint8); f_desc = np.ascontiguousarray(f_desc, np.uint8)
        kf_valid = np.ascontiguousarray(kf_valid, np.uint8)
        ptr_kf = np.ascontiguousarray(ptr_kf, np.int32); ptr_f = np.ascontiguousarray(ptr_f, np.int32)
        idx_kf = np.ascontiguousarray(idx_kf, np.int32); idx_f = np.ascontiguousarray(idx_f, np.int32)
        assigned = np.full(len(f_kp), -1, np.int32)
        nm = self.L.orc_search_by_bow(_p(kf_kp), _p(kf_desc), _p(kf_valid), _p(f_kp), _p(f_desc), len(f_kp), _p(ptr_kf), _p(ptr_f),
                                      len(ptr_kf) - 1, _p(idx_kf), _p(idx_f), C.c_float(nnratio), int(bool(check_orientation)), _p(assigned))
        return assigned, nm

    def search_by_bow_keyframes(self, kp1, d1, valid1, kp2, d2, valid2, ptr1, ptr2, idx1, idx2, nnratio=0.8, check_orientation=True):
        kp1 = np.ascontiguousarray(kp1); kp2 = np.ascontiguousarray(kp2)
        d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        valid1 = np.ascontiguousarray(valid1, np.uint8); valid2 = np.ascontiguousarray(valid2, np.uint8)
        ptr1 = np.ascontiguousarray(ptr1, np.int32); ptr2 = np.ascontiguousarray(ptr2, np.int32)
        idx1 = np.ascontiguousarray(idx1, np.int32); idx2 = np.ascontiguousarray(idx2, np.int32)
        m12 = np.full(len(kp1), -1, np.int32)
        nm = self.L.orc_search_by_bow_keyframes(_p(kp1), _p(d1), _p(valid1), len(kp1), _p(kp2), _p(d2), _p(valid2), len(kp2), _p(ptr1), _p(ptr2),
                                                len(ptr1) - 1, _p(idx1), _p(idx2), C.c_float(nnratio), int(bool(check_orientation)), _p(m12))
        return m12, nm

    def fuse_search(self, kind, chi2, feats, desc, queries, qdesc, uright=None, inv_level_sigma2=None, bounds=(0.0, 640.0, 0.0, 480.0)):
        feats = np.ascontiguousarray(feats); desc = np.ascontiguousarray(desc, np.uint8)
        queries = np.ascontiguousarray(queries); qdesc = np.ascontiguousarray(qdesc, np.uint8)
        b = np.asarray(bounds, np.float32)
        ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
        sg = np.ones(8, np.float32) if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
        bi = np.zeros(len(queries), np.int32); bd = np.zeros(len(queries), np.int32)
        self.L.orc_fuse_search(int(kind), int(chi2), _p(feats), _p(desc), len(feats), _p(b), _p(ur), _p(sg), _p(queries), _p(qdesc), len(queries), _p(bi), _p(bd))
        return bi, bd

    def search_for_triangulation(self, kp1, d1, ur1, free1, kp2, d2, ur2, free2, ptr1, ptr2, idx1, idx2, F12, ex, ey, scale_factors2,
                                 level_sigma2_2, only_stereo=False, check_orientation=True):
        kp1 = np.ascontiguousarray(kp1); kp2 = np.ascontiguousarray(kp2)
        d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        u1 = None if ur1 is None else np.ascontiguousarray(ur1, np.float32); u2 = None if ur2 is None else np.ascontiguousarray(ur2, np.float32)
        f1 = np.ascontiguousarray(free1, np.uint8); f2 = np.ascontiguousarray(free2, np.uint8)
        ptr1 = np.ascontiguousarray(ptr1, np.int32); ptr2 = np.ascontiguousarray(ptr2, np.int32)
        idx1 = np.ascontiguousarray(idx1, np.int32); idx2 = np.ascontiguousarray(idx2, np.int32)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        sf = np.ascontiguousarray(scale_factors2, np.float32); sg = np.ascontiguousarray(level_sigma2_2, np.float32)
        m12 = np.zeros(len(kp1), np.int32)
        self.L.orc_search_for_triangulation.argtypes = None
        n = self.L.orc_search_for_triangulation(_p(kp1), _p(d1), _p(u1), _p(f1), len(kp1), _p(kp2), _p(d2), _p(u2), _p(f2), len(kp2), _p(ptr1), _p(ptr2),
                                                len(ptr1) - 1, _p(idx1), _p(idx2), _p(F), C.c_float(ex), C.c_float(ey), _p(sf), _p(sg),
                                                int(bool(only_stereo)), int(bool(check_orientation)), _p(m12))
        return m12, n

    def bow_transform(self, levels, child_ptr, children, node_desc, word_id, weight, feat, levelsup=4):
        child_ptr = np.ascontiguousarray(child_ptr, np.int32); children = np.ascontiguousarray(children, np.int32)
        node_desc = np.ascontiguousarray(node_desc, np.uint8); word_id = np.ascontiguousarray(word_id, np.int32)
        weight = np.ascontiguousarray(weight, np.float64); feat = np.ascontiguousarray(feat, np.uint8)
        n = len(feat)
        w = np.zeros(n, np.int32); v = np.zeros(n, np.float64); nd = np.zeros(n, np.int32)
        self.L.orc_bow_transform(len(child_ptr) - 1, int(levels), _p(child_ptr), _p(children), _p(node_desc), _p(word_id), _p(weight), _p(feat), n,
                                 int(levelsup), _p(w), _p(v), _p(nd))
        return w, v, nd

    def fast_image(self, img, threshold):
        """cv::FAST(img, threshold, nms=True) on the whole image: (score map before NMS, keypoints (x, y, score) after NMS)"""
        img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
        sc = np.zeros((h, w), np.uint8); cap = w * h // 4; xys = np.zeros((cap, 3), np.int32)
        n = self.L.orc_fast_image(_p(img), w, h, w, int(threshold), _p(sc), _p(xys), cap)
        return sc, xys[:n].copy()

    def vocab_load_text(self, path):
        """-> dict(k, levels, scoring, weighting, nwords, child_ptr, children, node_desc, word_id, weight)"""
        import ctypes as C
        k, L, sc, wg, nn, nw = (C.c_int() for _ in range(6))
        rc = self.L.orc_vocab_load_text(str(path).encode(), C.byref(k), C.byref(L), C.byref(sc), C.byref(wg), C.byref(nn), C.byref(nw))
        if rc != 0:
            raise ValueError("orc_vocab_load_text failed: %d" % rc)
        n = nn.value
        cp = np.zeros(n + 1, np.int32); ch = np.zeros(max(n - 1, 1), np.int32); nd = np.zeros((n, 32), np.uint8); wi = np.zeros(n, np.int32); wt = np.zeros(n, np.float64)
        self.L.orc_vocab_arrays(_p(cp), _p(ch), _p(nd), _p(wi), _p(wt))
        return dict(k=k.value, levels=L.value, scoring=sc.value, weighting=wg.value, nwords=nw.value, child_ptr=cp, children=ch[:n - 1], node_desc=nd, word_id=wi, weight=wt)

    def compute_bow(self, levels, child_ptr, children, node_desc, word_id, weight, feat, levelsup=4, weighting=0, scoring=0):
        """-> (bow_word, bow_value, fv_node, fv_ptr, fv_feat): BowVector and FeatureVector flattened in key order"""
        import ctypes as C
        child_ptr = np.ascontiguousarray(child_ptr, np.int32); children = np.ascontiguousarray(children, np.int32)
        node_desc = np.ascontiguousarray(node_desc, np.uint8); word_id = np.ascontiguousarray(word_id, np.int32)
        weight = np.ascontiguousarray(weight, np.float64); feat = np.ascontiguousarray(feat, np.uint8)
        n = len(feat)
        bw = np.zeros(max(n, 1), np.int32); bv = np.zeros(max(n, 1), np.float64); fn = np.zeros(max(n, 1), np.int32); fp = np.zeros(n + 1, np.int32); ff = np.zeros(max(n, 1), np.int32)
        nb, nf = C.c_int32(), C.c_int32()
        self.L.orc_compute_bow(len(child_ptr) - 1, int(levels), _p(child_ptr), _p(children), _p(node_desc), _p(word_id), _p(weight), int(weighting), int(scoring),
                               _p(feat), n, int(levelsup), _p(bw), _p(bv), C.byref(nb), _p(fn), _p(fp), _p(ff), C.byref(nf))
        return bw[:nb.value], bv[:nb.value], fn[:nf.value], fp[:nf.value + 1], ff[:fp[nf.value]]

    def distinctive(self, desc, ptr):
        desc = np.ascontiguousarray(desc, np.uint8); ptr = np.ascontiguousarray(ptr, np.int32)
        best = np.zeros(len(ptr) - 1, np.int32)
        self.L.orc_distinctive(_p(desc), _p(ptr), len(ptr) - 1, _p(best))
        return best

    for f in (descriptor_distance, knn2, hamming_matrix, search_for_initialization, line_match, search_by_projection, search_by_projection_reloc, search_by_projection_sim3, search_by_bow, search_by_bow_keyframes, distinctive, fuse_search, search_for_triangulation, bow_transform, vocab_load_text, compute_bow, fast_image):
        setattr(Oracle, f.__name__, f)


_orc_match_methods()

KL_DTYPE = np.dtype([("angle", "<f4"), ("class_id", "<i4"), ("octave", "<i4"),
                     ("pt_x", "<f4"), ("pt_y", "<f4"), ("response", "<f4"), ("size", "<f4"),
                     ("startPointX", "<f4"), ("startPointY", "<f4"), ("endPointX", "<f4"), ("endPointY", "<f4"),
                     ("sPointInOctaveX", "<f4"), ("sPointInOctaveY", "<f4"),
                     ("ePointInOctaveX", "<f4"), ("ePointInOctaveY", "<f4"),
                     ("lineLength", "<f4"), ("numOfPixels", "<i4")])


def _orc_line_methods():
    def lines_extract(self, gray, max_lines=40, want_float=False):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        cap = 20000
        kl = np.zeros(cap, KL_DTYPE); ld = np.zeros((cap, 32), np.uint8); fn = np.zeros((cap, 3), np.float64)
        raw = np.zeros((cap, 4), np.float32); rn = C.c_int(0)
        fd = np.zeros((cap, 72), np.float32) if want_float else None
        n = self.L.orc_lines_extract(_p(gray), w, h, gray.strides[0], int(max_lines), _p(kl), _p(ld), _p(fn), cap, _p(raw), cap,
                                     C.byref(rn), _p(fd))
        n = min(n, cap)
        out = (kl[:n].copy(), ld[:n].copy(), fn[:n].copy(), raw[:rn.value].copy())
        return out + (fd[:n].copy(),) if want_float else out

    def lsd_scaled(self, gray):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        ow = C.c_int(0); oh = C.c_int(0)
        self.L.orc_lsd_scaled(_p(gray), w, h, gray.strides[0], C.c_void_p(0), C.byref(ow), C.byref(oh))
        out = np.zeros((oh.value, ow.value), np.uint8)
        self.L.orc_lsd_scaled(_p(gray), w, h, gray.strides[0], _p(out), C.byref(ow), C.byref(oh))
        return out

    for f in (lines_extract, lsd_scaled):
        setattr(Oracle, f.__name__, f)


_orc_line_methods()
