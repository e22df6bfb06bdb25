"""ctypes binding of libsslam_frontend.so (include/sslam_frontend.h) for the tests
and bench.py.  The product is the C-ABI library + the C++ shim in shim/; this module
is the harness-side view of the same entry points.  It never falls back to a CPU
path: loading fails loudly if the HIP library is missing, and every call raises on
a non-zero status."""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SSLAM_LIB") or os.path.join(HERE, "lib", "libsslam_frontend.so")      # SSLAM_LIB: kernel-variant experiments (tools/build_variant.sh)

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
KL_DTYPE = np.dtype([("angle", "<f4"), ("class_id", "<i4"), ("octave", "<i4"),
                     ("pt_x", "<f4"), ("pt_y", "<f4"), ("response", "<f4"), ("size", "<f4"),
                     ("startPointX", "<f4"), ("startPointY", "<f4"), ("endPointX", "<f4"), ("endPointY", "<f4"),
                     ("sPointInOctaveX", "<f4"), ("sPointInOctaveY", "<f4"),
                     ("ePointInOctaveX", "<f4"), ("ePointInOctaveY", "<f4"),
                     ("lineLength", "<f4"), ("numOfPixels", "<i4")])
assert KP_DTYPE.itemsize == 28 and KL_DTYPE.itemsize == 68

PQ_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("u2", "<f4"), ("v2", "<f4"), ("radius", "<f4"), ("min_level", "<i4"), ("max_level", "<i4"),
                     ("angle", "<f4"), ("ur", "<f4"), ("valid", "<i4"), ("obs_positive", "<i4")])
assert PQ_DTYPE.itemsize == 44

_lib = None


SSLAM_ERR_INVALID, SSLAM_ERR_NO_DEVICE, SSLAM_ERR_CAPACITY, SSLAM_ERR_HIP, SSLAM_ERR_UNSUPPORTED = -1, -2, -3, -4, -5      # include/sslam_frontend.h:31-35


class SslamError(RuntimeError):
    """raised for every non-zero status of the C ABI; `code` is the status (SSLAM_ERR_*)"""
    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SslamError("HIP library %s not built (run __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.sslam_status_str.restype = C.c_char_p
        _lib.sslam_last_error.restype = C.c_char_p
        _lib.sslam_ctx_stream.restype = C.c_void_p
    return _lib


TESTING_LIB_PATH = os.path.join(HERE, "lib", "libsslam_frontend_testing.so")
_testing = None


def testing_lib():
    """libsslam_frontend_testing.so: the product's sources compiled with -DSSLAM_TESTING -- every product entry point plus the self-tests, probes and the RCCL stand-in of
    include/sslam_testing.h.  The self-tests take a context handle; a handle made by the product library is fine (same structs, same HIP runtime)."""
    global _testing
    if _testing is None:
        if not os.path.exists(TESTING_LIB_PATH):
            raise SslamError("testing library %s not built (run __graft_entry__.build())" % TESTING_LIB_PATH)
        _testing = C.CDLL(TESTING_LIB_PATH)
        _testing.sslam_status_str.restype = C.c_char_p
        _testing.sslam_last_error.restype = C.c_char_p
        _testing.sslam_ctx_stream.restype = C.c_void_p
    return _testing


class use_testing_library:
    """with fe.use_testing_library(): every call of this binding goes to the testing library (what the group tests need: the stand-in for RCCL is bound when a group is
    CREATED, so the groups, contexts and extractors of such a test all live in that library).  Objects made inside must be closed inside."""
    def __enter__(self):
        global _lib
        lib(); self.prev = _lib; _lib = testing_lib()
        return _lib

    def __exit__(self, *a):
        global _lib
        _lib = self.prev


def _chk(rc):
    if rc != 0:
        L = lib()
        raise SslamError("%s: %s" % (L.sslam_status_str(rc).decode(), L.sslam_last_error().decode()), rc)


def _p(a):
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):          # torch tensor (device or host)
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(int(a))


class Context:
    def __init__(self, device=0):
        self.h = C.c_void_p()
        _chk(lib().sslam_ctx_create(int(device), C.byref(self.h)))

    def synchronize(self):
        _chk(lib().sslam_ctx_synchronize(self.h))

    @property
    def stream(self):
        return lib().sslam_ctx_stream(self.h)

    def close(self):
        if self.h:
            lib().sslam_ctx_destroy(self.h)
            self.h = C.c_void_p()

    # ---- Hamming ------------------------------------------------------------
    def hamming_knn2(self, q, t):
        q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
        nq, nt = q.shape[0], t.shape[0]
        idx = np.full((nq, 2), -1, np.int32); dist = np.full((nq, 2), -1, np.int32)
        _chk(lib().sslam_hamming_knn2(self.h, _p(q), nq, _p(t), nt, _p(idx), _p(dist)))
        return idx, dist

    def hamming_matrix(self, q, t):
        q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
        D = np.zeros((q.shape[0], t.shape[0]), np.uint16)
        _chk(lib().sslam_hamming_matrix(self.h, _p(q), q.shape[0], _p(t), t.shape[0], _p(D)))
        return D

    def search_for_initialization(self, kp1, d1, kp2, d2, prev_matched, window=100, nnratio=0.9,
                                  check_orientation=True, bounds=(0.0, 640.0, 0.0, 480.0)):
        kp1 = np.ascontiguousarray(kp1); kp2 = np.ascontiguousarray(kp2)
        d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        pm = np.ascontiguousarray(prev_matched, np.float32).copy()
        m12 = np.full(len(kp1), -1, np.int32)
        n = C.c_int(0)
        b = (C.c_float * 4)(*bounds)
        _chk(lib().sslam_orb_search_for_initialization(self.h, _p(kp1), _p(d1), len(kp1), _p(kp2), _p(d2), len(kp2),
                                                       _p(pm), _p(m12), int(window), C.c_float(nnratio),
                                                       int(bool(check_orientation)), b, C.byref(n)))
        return m12, pm, n.value

    def search_by_projection(self, kind, mode, feats, desc, queries, qdesc, occupied=None, uright=None, nnratio=0.8, th_dist=100,
                             check_orientation=True, bounds=(0.0, 640.0, 0.0, 480.0)):
        feats = np.ascontiguousarray(feats); desc = np.ascontiguousarray(desc, np.uint8)
        queries = np.ascontiguousarray(queries, PQ_DTYPE); qdesc = np.ascontiguousarray(qdesc, np.uint8)
        n = len(feats)
        assigned = np.full(n, -1, np.int32); nm = C.c_int(0)
        b = (C.c_float * 4)(*bounds)
        occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
        ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
        _chk(lib().sslam_search_by_projection(self.h, int(kind), int(mode), _p(feats), _p(desc), n, b, _p(ur), _p(occ), _p(queries), _p(qdesc),
                                              len(queries), C.c_float(nnratio), int(th_dist), int(bool(check_orientation)), _p(assigned), C.byref(nm)))
        return assigned, nm.value

    def search_by_bow(self, kf_kp, kf_desc, kf_valid, f_kp, f_desc, ptr_kf, ptr_f, idx_kf, idx_f, nnratio=0.9, check_orientation=True):
        kf_kp = np.ascontiguousarray(kf_kp); f_kp = np.ascontiguousarray(f_kp)
        kf_desc = np.ascontiguousarray(kf_desc, np.uint8); f_desc = np.ascontiguousarray(f_desc, np.uint8)
        kf_valid = np.ascontiguousarray(kf_valid, np.uint8)
        ptr_kf = np.ascontiguousarray(ptr_kf, np.int32); ptr_f = np.ascontiguousarray(ptr_f, np.int32)
        idx_kf = np.ascontiguousarray(idx_kf, np.int32); idx_f = np.ascontiguousarray(idx_f, np.int32)
        assigned = np.full(len(f_kp), -1, np.int32); nm = C.c_int(0)
        _chk(lib().sslam_orb_search_by_bow(self.h, _p(kf_kp), _p(kf_desc), _p(kf_valid), len(kf_kp), _p(f_kp), _p(f_desc), len(f_kp), _p(ptr_kf),
                                           _p(ptr_f), len(ptr_kf) - 1, _p(idx_kf), _p(idx_f), C.c_float(nnratio), int(bool(check_orientation)),
                                           _p(assigned), C.byref(nm)))
        return assigned, nm.value

    def search_by_bow_keyframes(self, kp1, d1, valid1, kp2, d2, valid2, ptr1, ptr2, idx1, idx2, nnratio=0.8, check_orientation=True):
        """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (src/ORBmatcher.cc:525-658): -> (matches12, nmatches)"""
        kp1 = np.ascontiguousarray(kp1); kp2 = np.ascontiguousarray(kp2)
        d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        valid1 = np.ascontiguousarray(valid1, np.uint8); valid2 = np.ascontiguousarray(valid2, np.uint8)
        ptr1 = np.ascontiguousarray(ptr1, np.int32); ptr2 = np.ascontiguousarray(ptr2, np.int32)
        idx1 = np.ascontiguousarray(idx1, np.int32); idx2 = np.ascontiguousarray(idx2, np.int32)
        m12 = np.full(max(len(kp1), 1), -1, np.int32); nm = C.c_int(0)
        _chk(lib().sslam_orb_search_by_bow_keyframes(self.h, _p(kp1), _p(d1), _p(valid1), len(kp1), _p(kp2), _p(d2), _p(valid2), len(kp2), _p(ptr1), _p(ptr2),
                                                     len(ptr1) - 1, _p(idx1), _p(idx2), C.c_float(nnratio), int(bool(check_orientation)), _p(m12), C.byref(nm)))
        return m12[:len(kp1)], nm.value

    def distinctive_descriptors(self, desc, ptr):
        """MapPoint/MapLine::ComputeDistinctiveDescriptors for every set ptr[s]..ptr[s+1] (src/MapPoint.cc:247-312)."""
        desc = np.ascontiguousarray(desc, np.uint8); ptr = np.ascontiguousarray(ptr, np.int32)
        best = np.full(len(ptr) - 1, -2, np.int32)
        _chk(lib().sslam_distinctive_descriptors(self.h, _p(desc), _p(ptr), len(ptr) - 1, _p(best)))
        return best

    def frame_upload(self, kind, feats, desc, uright=None, bounds=(0.0, 640.0, 0.0, 480.0)):
        return Frame(self, kind=kind, feats=feats, desc=desc, uright=uright, bounds=bounds)

    def line_match(self, l1, l2, gate_scale=0.5, ratio_mode=False):
        l1 = np.ascontiguousarray(l1, np.uint8); l2 = np.ascontiguousarray(l2, np.uint8)
        cap = max(len(l1), 1)
        pairs = np.zeros((cap, 2), np.int32)
        n = C.c_int(0); mad = C.c_double(0); mad12 = C.c_double(0)
        _chk(lib().sslam_line_match(self.h, _p(l1), len(l1), _p(l2), len(l2), C.c_double(gate_scale), int(bool(ratio_mode)),
                                    _p(pairs), cap, C.byref(n), C.byref(mad), C.byref(mad12)))
        return pairs[:n.value].copy(), mad.value, mad12.value


class Vocabulary:
    """Device-resident DBoW2 vocabulary tree (sslam_vocab_*): CSR children lists in DBoW2's node numbering."""

    def __init__(self, ctx, levels, child_ptr, children, node_desc, word_id, weight):
        self.ctx = ctx
        self.h = C.c_void_p()
        child_ptr = np.ascontiguousarray(child_ptr, np.int32); children = np.ascontiguousarray(children, np.int32)
        node_desc = np.ascontiguousarray(node_desc, np.uint8); word_id = np.ascontiguousarray(word_id, np.int32)
        weight = np.ascontiguousarray(weight, np.float64)
        _chk(lib().sslam_vocab_create(ctx.h, len(child_ptr) - 1, int(levels), _p(child_ptr), _p(children), _p(node_desc), _p(word_id), _p(weight), C.byref(self.h)))

    @classmethod
    def from_text_file(cls, ctx, path):
        """ORBVocabulary::loadFromTextFile (src/System.cc:64-73): an ORBvoc.txt-format file -> device-resident tree"""
        self = cls.__new__(cls)
        self.ctx = ctx
        self.h = C.c_void_p()
        _chk(lib().sslam_vocab_load_text(ctx.h, str(path).encode(), C.byref(self.h)))
        return self

    def info(self):
        v = [C.c_int() for _ in range(6)]
        _chk(lib().sslam_vocab_info(self.h, *[C.byref(x) for x in v]))
        return dict(zip(("k", "levels", "scoring", "weighting", "nnodes", "nwords"), (x.value for x in v)))

    def set_types(self, weighting=0, scoring=0):
        _chk(lib().sslam_vocab_set_types(self.h, int(weighting), int(scoring)))

    def compute_bow(self, desc, levelsup=4):
        """Frame::ComputeBoW (src/Frame.cc:474-481): -> (BowVector as {word: value}, FeatureVector as {node: [feature indices]}), both in key order"""
        if isinstance(desc, Frame):
            n = len(desc)
        else:
            desc = np.ascontiguousarray(desc, np.uint8); n = len(desc)
        m = max(n, 1)
        bw = np.zeros(m, np.int32); bv = np.zeros(m, np.float64); fn = np.zeros(m, np.int32); fp = np.zeros(n + 1, np.int32); ff = np.zeros(m, np.int32)
        nb, nf = C.c_int(), C.c_int()
        if isinstance(desc, Frame):
            _chk(lib().sslam_compute_bow_frame(self.ctx.h, self.h, desc.h, int(levelsup), _p(bw), _p(bv), C.byref(nb), _p(fn), _p(fp), _p(ff), C.byref(nf)))
        else:
            _chk(lib().sslam_compute_bow(self.ctx.h, self.h, _p(desc), n, int(levelsup), _p(bw), _p(bv), C.byref(nb), _p(fn), _p(fp), _p(ff), C.byref(nf)))
        bow = {int(bw[i]): float(bv[i]) for i in range(nb.value)}
        fv = {int(fn[j]): ff[fp[j]:fp[j + 1]].tolist() for j in range(nf.value)}
        return bow, fv

    def transform(self, desc, levelsup=4):
        """per feature: (word id, word weight, node at level L - levelsup) -- Frame::ComputeBoW's device part"""
        if isinstance(desc, Frame):
            n = len(desc)
            w = np.zeros(n, np.int32); v = np.zeros(n, np.float64); nd = np.zeros(n, np.int32)
            _chk(lib().sslam_bow_transform_frame(self.ctx.h, self.h, desc.h, int(levelsup), _p(w), _p(v), _p(nd)))
            return w, v, nd
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        w = np.zeros(n, np.int32); v = np.zeros(n, np.float64); nd = np.zeros(n, np.int32)
        _chk(lib().sslam_bow_transform(self.ctx.h, self.h, _p(desc), n, int(levelsup), _p(w), _p(v), _p(nd)))
        return w, v, nd

    def close(self):
        if self.h:
            lib().sslam_vocab_destroy(self.h); self.h = C.c_void_p()


class Frame:
    """Device-resident features of one Frame (sslam_frame_*): upload once, or snapshot what an extractor just produced."""

    def __init__(self, ctx, kind=0, feats=None, desc=None, uright=None, bounds=(0.0, 640.0, 0.0, 480.0), orb=None, lines=None):
        self.ctx = ctx
        self.h = C.c_void_p()
        b = (C.c_float * 4)(*bounds)
        if orb is not None:
            _chk(lib().sslam_frame_from_orb(orb.h, b, C.byref(self.h)))
        elif lines is not None:
            _chk(lib().sslam_frame_from_lines(lines.h, b, C.byref(self.h)))
        else:
            feats = np.ascontiguousarray(feats); desc = np.ascontiguousarray(desc, np.uint8)
            ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
            _chk(lib().sslam_frame_upload(ctx.h, int(kind), _p(feats), _p(desc), len(feats), _p(ur), b, C.byref(self.h)))

    def __len__(self):
        return int(lib().sslam_frame_count(self.h))

    def search_by_projection(self, mode, queries, qdesc, occupied=None, nnratio=0.8, th_dist=100, check_orientation=True):
        queries = np.ascontiguousarray(queries, PQ_DTYPE); qdesc = np.ascontiguousarray(qdesc, np.uint8)
        n = len(self)
        assigned = np.full(n, -1, np.int32); nm = C.c_int(0)
        occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
        _chk(lib().sslam_search_by_projection_frame(self.ctx.h, self.h, int(mode), _p(occ), _p(queries), _p(qdesc), len(queries),
                                                    C.c_float(nnratio), int(th_dist), int(bool(check_orientation)), _p(assigned), C.byref(nm)))
        return assigned, nm.value

    def fuse_search(self, queries, qdesc, chi2_mode=0, inv_level_sigma2=None):
        """candidate search of ORBmatcher::Fuse / LSDmatcher::Fuse on this (key)frame: (best_idx, best_dist) per query"""
        queries = np.ascontiguousarray(queries, PQ_DTYPE); qdesc = np.ascontiguousarray(qdesc, np.uint8)
        sg = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
        bi = np.zeros(len(queries), np.int32); bd = np.zeros(len(queries), np.int32)
        _chk(lib().sslam_fuse_search(self.ctx.h, self.h, int(chi2_mode), _p(sg), 0 if sg is None else len(sg), _p(queries), _p(qdesc), len(queries), _p(bi), _p(bd)))
        return bi, bd

    def search_for_triangulation(self, kf2, free1, free2, ptr1, ptr2, idx1, idx2, F12, ex, ey, scale_factors2, level_sigma2_2,
                                 only_stereo=False, check_orientation=True):
        """ORBmatcher::SearchForTriangulation(this, kf2, F12, ...) -> (vMatches12, nmatches)"""
        f1 = np.ascontiguousarray(free1, np.uint8); f2 = np.ascontiguousarray(free2, np.uint8)
        ptr1 = np.ascontiguousarray(ptr1, np.int32); ptr2 = np.ascontiguousarray(ptr2, np.int32)
        idx1 = np.ascontiguousarray(idx1, np.int32); idx2 = np.ascontiguousarray(idx2, np.int32)
        F = (C.c_float * 9)(*np.asarray(F12, np.float32).reshape(9).tolist())
        sf = np.ascontiguousarray(scale_factors2, np.float32); sg = np.ascontiguousarray(level_sigma2_2, np.float32)
        m12 = np.full(len(self), -2, np.int32); nm = C.c_int(0)
        _chk(lib().sslam_orb_search_for_triangulation(self.ctx.h, self.h, kf2.h, _p(f1), _p(f2), _p(ptr1), _p(ptr2), len(ptr1) - 1, _p(idx1), _p(idx2),
                                                      F, C.c_float(ex), C.c_float(ey), _p(sf), _p(sg), len(sf), int(bool(only_stereo)),
                                                      int(bool(check_orientation)), _p(m12), C.byref(nm)))
        return m12, nm.value

    def knn2(self, train):
        n = len(self)
        idx = np.full((n, 2), -1, np.int32); dist = np.full((n, 2), -1, np.int32)
        _chk(lib().sslam_hamming_knn2_frames(self.ctx.h, self.h, train.h, _p(idx), _p(dist)))
        return idx, dist

    def close(self):
        if self.h:
            lib().sslam_frame_destroy(self.h); self.h = C.c_void_p()


def frontend_batch_alloc(n, cap, lcap, pinned=False):
    """result arrays of sslam_frontend_batch for n frames (pinned=True: in pinned memory, so that the library copies straight into them);
    pageable arrays are touched once so that a timed call does not pay their first page faults"""
    def alloc(shape, dt):
        if not pinned:
            a = np.empty(shape, dt); a.view(np.uint8).reshape(-1)[::4096] = 0
            return a
        import torch
        nbytes = int(np.prod(shape)) * np.dtype(dt).itemsize
        return torch.empty(max(nbytes, 1), dtype=torch.uint8, pin_memory=True).numpy()[:nbytes].view(dt).reshape(shape)
    return (alloc((n, cap), KP_DTYPE), alloc((n, cap, 32), np.uint8), alloc((n,), np.int32),
            alloc((n, lcap), KL_DTYPE), alloc((n, lcap, 32), np.uint8), alloc((n, lcap, 3), np.float64), alloc((n,), np.int32))


def frontend_batch_raw(orb, lines, images, out, chunk=0, lcap=None):
    """sslam_frontend_batch into arrays from frontend_batch_alloc: the bare library call (what the PCIe-inclusive measurements time)"""
    n, h, w = images.shape
    kp, desc, nk, kl, ld, fn, nl = out
    lcap = int(lcap if lcap is not None else kl.shape[1])
    _chk(lib().sslam_frontend_batch(orb.h, lines.h if lines is not None else None, _p(images), n, w, h, C.c_size_t(w), C.c_size_t(w * h), int(chunk),
                                    _p(kp), _p(desc), _p(nk), orb.cap, _p(kl), _p(ld), _p(fn), _p(nl), lcap))
    return out


class BatchMatch(C.Structure):
    """sslam_batch_match (include/sslam_frontend.h)"""
    _fields_ = [("window_size", C.c_int32), ("nnratio", C.c_float), ("check_orientation", C.c_int32), ("bounds", C.c_float * 4),
                ("line_gate_scale", C.c_double), ("line_ratio_mode", C.c_int32),
                ("init_matches12", C.c_void_p), ("init_nmatches", C.c_void_p), ("knn_idx", C.c_void_p), ("knn_dist", C.c_void_p),
                ("line_pairs", C.c_void_p), ("line_npairs", C.c_void_p)]


def frontend_batch_match_alloc(n, cap, lcap, pinned=False, knn=True):
    """the match-stage arrays of sslam_frontend_batch_match: (init_matches12, init_nmatches, knn_idx, knn_dist, line_pairs, line_npairs)"""
    def alloc(shape, dt):
        if not pinned:
            a = np.empty(shape, dt); a.view(np.uint8).reshape(-1)[::4096] = 0
            return a
        import torch
        nbytes = int(np.prod(shape)) * np.dtype(dt).itemsize
        return torch.empty(max(nbytes, 1), dtype=torch.uint8, pin_memory=True).numpy()[:nbytes].view(dt).reshape(shape)
    return (alloc((n, cap), np.int32), alloc((n,), np.int32), alloc((n, cap, 2), np.int32) if knn else None, alloc((n, cap, 2), np.int32) if knn else None,
            alloc((n, lcap, 2), np.int32), alloc((n,), np.int32))


def frontend_batch_match_raw(orb, lines, images, out, mout, chunk=0, window=100, nnratio=0.9, check_orientation=True, bounds=None, line_gate_scale=0.5, line_ratio_mode=False):
    """sslam_frontend_batch_match into arrays from frontend_batch_alloc / frontend_batch_match_alloc: frame i is matched against frame i-1"""
    n, h, w = images.shape
    kp, desc, nk, kl, ld, fn, nl = out
    m12, nm, ki, kd, lp, nlp = mout
    M = BatchMatch()
    M.window_size = int(window); M.nnratio = float(nnratio); M.check_orientation = int(bool(check_orientation))
    M.bounds = (C.c_float * 4)(*(bounds if bounds is not None else (0.0, float(w), 0.0, float(h))))
    M.line_gate_scale = float(line_gate_scale); M.line_ratio_mode = int(bool(line_ratio_mode))
    vp = lambda a: a.ctypes.data if a is not None else None
    M.init_matches12 = vp(m12); M.init_nmatches = vp(nm); M.knn_idx = vp(ki); M.knn_dist = vp(kd); M.line_pairs = vp(lp); M.line_npairs = vp(nlp)
    _chk(lib().sslam_frontend_batch_match(orb.h, lines.h if lines is not None else None, _p(images), n, w, h, C.c_size_t(w), C.c_size_t(w * h), int(chunk),
                                          _p(kp), _p(desc), _p(nk), orb.cap, _p(kl), _p(ld), _p(fn), _p(nl), int(kl.shape[1]), C.byref(M)))
    return out, mout


def frontend_batch(orb, lines, images, chunk=0, max_lines=None, pinned=False):
    """sslam_frontend_batch: images = uint8 array [n, h, w] in HOST memory -> per-frame (keypoints, descriptors, keylines, line descriptors,
    line functions) lists; `lines` may be None (ORB only).  pinned=True allocates the result arrays in pinned memory (torch), so that the
    library copies straight into them; pass a pinned `images` array for the same on the way in."""
    images = np.ascontiguousarray(images, np.uint8)
    n, h, w = images.shape
    cap = orb.cap
    lcap = int(max_lines if max_lines is not None else (lines.max_lines if lines is not None else 1))

    def alloc(shape, dt):
        if not pinned:
            return np.zeros(shape, dt)
        import torch
        nbytes = int(np.prod(shape)) * np.dtype(dt).itemsize
        return torch.empty(max(nbytes, 1), dtype=torch.uint8, pin_memory=True).numpy()[:nbytes].view(dt).reshape(shape)
    kp = alloc((n, cap), KP_DTYPE); desc = alloc((n, cap, 32), np.uint8); nk = alloc((n,), np.int32)
    kl = alloc((n, lcap), KL_DTYPE); ld = alloc((n, lcap, 32), np.uint8); fn = alloc((n, lcap, 3), np.float64); nl = alloc((n,), np.int32)
    _chk(lib().sslam_frontend_batch(orb.h, lines.h if lines is not None else None, _p(images), n, w, h, C.c_size_t(w), C.c_size_t(w * h), int(chunk),
                                    _p(kp), _p(desc), _p(nk), cap, _p(kl), _p(ld), _p(fn), _p(nl), lcap))
    if lines is None:
        nl = np.zeros(n, np.int32)
    return [(kp[i, :nk[i]], desc[i, :nk[i]], kl[i, :nl[i]], ld[i, :nl[i]], fn[i, :nl[i]]) for i in range(n)]


class OrbExtractor:
    """Harness-side mirror of StructureSLAM::ORBextractor (include/ORBextractor.h:45-111)."""

    def __init__(self, ctx, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.ctx = ctx
        self.nlevels = nlevels
        self.h = C.c_void_p()
        _chk(lib().sslam_orb_create(ctx.h, int(nfeatures), C.c_float(scale_factor), int(nlevels), int(ini_th), int(min_th), C.byref(self.h)))
        self.cap = lib().sslam_orb_max_keypoints(self.h)

    def set_blur_variant(self, variant):
        """0: the bit-exact 8-bit GaussianBlur of OpenCV >= 3.4.1 (default, decision D6); 1: OpenCV 3.4.0's rounded taps (sslam_orb_set_blur_variant)"""
        _chk(lib().sslam_orb_set_blur_variant(self.h, int(variant)))

    def set_gate_event(self, hip_event):
        """hipEvent_t handle (int) every following batch call waits for between its pyramid kernels and the rest; 0 / None clears it (sslam_orb_set_gate_event)"""
        _chk(lib().sslam_orb_set_gate_event(self.h, C.c_void_p(int(hip_event or 0))))

    def scales(self):
        n = self.nlevels
        s = np.zeros(n, np.float32); i = np.zeros(n, np.float32); g = np.zeros(n, np.float32); ig = np.zeros(n, np.float32)
        f = np.zeros(n, np.int32)
        _chk(lib().sslam_orb_get_scales(self.h, _p(s), _p(i), _p(g), _p(ig), _p(f)))
        return s, i, g, ig, f

    def __call__(self, gray):
        """operator()(image, mask, keypoints, descriptors) on a host image."""
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape if gray.ndim == 2 else (0, 0)
        kp = np.zeros(self.cap, KP_DTYPE); desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int(0)
        _chk(lib().sslam_orb_extract(self.h, _p(gray), w, h, C.c_size_t(gray.strides[0] if gray.ndim == 2 and h else 0),
                                     _p(kp), _p(desc), self.cap, C.byref(n)))
        return kp[:n.value].copy(), desc[:n.value].copy()

    def extract_batch_dev(self, d_images, w, h, pitch, image_stride, nframes, d_kp, d_desc, d_counts, cap, stream=None):
        _chk(lib().sslam_orb_extract_batch_dev(self.h, _p(d_images), int(w), int(h), C.c_size_t(pitch), C.c_size_t(image_stride),
                                               int(nframes), _p(d_kp), _p(d_desc), _p(d_counts), int(cap), C.c_void_p(stream or 0)))

    def debug_level(self, frame, level):
        w = C.c_int(0); h = C.c_int(0)
        _chk(lib().sslam_orb_debug_level(self.h, frame, level, C.c_void_p(0), C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        _chk(lib().sslam_orb_debug_level(self.h, frame, level, _p(out), C.byref(w), C.byref(h)))
        return out

    def debug_blur_patches(self, frame=0):
        """a7 stage tap: (keypoints, blurred 37x37 windows around them) of frame `frame` of the last call, in output order."""
        kp = np.zeros(self.cap, KP_DTYPE); pat = np.zeros((self.cap, 37, 37), np.uint8); n = C.c_int(0)
        _chk(lib().sslam_orb_debug_blur_patches(self.h, frame, _p(kp), _p(pat), self.cap, C.byref(n)))
        return kp[:n.value].copy(), pat[:n.value].copy()

    def debug_candidates(self, frame, level, cap=400000):
        out = np.zeros((cap, 3), np.int32); n = C.c_int(0)
        _chk(lib().sslam_orb_debug_candidates(self.h, frame, level, _p(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def close(self):
        if self.h:
            lib().sslam_orb_destroy(self.h)
            self.h = C.c_void_p()


class LineExtractor:
    """Harness-side mirror of LineSegment::ExtractLineSegment (src/ExtractLineSegment.cpp:18-69)."""

    def __init__(self, ctx, max_lines=40):
        self.ctx = ctx
        self.max_lines = max_lines
        self.h = C.c_void_p()
        _chk(lib().sslam_lines_create(ctx.h, int(max_lines), C.byref(self.h)))

    def __call__(self, gray, cap=None):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        cap = cap or 8192
        kl = np.zeros(cap, KL_DTYPE); ld = np.zeros((cap, 32), np.uint8); fn = np.zeros((cap, 3), np.float64)
        n = C.c_int(0)
        _chk(lib().sslam_lines_extract(self.h, _p(gray), w, h, C.c_size_t(gray.strides[0]), _p(kl), _p(ld), _p(fn), cap, C.byref(n)))
        return kl[:n.value].copy(), ld[:n.value].copy(), fn[:n.value].copy()

    def extract_batch_dev(self, d_images, w, h, pitch, image_stride, nframes, d_kl, d_ldesc, d_linefn, d_counts, cap, stream=None):
        _chk(lib().sslam_lines_extract_batch_dev(self.h, _p(d_images), int(w), int(h), C.c_size_t(pitch), C.c_size_t(image_stride),
                                                 int(nframes), _p(d_kl), _p(d_ldesc), _p(d_linefn), _p(d_counts), int(cap),
                                                 C.c_void_p(stream or 0)))

    def set_blur_variant(self, variant):
        """0: OpenCV >= 3.4.1's bit-exact 8-bit GaussianBlur (default, decision D6); 1: OpenCV 3.4.0's rounded taps (sslam_lines_set_blur_variant)"""
        _chk(lib().sslam_lines_set_blur_variant(self.h, int(variant)))

    def set_nfa_variant(self, variant):
        """decision D11: 1 = nfa()'s first term is (double(n) + 1) (default since round 5, OpenCV as recalled); 0 = log_gamma(n + 1) (sslam_lines_set_nfa_variant)"""
        _chk(lib().sslam_lines_set_nfa_variant(self.h, int(variant)))

    def set_lbd_bit_order(self, variant):
        """decision D12: 1 = comparison i -> 0x80 >> i (default since round 5, OpenCV as recalled); 0 = comparison i -> bit i (sslam_lines_set_lbd_bit_order)"""
        _chk(lib().sslam_lines_set_lbd_bit_order(self.h, int(variant)))

    def set_resize_variant(self, variant):
        """decision D7: 0 = INTER_LINEAR_EXACT for LSD's 0.8x rescale (default); 1 = INTER_LINEAR (sslam_lines_set_resize_variant)"""
        _chk(lib().sslam_lines_set_resize_variant(self.h, int(variant)))

    def set_seed_order(self, variant):
        """decision D2: 0 = raster order inside a gradient bin (default, on the device); 1 = the host's std::sort (sslam_lines_set_seed_order)"""
        _chk(lib().sslam_lines_set_seed_order(self.h, int(variant)))

    def set_core_event(self, hip_event):
        """hipEvent_t handle (int) recorded right before the sequential LSD core of every following batch call; 0 / None clears it"""
        _chk(lib().sslam_lines_set_core_event(self.h, C.c_void_p(int(hip_event or 0))))

    def set_core_gate(self, wait_event, done_event):
        """sslam_lines_set_core_gate: hipEvent_t handles (int; 0 / None: none) waited for right before / recorded right behind the sequential core"""
        _chk(lib().sslam_lines_set_core_gate(self.h, C.c_void_p(int(wait_event or 0)), C.c_void_p(int(done_event or 0))))

    def debug_segments(self, frame, cap=20000):
        out = np.zeros((cap, 4), np.float32); n = C.c_int(0)
        _chk(lib().sslam_lines_debug_segments(self.h, frame, _p(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def close(self):
        if self.h:
            lib().sslam_lines_destroy(self.h)
            self.h = C.c_void_p()


# ---- multi-GPU batch mode (include/sslam_frontend.h: sslam_group_*, record stream) ------------------------------------------------
class FrontendParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32), ("ini_th_fast", C.c_int32),
                ("min_th_fast", C.c_int32), ("max_lines", C.c_int32)]


def record_bytes(nkp, nl):
    return (16 + nkp * 60 + nl * 124 + 15) & ~15


def record_stream_capacity(nframes, cap, lcap):
    lib().sslam_record_stream_capacity.restype = C.c_uint64
    return int(lib().sslam_record_stream_capacity(int(nframes), int(cap), int(lcap)))


def pack_records_dev(ctx, nframes, frame0, frame_step, d_kp, d_desc, d_nkp, cap, d_kl, d_ldesc, d_linefn, d_nl, lcap, d_out, out_capacity, d_total, stream=None):
    """device tensors / pointers in, record stream in d_out, length in the device word d_total (enqueued, not synchronised)"""
    _chk(lib().sslam_pack_records_dev(ctx.h, int(nframes), int(frame0), int(frame_step), _p(d_kp), _p(d_desc), _p(d_nkp), int(cap),
                                      _p(d_kl), _p(d_ldesc), _p(d_linefn), _p(d_nl), int(lcap), _p(d_out), C.c_uint64(out_capacity), _p(d_total),
                                      C.c_void_p(stream or 0)))


def unpack_records(stream, nframes, cap, lcap, with_lines=True):
    """host record stream (uint8 array) -> (kp[n,cap], desc, nkp, kl, ldesc, linefn, nl, nrecords) as sslam_frontend_batch lays them out"""
    stream = np.ascontiguousarray(stream, np.uint8)
    kp = np.zeros((nframes, cap), KP_DTYPE); desc = np.zeros((nframes, cap, 32), np.uint8); nk = np.full(nframes, -1, np.int32)
    kl = np.zeros((nframes, lcap), KL_DTYPE); ld = np.zeros((nframes, lcap, 32), np.uint8); fn = np.zeros((nframes, lcap, 3), np.float64); nl = np.full(nframes, -1, np.int32)
    nrec = C.c_int(0)
    _chk(lib().sslam_unpack_records(_p(stream), C.c_uint64(stream.size), int(nframes), _p(kp), _p(desc), _p(nk), int(cap),
                                    _p(kl) if with_lines else None, _p(ld), _p(fn), _p(nl), int(lcap), C.byref(nrec)))
    return kp, desc, nk, kl, ld, fn, nl, nrec.value


class Group:
    """sslam_group: Group(ngpu=G) drives G devices from this process; Group(device=, rank=, nranks=, uid=) is one rank of a
    one-process-per-GPU job (uid from Group.unique_id() on rank 0, broadcast by the launcher)."""

    def __init__(self, ngpu=None, device=0, rank=0, nranks=1, uid=None):
        self.h = C.c_void_p()
        if ngpu is not None:
            _chk(lib().sslam_group_create(int(ngpu), C.byref(self.h)))
        else:
            uid = np.ascontiguousarray(uid, np.uint8)
            assert uid.size == 128
            _chk(lib().sslam_group_create_rank(int(device), int(rank), int(nranks), _p(uid), C.byref(self.h)))
        self.size = lib().sslam_group_size(self.h); self.rank = lib().sslam_group_rank(self.h)

    @staticmethod
    def unique_id():
        uid = np.zeros(128, np.uint8)
        _chk(lib().sslam_group_unique_id(_p(uid)))
        return uid

    def gather_dev(self, d_send, d_send_bytes, d_recv=None, recv_capacity=0, stream=None):
        """the exchange step (blocking); returns the per-rank stream lengths on rank 0, None elsewhere"""
        sizes = np.zeros(self.size, np.uint64)
        _chk(lib().sslam_group_gather_dev(self.h, _p(d_send), _p(d_send_bytes), _p(d_recv), C.c_uint64(recv_capacity), _p(sizes), C.c_void_p(stream or 0)))
        return sizes if self.rank == 0 else None

    def frontend_batch_sharded(self, images, nfeatures=1000, max_lines=200, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, cap=None, lcap=None):
        images = np.ascontiguousarray(images, np.uint8)
        n, h, w = images.shape
        prm = FrontendParams(int(nfeatures), float(scale_factor), int(nlevels), int(ini_th), int(min_th), int(max_lines))
        if cap is None:
            cap = int(nfeatures) + 64 * int(nlevels)
        lcap = int(lcap if lcap is not None else max(max_lines, 1))
        kp = np.zeros((n, cap), KP_DTYPE); desc = np.zeros((n, cap, 32), np.uint8); nk = np.zeros(n, np.int32)
        kl = np.zeros((n, lcap), KL_DTYPE); ld = np.zeros((n, lcap, 32), np.uint8); fn = np.zeros((n, lcap, 3), np.float64); nl = np.zeros(n, np.int32)
        _chk(lib().sslam_frontend_batch_sharded(self.h, C.byref(prm), _p(images), n, w, h, C.c_size_t(w), C.c_size_t(w * h), _p(kp), _p(desc), _p(nk), cap,
                                                _p(kl), _p(ld), _p(fn), _p(nl), lcap))
        return [(kp[i, :nk[i]], desc[i, :nk[i]], kl[i, :nl[i]], ld[i, :nl[i]], fn[i, :nl[i]]) for i in range(n)]

    def close(self):
        if self.h:
            lib().sslam_group_destroy(self.h)
            self.h = C.c_void_p()
