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


class Oracle:
    def __init__(self):
        build()
        self.L = C.CDLL(OLIB)
        self.L.orc_fast_atan2.restype = C.c_float
        self.L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]

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
