"""Batch-of-frames front-end pipeline over the C ABI (harness side, used by bench.py and
the batch/multi-GPU tests).  One instance per GPU/process.  All device memory is held in
torch tensors (plumbing); every computation is a libsslam_frontend.so kernel launched on
torch's current stream through the *_dev entry points.

Workload = BASELINE config 3: per frame ORB extract (1000 kp) + LSD/LBD extract (<=200
lines) + match against the previous frame's features: ORBmatcher::SearchForInitialization
(window 100, level 0) + dense knn-2 over all ORB descriptors, and the LSD knn-2 + MAD gate.
"""
import ctypes as C
import numpy as np
import torch


class FrontendBatch:
    def __init__(self, fe, ctx, w, h, batch, nfeatures=1000, max_lines=200, device="cuda:0", with_lines=True, with_match=True):
        self.fe, self.ctx = fe, ctx
        self.w, self.h, self.B = w, h, batch
        self.dev = torch.device(device)
        self.with_lines, self.with_match = with_lines, with_match
        self.orb = fe.OrbExtractor(ctx, nfeatures, 1.2, 8, 20, 7)
        self.lines = fe.LineExtractor(ctx, max_lines) if with_lines else None
        self.cap, self.lcap = self.orb.cap, max_lines
        B, cap, lcap = batch, self.cap, self.lcap
        z = lambda *s, dt=torch.uint8: torch.zeros(*s, dtype=dt, device=self.dev)
        self.feat = {}
        for tag in ("cur", "prev"):
            self.feat[tag] = dict(kp=z(B, cap, 7, dt=torch.float32), desc=z(B, cap, 32), n=z(B, dt=torch.int32),
                                  kl=z(B, lcap, 17, dt=torch.float32), ldesc=z(B, lcap, 32),
                                  linefn=z(B, lcap, 3, dt=torch.float64), nl=z(B, dt=torch.int32))
        self.pm = z(B, cap, 2, dt=torch.float32)
        self.m12 = z(B, cap, dt=torch.int32)
        self.nmatch = z(B, dt=torch.int32)
        self.knn_idx = z(B, cap, 2, dt=torch.int32)
        self.knn_dist = z(B, cap, 2, dt=torch.int32)
        self.lpairs = z(B, lcap, 2, dt=torch.int32)
        self.nlpairs = z(B, dt=torch.int32)
        self.bounds = (C.c_float * 4)(0.0, float(w), 0.0, float(h))

    def _stream(self):
        h = torch.cuda.current_stream(self.dev).cuda_stream
        assert h != 0, "pipeline work must run on an explicit (non-default) stream: handle 0 means 'context stream' in the C ABI"
        return h

    def _streams(self):
        if not hasattr(self, "_s1"):
            import os
            pr = int(os.environ.get("SSLAM_LINE_STREAM_PRIORITY", "0"))      # experiment knob: -1 = the line branch (critical path) on a high-priority stream
            self._s1, self._s2 = torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev, priority=pr)
        return self._s1, self._s2

    def extract(self, images, tag="cur", lines_first=False):
        """images: uint8 device tensor [B, h, w] (contiguous).  Runs on the pipeline's point stream; the
        caller's current stream is ordered before and after."""
        assert images.is_cuda and images.dtype == torch.uint8 and images.shape == (self.B, self.h, self.w) and images.is_contiguous()
        f = self.feat[tag]
        s1, _ = self._streams()
        cur = torch.cuda.current_stream(self.dev)
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            st = self._stream()

            def _orb():
                self.orb.extract_batch_dev(images, self.w, self.h, self.w, self.w * self.h, self.B, f["kp"], f["desc"], f["n"], self.cap, st)

            def _lines():
                if self.with_lines:
                    self.lines.extract_batch_dev(images, self.w, self.h, self.w, self.w * self.h, self.B, f["kl"], f["ldesc"], f["linefn"],
                                                 f["nl"], self.lcap, st)
            for fn in ((_lines, _orb) if lines_first else (_orb, _lines)):
                fn()
        cur.wait_stream(s1)

    def match(self):
        """cur (F2 / train) against prev (F1 / query)."""
        s1, _ = self._streams()
        cur = torch.cuda.current_stream(self.dev)
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            self._match_points()
            if self.with_lines:
                self._match_lines()
        cur.wait_stream(s1)

    def _match_points(self):
        L = self.fe.lib()
        p, c = self.feat["prev"], self.feat["cur"]
        st = C.c_void_p(self._stream())
        _p = lambda t: C.c_void_p(t.data_ptr())
        self.pm.copy_(p["kp"][:, :, :2])          # vbPrevMatched starts at F1's keypoint positions (Tracking.cc:340-342)
        rc = L.sslam_orb_search_for_initialization_batch_dev(self.ctx.h, _p(p["kp"]), _p(p["desc"]), _p(p["n"]), _p(c["kp"]), _p(c["desc"]),
                                                             _p(c["n"]), self.cap, self.B, _p(self.pm), _p(self.m12), _p(self.nmatch), 100,
                                                             C.c_float(0.9), 1, self.bounds, st)
        assert rc == 0, L.sslam_last_error()
        rc = L.sslam_hamming_knn2_batch_dev(self.ctx.h, _p(p["desc"]), _p(p["n"]), _p(c["desc"]), _p(c["n"]), self.cap, self.B,
                                            _p(self.knn_idx), _p(self.knn_dist), st)
        assert rc == 0, L.sslam_last_error()

    def _match_lines(self):
        L = self.fe.lib()
        p, c = self.feat["prev"], self.feat["cur"]
        st = C.c_void_p(self._stream())
        _p = lambda t: C.c_void_p(t.data_ptr())
        rc = L.sslam_line_match_batch_dev(self.ctx.h, _p(p["ldesc"]), _p(p["nl"]), _p(c["ldesc"]), _p(c["nl"]), self.lcap, self.B,
                                          C.c_double(0.5), 0, _p(self.lpairs), _p(self.nlpairs), st)
        assert rc == 0, L.sslam_last_error()

    def step(self, images, overlap=False, lines_first=False, join=True):
        """One pass of the hot path.  overlap=True runs the point branch (ORB extract + ORB matching)
        and the line branch (LSD/LBD extract + line matching) on two HIP streams: the line branch is
        latency-bound (one persistent wave per frame), the point branch fills the idle issue slots."""
        if not (overlap and self.with_lines):
            self.extract(images, "cur", lines_first=lines_first)
            if self.with_match:
                self.match()
            return
        self._streams()
        if not hasattr(self, "_core_event"):
            # The kernels in front of the sequential core (blur, gradient, counting sort: ~22 ms of 12 288 frames) are bandwidth-bound; the core itself is latency-bound
            # and leaves issue slots free.  The library records this event right before the core.  Default (round 6, where the library takes the guest form): the pyramid is built at once, beside the
            # line prologue, and FAST .. matching wait for the event (sslam_orb_set_gate_event) -- announced this way, the library launches the core in its guest form
            # (a third of the registers free: csrc/lsd_regions.h), so FAST starts with the core instead of behind its first 6 144 waves (profiles/r06a_timeline_*).
            # SSLAM_POINTS_AT_CORE=1: the whole point branch waits for the event (rounds 3-5); =0: both branches start together.
            import os
            self._core_event = None
            mode = os.environ.get("SSLAM_POINTS_AT_CORE", "auto")
            if mode != "0":
                ev = torch.cuda.Event(); ev.record(self._s2)          # (recording creates the hipEvent_t)
                self._core_event = ev
                self.lines.set_core_event(ev.cuda_event)
                if mode == "auto":      # the pyramid goes ahead of the event where the core runs in its guest form (batches above 16 workgroups per CU); smaller batches: measured better with the whole branch behind it
                    mode = "pyr" if self.fe.lib().sslam_lines_core_guest_form(self.lines.h, self.B) else "1"
                self._gate_in_orb = mode == "pyr"
                if self._gate_in_orb:
                    self.orb.set_gate_event(ev.cuda_event)
            else:
                self._gate_in_orb = False
        cur = torch.cuda.current_stream(self.dev)
        self._s1.wait_stream(cur); self._s2.wait_stream(cur)
        f = self.feat["cur"]
        with torch.cuda.stream(self._s2):
            self.lines.extract_batch_dev(images, self.w, self.h, self.w, self.w * self.h, self.B, f["kl"], f["ldesc"], f["linefn"],
                                         f["nl"], self.lcap, self._stream())
            if self.with_match:
                self._match_lines()
        with torch.cuda.stream(self._s1):
            if self._core_event is not None and not self._gate_in_orb:      # the point branch starts when the sequential LSD core does (sslam_lines_set_core_event)
                self._s1.wait_event(self._core_event)
            self.orb.extract_batch_dev(images, self.w, self.h, self.w, self.w * self.h, self.B, f["kp"], f["desc"], f["n"], self.cap, self._stream())
            if self.with_match:
                self._match_points()
        if join:      # join=False: the two branches may drift apart across steps (the point branch of step k+1 under the line branch of step k);
            cur.wait_stream(self._s1); cur.wait_stream(self._s2)      # the caller synchronises both streams (or the device) before it reads results

    def packed_stream(self):
        """The step's results as the C ABI's record stream (sslam_pack_records_dev): device uint8 tensor trimmed to its length (synchronises)."""
        c = self.feat["cur"]
        capb = self.fe.record_stream_capacity(self.B, self.cap, self.lcap if self.with_lines else 0)
        out = torch.zeros(capb + 16, dtype=torch.uint8, device=self.dev); tot = torch.zeros(2, dtype=torch.int64, device=self.dev)
        s1, _ = self._streams()
        s1.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s1):
            ln = self.with_lines
            self.fe.pack_records_dev(self.ctx, self.B, 0, 1, c["kp"], c["desc"], c["n"], self.cap, c["kl"] if ln else None, c["ldesc"] if ln else None,
                                     c["linefn"] if ln else None, c["nl"] if ln else None, self.lcap, out, capb, tot, self._stream())
        s1.synchronize()
        return out[:int(tot[0].item())]

    def close(self):
        self.orb.close()
        if self.lines:
            self.lines.close()


def profile_drain(fe, ctx, cap=64):
    L = fe.lib()
    names = (C.c_char_p * cap)(); ms = (C.c_double * cap)(); cnt = (C.c_int * cap)()
    n = L.sslam_profile_drain(ctx.h, names, ms, cnt, cap)
    assert n >= 0
    return {names[i].decode(): (ms[i], cnt[i]) for i in range(min(n, cap))}
