"""GPU: the batch-of-frames device API equals per-frame extraction, frame by frame, and full-size
size-independent properties of the hot path (BASELINE configs at their full sizes)."""
import os
import numpy as np
import pytest
import torch
import pkg
from synth import synth_frame, warp_prev, noise_frame, const_frame

pytestmark = pytest.mark.gpu


def test_batch_equals_single_and_oracle(fe, ctx, oracle):
    pipeline = pkg._load("sslam_pipeline", os.path.join(pkg.PKG_DIR, "pipeline.py"))
    frames = [synth_frame(2000), noise_frame(9), const_frame(), synth_frame(2001), warp_prev(synth_frame(2000))]
    B = len(frames)
    pipe = pipeline.FrontendBatch(fe, ctx, 640, 480, B, 1000, 200, "cuda:0")
    imgs = torch.from_numpy(np.stack(frames)).cuda()
    prev = torch.from_numpy(np.stack([warp_prev(f) for f in frames])).cuda()
    pipe.extract(prev, "prev")
    pipe.step(imgs, overlap=True)
    torch.cuda.synchronize()
    rkp, rdesc, rn, rkl, rld, rfn, rnl, nrec = fe.unpack_records(pipe.packed_stream().cpu().numpy(), B, pipe.cap, pipe.lcap)      # the record stream of the gather
    assert nrec == B
    ox = fe.OrbExtractor(ctx, 1000); lx = fe.LineExtractor(ctx, 200)
    for i, f in enumerate(frames):
        kp, desc = ox(f); kl, ld, fn = lx(f)
        assert rn[i] == len(kp) and rnl[i] == len(kl)
        np.testing.assert_array_equal(rkp[i, :rn[i]].view(np.uint8).reshape(-1, 28), kp.view(np.uint8).reshape(len(kp), 28))
        np.testing.assert_array_equal(rdesc[i, :rn[i]], desc)
        np.testing.assert_array_equal(rkl[i, :rnl[i]].view(np.uint8).reshape(-1, 68), kl.view(np.uint8).reshape(len(kl), 68))
        np.testing.assert_array_equal(rld[i, :rnl[i]], ld)
        np.testing.assert_array_equal(rfn[i, :rnl[i]], fn)
    # matching of the batch == host API == oracle (frame 0 and 3)
    m12 = pipe.m12.cpu().numpy(); nm = pipe.nmatch.cpu().numpy()
    kidx = pipe.knn_idx.cpu().numpy(); lp = pipe.lpairs.cpu().numpy(); nlp = pipe.nlpairs.cpu().numpy()
    for i in (0, 3):
        kp1, d1 = oracle.orb_extract(warp_prev(frames[i]), 1000); kp2, d2 = oracle.orb_extract(frames[i], 1000)
        pm = np.stack([kp1["x"], kp1["y"]], axis=1).astype(np.float32)
        om12, _, on = oracle.search_for_initialization(kp1, d1, kp2, d2, pm, 100, 0.9, True)
        assert nm[i] == on
        np.testing.assert_array_equal(m12[i][:len(kp1)], om12)
        oi, od = oracle.knn2(d1, d2)
        np.testing.assert_array_equal(kidx[i][:len(kp1)], oi)
        l1 = oracle.lines_extract(warp_prev(frames[i]), 200); l2 = oracle.lines_extract(frames[i], 200)
        opairs, _, _ = oracle.line_match(l1[1], l2[1], 0.5, False)
        assert nlp[i] == len(opairs)
        np.testing.assert_array_equal(lp[i][:nlp[i]], opairs)
    assert nm[2] == 0 and rn[2] == 0 and rnl[2] == 0          # constant frame: nothing, no error
    ox.close(); lx.close(); pipe.close()


def test_fullsize_properties_1280(fe, ctx):
    """BASELINE configs[3] size: properties that do not need the (slow) oracle."""
    img = synth_frame(1235, w=1280, h=960)
    ox = fe.OrbExtractor(ctx, 2000); lx = fe.LineExtractor(ctx, 400)
    kp, desc = ox(img); kp2, desc2 = ox(img)
    np.testing.assert_array_equal(kp.view(np.uint8), kp2.view(np.uint8))       # idempotent / deterministic
    np.testing.assert_array_equal(desc, desc2)
    assert 1900 <= len(kp) <= 2016 and (np.diff(kp["octave"]) >= 0).all()
    per = np.bincount(kp["octave"], minlength=8)
    assert (per <= np.array([434, 362, 302, 251, 209, 175, 145, 122]) + 2).all()   # quota + quadtree overshoot (D.2)
    D = ctx.hamming_matrix(desc, desc)
    assert (np.diag(D) == 0).all() and (D == D.T).all()                           # Hamming metric properties
    idx, dist = ctx.hamming_knn2(desc, desc)
    assert (dist[:, 0] == 0).all() and (dist[:, 1] >= 0).all()
    assert (D[np.arange(len(D)), idx[:, 0]] == 0).all()
    srt = np.sort(D, axis=1)
    assert (srt[:, 0] == dist[:, 0]).all() and (srt[:, 1] == dist[:, 1]).all()    # knn2 distances == two smallest of the dense matrix
    kl, ld, fn = lx(img)
    assert len(kl) == 400 and (np.diff(kl["response"]) <= 0).all()
    assert np.allclose(np.hypot(fn[:, 0], fn[:, 1]), 1.0)
    ends = np.stack([kl["startPointX"], kl["startPointY"], np.ones(len(kl))], 1)
    assert np.abs((fn * ends).sum(1)).max() < 1e-6                                 # endpoints lie on their line
    ox.close(); lx.close()


def test_empty_and_tiny_inputs(fe, ctx):
    ox = fe.OrbExtractor(ctx, 1000)
    kp, desc = ox(np.zeros((0, 0), np.uint8))
    assert len(kp) == 0
    kp, desc = ox(synth_frame(5, w=64, h=48))          # most levels smaller than a FAST cell
    assert desc.shape == (len(kp), 32)
    ox.close()
    assert ctx.hamming_knn2(np.zeros((0, 32), np.uint8), np.zeros((4, 32), np.uint8))[0].shape == (0, 2)


def test_batch_of_19_distinct_frames(fe, ctx, oracle):
    """two full groups of eight plus a tail: the XCD-aware frame mapping (xcd_mix_frame) permutes which workgroup handles which
    frame; every slot must still hold ITS frame's results"""
    pipeline = pkg._load("sslam_pipeline", os.path.join(pkg.PKG_DIR, "pipeline.py"))
    w, h, B = 320, 240, 19
    frames = [synth_frame(4000 + i, w, h, nshapes=10 + 7 * i, nstrokes=3 * i, noise=float(i % 4)) for i in range(B)]
    pipe = pipeline.FrontendBatch(fe, ctx, w, h, B, 500, 100, "cuda:0")
    imgs = torch.from_numpy(np.stack(frames)).cuda()
    prev = torch.from_numpy(np.stack([warp_prev(f) for f in frames])).cuda()
    pipe.extract(prev, "prev")
    pipe.step(imgs)
    torch.cuda.synchronize()
    c = pipe.feat["cur"]
    n = c["n"].cpu().numpy(); nl = c["nl"].cpu().numpy()
    m12 = pipe.m12.cpu().numpy(); nm = pipe.nmatch.cpu().numpy(); lp = pipe.lpairs.cpu().numpy(); nlp = pipe.nlpairs.cpu().numpy()
    for i, f in enumerate(frames):
        okp, od = oracle.orb_extract(f, 500); okl, old, ofn, _ = oracle.lines_extract(f, 100)
        assert n[i] == len(okp) and nl[i] == len(okl), i
        np.testing.assert_array_equal(c["kp"][i, :n[i]].cpu().numpy().view(np.uint8).reshape(-1, 28), okp.view(np.uint8).reshape(-1, 28))
        np.testing.assert_array_equal(c["desc"][i, :n[i]].cpu().numpy(), od)
        np.testing.assert_array_equal(c["ldesc"][i, :nl[i]].cpu().numpy(), old)
        np.testing.assert_array_equal(c["linefn"][i, :nl[i]].cpu().numpy(), ofn)
        kp1, d1 = oracle.orb_extract(warp_prev(f), 500)
        pm = np.stack([kp1["x"], kp1["y"]], axis=1).astype(np.float32)
        om12, _, on = oracle.search_for_initialization(kp1, d1, okp, od, pm, 100, 0.9, True, (0.0, float(w), 0.0, float(h)))
        assert nm[i] == on, i
        np.testing.assert_array_equal(m12[i][:len(kp1)], om12)
        l1 = oracle.lines_extract(warp_prev(f), 100)
        opairs, _, _ = oracle.line_match(l1[1], old, 0.5, False)
        assert nlp[i] == len(opairs)
        np.testing.assert_array_equal(lp[i][:nlp[i]], opairs)
    pipe.close()


@pytest.mark.parametrize("form", ["one_stream", "guest"])
def test_batch_of_2176_frames_throughput_kernels(fe, ctx, oracle, form, monkeypatch):
    """from 1024 frames on the LSD core runs its six-waves-per-SIMD flavour (k_lsd_regions<false, 6>) and the NFA stages one wave per frame
    (from 2048 on also the rectangle counter): the kernels the benchmark times.  2176 = 17 * 128 small frames, 17 distinct ones tiled, so
    every distinct frame lands on many different workgroups / XCDs; a sample of slots is compared with the oracle and all copies of a
    frame must agree with each other byte for byte.
    form "guest" (round 6): the two-stream step with a core event announced and a persistent grid of 136 workgroups (SSLAM_LSD_PERSIST; the library's own grid, 16 per
    compute unit, would be larger than this batch) -- k_lsd_regions<false, 4> claiming its 2 176 frames dynamically, the pyramid built ahead of the event, FAST .. gated on it."""
    pipeline = pkg._load("sslam_pipeline", os.path.join(pkg.PKG_DIR, "pipeline.py"))
    w, h, U, REP = (240, 180, 17, 128) if form == "guest" else (192, 144, 17, 128)      # 240 x 180 is a 5-to-4 geometry (fused gradient kernel), 192 x 144 is not (k_blur7 + k_lsd_grad)
    B = U * REP
    frames = [synth_frame(5000 + i, w, h, nshapes=8 + 3 * i, nstrokes=2 * i, noise=float(i % 3)) for i in range(U)]
    if form == "guest": monkeypatch.setenv("SSLAM_LSD_PERSIST", "136")      # (& ~7 = 136 workgroups: 16 rounds of this batch)
    pipe = pipeline.FrontendBatch(fe, ctx, w, h, B, 300, 60, "cuda:0", with_match=False)
    imgs = torch.from_numpy(np.stack(frames)).cuda().repeat(REP, 1, 1).contiguous()          # slot s holds frame s % 17
    if form == "guest":
        for _ in range(2): pipe.step(imgs, overlap=True)      # twice: the frame counter of the persistent grid is reset per launch
        assert fe.lib().sslam_lines_core_guest_form(pipe.lines.h, B) == 1 and pipe._gate_in_orb
    else:
        pipe.step(imgs)
    torch.cuda.synchronize()
    c = pipe.feat["cur"]
    n = c["n"].cpu().numpy(); nl = c["nl"].cpu().numpy()
    kp = c["kp"].cpu().numpy(); desc = c["desc"].cpu().numpy(); kl = c["kl"].cpu().numpy(); ld = c["ldesc"].cpu().numpy(); fn = c["linefn"].cpu().numpy()
    total_lines = 0
    for i, f in enumerate(frames):
        okp, od = oracle.orb_extract(f, 300); okl, old, ofn, _ = oracle.lines_extract(f, 60)
        total_lines += len(okl)
        slots = np.arange(i, B, U)
        assert (n[slots] == len(okp)).all() and (nl[slots] == len(okl)).all(), i
        for s in (slots[0], slots[37], slots[-1]):
            np.testing.assert_array_equal(kp[s, :n[s]].view(np.uint8).reshape(-1, 28), okp.view(np.uint8).reshape(-1, 28))
            np.testing.assert_array_equal(desc[s, :n[s]], od)
            a = kl[s, :nl[s]].view(np.uint8).reshape(-1, 68).copy(); b = okl.view(np.uint8).reshape(-1, 68).copy()
            a[:, 0:4] = 0; b[:, 0:4] = 0                                # KeyLine.angle: atan2, <= 1 ulp (test_lines_gpu)
            np.testing.assert_array_equal(a, b)
            np.testing.assert_array_equal(ld[s, :nl[s]], old); np.testing.assert_array_equal(fn[s, :nl[s]], ofn)
        ref = slots[0]
        for arr, cnt in ((kp, n[ref]), (desc, n[ref]), (kl, nl[ref]), (ld, nl[ref]), (fn, nl[ref])):
            assert (arr[slots, :cnt].view(np.uint8) == arr[ref, :cnt].view(np.uint8)).all(), i          # every copy identical, wherever it ran (rows past the count are unspecified)
    assert total_lines > 100
    pipe.close()


def test_frontend_batch_host_buffers(fe, ctx, oracle):
    """sslam_frontend_batch: host images in, host records out, chunked with copy/compute overlap.  37 frames in chunks of 8
    (five chunks, the last one short, both buffer slots reused) must equal the per-frame host entry points frame by frame."""
    w, h, n = 256, 192, 37
    frames = np.stack([synth_frame(6000 + i, w, h, nshapes=10 + i, nstrokes=i % 7, noise=float(i % 3)) for i in range(n)])
    orb = fe.OrbExtractor(ctx, 400); lines = fe.LineExtractor(ctx, 80)
    out = fe.frontend_batch(orb, lines, frames, chunk=8)
    assert len(out) == n
    for i in (0, 7, 8, 15, 16, 31, 32, 36):
        kp, d = orb(frames[i]); kl, ld, fn = lines(frames[i])
        bkp, bd, bkl, bld, bfn = out[i]
        assert len(bkp) == len(kp) and len(bkl) == len(kl), i
        np.testing.assert_array_equal(bkp.view(np.uint8), kp.view(np.uint8)); np.testing.assert_array_equal(bd, d)
        np.testing.assert_array_equal(bkl.view(np.uint8), kl.view(np.uint8)); np.testing.assert_array_equal(bld, ld); np.testing.assert_array_equal(bfn, fn)
    okp, od = oracle.orb_extract(frames[20], 400)
    np.testing.assert_array_equal(out[20][0].view(np.uint8).reshape(-1, 28), okp.view(np.uint8).reshape(-1, 28)); np.testing.assert_array_equal(out[20][1], od)
    pin = torch.empty(frames.shape, dtype=torch.uint8, pin_memory=True); pin.numpy()[:] = frames
    direct = fe.frontend_batch(orb, lines, pin.numpy(), chunk=16, pinned=True)          # pinned memory in and out: no staging copies
    for i in (0, 15, 16, 36):
        for a, b in zip(direct[i], out[i]):
            np.testing.assert_array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))
    only = fe.frontend_batch(orb, None, frames[:5])                       # ORB only, one chunk
    np.testing.assert_array_equal(only[3][1], out[3][1]); assert len(only[3][2]) == 0
    orb.close(); lines.close()


def test_frontend_batch_match_stage(fe, ctx, oracle):
    """sslam_frontend_batch_match (BASELINE configs[2] through host buffers): frame i against frame i-1 of the call -- SearchForInitialization,
    the dense 2-NN and the line matcher behind the chunked extraction.  23 frames in chunks of 6 (the predecessor of a chunk's first frame
    comes from the chunk before): every pair must equal the single-call host matchers on the per-frame results, two pairs are checked
    against the oracle directly, and the extraction half must equal sslam_frontend_batch."""
    w, h, n = 256, 192, 23
    base = synth_frame(7100, w, h, nshapes=18, nstrokes=5, noise=1.0)
    frames = np.stack([np.roll(base, (i // 3, i), axis=(0, 1)) if i % 5 else synth_frame(7200 + i, w, h, nshapes=12 + i, nstrokes=i % 4, noise=1.0) for i in range(n)])
    orb = fe.OrbExtractor(ctx, 400); lines = fe.LineExtractor(ctx, 80)
    ref = fe.frontend_batch(orb, lines, frames, chunk=6)
    for pinned in (False, True):
        out = fe.frontend_batch_alloc(n, orb.cap, 80, pinned=pinned); mout = fe.frontend_batch_match_alloc(n, orb.cap, 80, pinned=pinned)
        src = frames
        if pinned:
            pin = torch.empty(frames.shape, dtype=torch.uint8, pin_memory=True); pin.numpy()[:] = frames; src = pin.numpy()
        fe.frontend_batch_match_raw(orb, lines, src, out, mout, chunk=6, bounds=(0.0, float(w), 0.0, float(h)))
        kp, desc, nk, kl, ld, fn, nl = out
        m12, nm, ki, kd, lp, nlp = mout
        assert nm[0] == 0 and nlp[0] == 0
        total = 0
        for i in range(n):
            np.testing.assert_array_equal(kp[i, :nk[i]].view(np.uint8), ref[i][0].view(np.uint8)); np.testing.assert_array_equal(desc[i, :nk[i]], ref[i][1])
            np.testing.assert_array_equal(ld[i, :nl[i]], ref[i][3])
            if i == 0:
                continue
            k1, d1, l1 = ref[i - 1][0], ref[i - 1][1], ref[i - 1][3]
            k2, d2, l2 = ref[i][0], ref[i][1], ref[i][3]
            pm = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)
            em12, _, enm = ctx.search_for_initialization(k1, d1, k2, d2, pm, 100, 0.9, True, (0.0, float(w), 0.0, float(h)))
            np.testing.assert_array_equal(m12[i, :len(k1)], em12); assert nm[i] == enm, i
            eidx, edist = ctx.hamming_knn2(d1, d2)
            np.testing.assert_array_equal(ki[i, :len(k1)], eidx); np.testing.assert_array_equal(kd[i, :len(k1)], edist)
            ep, _, _ = ctx.line_match(l1, l2, 0.5, False)
            assert nlp[i] == len(ep), i
            np.testing.assert_array_equal(lp[i, :nlp[i]], ep)
            total += enm
            if i in (6, 13):      # chunk boundaries (6 | 12 | 18): the pair straddles two chunks for i = 6; against the oracle itself
                om12, _, onm = oracle.search_for_initialization(k1, d1, k2, d2, pm, 100, 0.9, True, (0.0, float(w), 0.0, float(h)))
                np.testing.assert_array_equal(m12[i, :len(k1)], om12); assert nm[i] == onm
        assert total > 200
    only = fe.frontend_batch_alloc(5, orb.cap, 1); monly = fe.frontend_batch_match_alloc(5, orb.cap, 1, knn=False)          # ORB only, no 2-NN, one chunk
    fe.frontend_batch_match_raw(orb, None, frames[:5], only, monly)
    np.testing.assert_array_equal(monly[0][3, :len(ref[2][0])], m12[3, :len(ref[2][0])])
    orb.close(); lines.close()
