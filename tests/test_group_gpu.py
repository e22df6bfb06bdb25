"""GPU: the multi-GPU entry points of the C ABI on the one GPU a test box has -- record stream pack / unpack, the single-process group
(ngpu = 1, with and without routing the root's own records through ncclSend / ncclRecv), and the one-process-per-GPU form with a world of
one rank (unique id, ncclCommInitRank, the 8-byte all-gather of the lengths, a self send/recv of the payload).  N > 1 is the driver's
scaling run; the CPU suite covers the N = 2 bookkeeping with gloo (tests/test_dist_cpu.py)."""
import os
import numpy as np
import pytest
import torch
import pkg
from synth import synth_frame

pytestmark = pytest.mark.gpu
W, H = 256, 192


def _frames(n):
    return np.stack([synth_frame(8100 + i, W, H, nshapes=6 + 3 * i, nstrokes=i % 5, noise=float(i % 3)) if i != 3 else np.full((H, W), 70, np.uint8) for i in range(n)])


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert len(x) == len(y)
        for u, v in zip(x, y):
            np.testing.assert_array_equal(np.ascontiguousarray(u).view(np.uint8), np.ascontiguousarray(v).view(np.uint8))


def test_pack_unpack_records(fe, ctx):
    pipeline = pkg._load("sslam_pipeline", os.path.join(pkg.PKG_DIR, "pipeline.py"))
    B = 9
    frames = _frames(B)
    pipe = pipeline.FrontendBatch(fe, ctx, W, H, B, 400, 80, "cuda:0", with_match=False)
    pipe.step(torch.from_numpy(frames).cuda())
    torch.cuda.synchronize()
    c = pipe.feat["cur"]
    capb = fe.record_stream_capacity(B, pipe.cap, pipe.lcap)
    out = torch.zeros(capb, dtype=torch.uint8, device="cuda"); tot = torch.zeros(1, dtype=torch.int64, device="cuda")
    fe.pack_records_dev(ctx, B, 5, 3, c["kp"], c["desc"], c["n"], pipe.cap, c["kl"], c["ldesc"], c["linefn"], c["nl"], pipe.lcap, out, capb, tot)
    ctx.synchronize()
    n = c["n"].cpu().numpy(); nl = c["nl"].cpu().numpy()
    total = int(tot.item())
    assert total == sum(fe.record_bytes(int(a), int(b)) for a, b in zip(n, nl)) and total % 16 == 0
    kp, desc, nk, kl, ld, fn, nll, nrec = fe.unpack_records(out[:total].cpu().numpy(), 5 + 3 * B, pipe.cap, pipe.lcap)
    assert nrec == B and n[3] == 0 and nl[3] == 0
    for i in range(B):
        f = 5 + 3 * i                                  # header.frame = frame0 + i * frame_step
        assert nk[f] == n[i] and nll[f] == nl[i]
        np.testing.assert_array_equal(kp[f, :n[i]].view(np.uint8).reshape(-1, 28), c["kp"][i, :n[i]].cpu().numpy().view(np.uint8).reshape(-1, 28))
        np.testing.assert_array_equal(desc[f, :n[i]], c["desc"][i, :n[i]].cpu().numpy())
        np.testing.assert_array_equal(kl[f, :nl[i]].view(np.uint8).reshape(-1, 68), c["kl"][i, :nl[i]].cpu().numpy().view(np.uint8).reshape(-1, 68))
        np.testing.assert_array_equal(ld[f, :nl[i]], c["ldesc"][i, :nl[i]].cpu().numpy())
        np.testing.assert_array_equal(fn[f, :nl[i]], c["linefn"][i, :nl[i]].cpu().numpy())      # the 24-byte line equations travel too
    assert (nk[[0, 1, 2, 3, 4, 6, 7]] == -1).all()                                                # frames nobody sent stay untouched
    # too small a stream buffer is reported through the length word, nothing is written past it
    small = torch.zeros(total - 16, dtype=torch.uint8, device="cuda")
    fe.pack_records_dev(ctx, B, 0, 1, c["kp"], c["desc"], c["n"], pipe.cap, c["kl"], c["ldesc"], c["linefn"], c["nl"], pipe.lcap, small, total - 16, tot)
    ctx.synchronize()
    assert int(tot.item()) == -1
    with pytest.raises(fe.SslamError):
        fe.unpack_records(out[:total - 4].cpu().numpy(), 5 + 3 * B, pipe.cap, pipe.lcap)         # truncated stream
    pipe.close()


@pytest.mark.parametrize("self_rccl", [False, True])
def test_group_single_process_ngpu1(fe, ctx, self_rccl):
    """sslam_group_create(1) + sslam_frontend_batch_sharded == sslam_frontend_batch, frame by frame; with SSLAM_GROUP_SELF_SENDRECV the root's
    own records take the ncclSend / ncclRecv path, so RCCL itself moves the bytes even on one GPU"""
    frames = _frames(11)
    orb = fe.OrbExtractor(ctx, 400); lines = fe.LineExtractor(ctx, 80)
    ref = fe.frontend_batch(orb, lines, frames)
    orb.close(); lines.close()
    if self_rccl:
        os.environ["SSLAM_GROUP_SELF_SENDRECV"] = "1"
    try:
        g = fe.Group(ngpu=1)
        assert g.size == 1 and g.rank == 0
        got = g.frontend_batch_sharded(frames, 400, 80)
        _same(got, ref)
        got2 = g.frontend_batch_sharded(frames[:3], 400, 0)          # other parameters on the same group: ORB only
        for a, b in zip(got2, ref[:3]):
            np.testing.assert_array_equal(a[1], b[1]); assert len(a[2]) == 0
        with pytest.raises(fe.SslamError):
            g.frontend_batch_sharded(frames, 400, 80, lcap=3)        # capacity errors come back like sslam_frontend_batch's
        g.close()
    finally:
        os.environ.pop("SSLAM_GROUP_SELF_SENDRECV", None)
    with pytest.raises(fe.SslamError):
        fe.Group(ngpu=2)                                             # the box has one GPU: refused, no silent fallback


def test_group_rank_form_world1(fe, ctx):
    os.environ["SSLAM_GROUP_SELF_SENDRECV"] = "1"
    try:
        g = fe.Group(device=0, rank=0, nranks=1, uid=fe.Group.unique_id())
        payload = torch.arange(0, 4096 * 16, dtype=torch.int32, device="cuda").view(torch.uint8)
        nbytes = torch.tensor([payload.numel() - 48], dtype=torch.int64, device="cuda")
        recv = torch.zeros(payload.numel(), dtype=torch.uint8, device="cuda")
        sizes = g.gather_dev(payload, nbytes, recv, recv.numel())
        assert list(sizes) == [payload.numel() - 48]
        assert torch.equal(recv[:payload.numel() - 48], payload[:payload.numel() - 48]) and int(recv[payload.numel() - 48:].sum()) == 0
        with pytest.raises(fe.SslamError):
            g.gather_dev(payload, nbytes, recv, 1024)                # receive buffer too small
        g.close()
    finally:
        os.environ.pop("SSLAM_GROUP_SELF_SENDRECV", None)


# ---------------------------------------------------------------------------------------------------------------------------------------
# N > 1 without an N-GPU box: sslam_testing_use_rccl_standin(1) (include/sslam_testing.h) binds a new group to an in-process stand-in for the RCCL entry points (group.hip), the
# members of a group become contexts (streams) of the GPUs that are visible.  What runs is the group code itself -- one host thread per
# member, the dealing of frames (uneven tails, members without frames), the collective decisions, grouped send / receive to the root,
# reassembly by header.frame -- with real device buffers and real copies.  The xGMI run stays the driver's scaling run.
@pytest.fixture
def fake_rccl(fe):
    # the stand-in exists in libsslam_frontend_testing.so only (the product's sources + -DSSLAM_TESTING): for the duration of the test the binding calls that library
    with fe.use_testing_library() as L:
        L.sslam_testing_use_rccl_standin(1)
        yield
        L.sslam_testing_use_rccl_standin(0)
    os.environ.pop("SSLAM_GROUP_SELF_SENDRECV", None)


def _raw_sharded(fe, g, frames, nfeat, nlines, cap, lcap):
    import ctypes as C
    from oracle_lib import _p
    frames = np.ascontiguousarray(frames, np.uint8); n, h, w = frames.shape
    prm = fe.FrontendParams(nfeat, 1.2, 8, 20, 7, nlines)
    kp = np.zeros((n, cap), fe.KP_DTYPE); desc = np.zeros((n, cap, 32), np.uint8); nk = np.zeros(n, np.int32)
    kl = np.zeros((n, lcap), fe.KL_DTYPE); ld = np.zeros((n, lcap, 32), np.uint8); fn = np.zeros((n, lcap, 3), np.float64); nl = np.zeros(n, np.int32)
    rc = fe.lib().sslam_frontend_batch_sharded(g.h, C.byref(prm), _p(frames), n, w, h, C.c_size_t(w), C.c_size_t(w * h), _p(kp), _p(desc), _p(nk), cap, _p(kl), _p(ld), _p(fn), _p(nl), lcap)
    return rc, (kp, desc, nk, kl, ld, fn, nl)


def _raw_batch(fe, orb, lines, frames, cap, lcap):
    import ctypes as C
    from oracle_lib import _p
    frames = np.ascontiguousarray(frames, np.uint8); n, h, w = frames.shape
    kp = np.zeros((n, cap), fe.KP_DTYPE); desc = np.zeros((n, cap, 32), np.uint8); nk = np.zeros(n, np.int32)
    kl = np.zeros((n, lcap), fe.KL_DTYPE); ld = np.zeros((n, lcap, 32), np.uint8); fn = np.zeros((n, lcap, 3), np.float64); nl = np.zeros(n, np.int32)
    rc = fe.lib().sslam_frontend_batch(orb.h, lines.h, _p(frames), n, w, h, C.c_size_t(w), C.c_size_t(w * h), 0, _p(kp), _p(desc), _p(nk), cap, _p(kl), _p(ld), _p(fn), _p(nl), lcap)
    return rc, (kp, desc, nk, kl, ld, fn, nl)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("G,n,self_rccl", [(2, 11, False), (3, 37, False), (8, 13, True), (5, 3, False), (2, 1061, False)])
def test_sharded_batch_over_several_members(fe, ctx, fake_rccl, G, n, self_rccl):
    """sslam_group_create(G) + sslam_frontend_batch_sharded == sslam_frontend_batch frame by frame: uneven tails (37 over 3), members that hold no
    frame at all (3 frames over 5), more than one chunk per member with an uneven last chunk (1061 over 2: chunks of 512 and 19 / 18 slots)"""
    frames = _frames(min(n, 40)); frames = np.ascontiguousarray(np.tile(frames, ((n + len(frames) - 1) // len(frames), 1, 1))[:n])
    orb = fe.OrbExtractor(ctx, 400); lines = fe.LineExtractor(ctx, 80)
    ref = fe.frontend_batch(orb, lines, frames)
    orb.close(); lines.close()
    if self_rccl:
        os.environ["SSLAM_GROUP_SELF_SENDRECV"] = "1"
    g = fe.Group(ngpu=G)
    try:
        assert g.size == G
        _same(g.frontend_batch_sharded(frames, 400, 80), ref)
        _same(g.frontend_batch_sharded(frames[:G + 1], 400, 80), ref[:G + 1])      # the same group again, another batch size
    finally:
        g.close()


@pytest.mark.timeout(300)
def test_sharded_batch_soft_statuses_from_a_non_root_member(fe, ctx, fake_rccl):
    """a frame that overflows the caller's capacities on a member OTHER than the root: the call returns the deferred status sslam_frontend_batch
    returns for the same batch, and every frame is still delivered, clamped, byte for byte as the single-GPU entry delivers it"""
    n, G = 10, 3
    frames = np.stack([np.full((H, W), 90, np.uint8)] * n)
    frames[4] = synth_frame(8200, W, H, nshapes=40, nstrokes=25, noise=2.0)       # global frame 4 lives on member 4 % 3 = 1
    frames[7] = synth_frame(8201, W, H, nshapes=5, nstrokes=2, noise=0.0)
    orb = fe.OrbExtractor(ctx, 400); lines = fe.LineExtractor(ctx, 80)
    g = fe.Group(ngpu=G)
    try:
        for cap, lcap, want in [(120, 80, fe.SSLAM_ERR_CAPACITY), (orb.cap, 6, fe.SSLAM_ERR_CAPACITY), (orb.cap, 80, 0)]:
            rc_ref, ref = _raw_batch(fe, orb, lines, frames, cap, lcap)
            rc, got = _raw_sharded(fe, g, frames, 400, 80, cap, lcap)
            assert rc == rc_ref == want, (rc, rc_ref, want)
            (kp, desc, nk, kl, ld, fn, nl), (rkp, rdesc, rnk, rkl, rld, rfn, rnl) = got, ref
            np.testing.assert_array_equal(nk, rnk); np.testing.assert_array_equal(nl, rnl)      # counts are clamped to the capacities on both paths
            assert want == 0 or nk.max() == cap or nl.max() == lcap
            for i in range(n):                                                                  # rows up to min(count, capacity) are defined; what lies behind them is not
                a, b = min(int(nk[i]), cap), min(int(nl[i]), lcap)
                np.testing.assert_array_equal(kp[i, :a].view(np.uint8), rkp[i, :a].view(np.uint8)); np.testing.assert_array_equal(desc[i, :a], rdesc[i, :a])
                np.testing.assert_array_equal(kl[i, :b].view(np.uint8), rkl[i, :b].view(np.uint8)); np.testing.assert_array_equal(ld[i, :b], rld[i, :b])
                np.testing.assert_array_equal(fn[i, :b], rfn[i, :b])
    finally:
        g.close(); orb.close(); lines.close()


@pytest.mark.timeout(300)
def test_rank_form_two_ranks_in_one_process(fe, ctx, fake_rccl):
    """sslam_group_create_rank x 2 (two threads, one id) + sslam_group_gather_dev: the length all-gather, the payload to the root, and the
    collective exit when the root's buffer is too small -- BOTH ranks return SSLAM_ERR_CAPACITY, nobody is left blocked in a send"""
    import threading
    uid = fe.Group.unique_id()
    pay = [torch.arange(0, 3000, dtype=torch.int32, device="cuda").view(torch.uint8), (torch.arange(0, 5000, dtype=torch.int32, device="cuda") * 7).view(torch.uint8)]
    nby = [torch.tensor([pay[0].numel() - 16], dtype=torch.int64, device="cuda"), torch.tensor([pay[1].numel()], dtype=torch.int64, device="cuda")]
    recv = torch.zeros(pay[0].numel() + pay[1].numel(), dtype=torch.uint8, device="cuda")
    res = {}

    def work(r):
        try:
            g = fe.Group(device=0, rank=r, nranks=2, uid=uid)
            st = torch.cuda.Stream()
            res[r, "sizes"] = g.gather_dev(pay[r], nby[r], recv if r == 0 else None, recv.numel() if r == 0 else 0, st.cuda_stream)
            try:
                g.gather_dev(pay[r], nby[r], recv if r == 0 else None, 4096 if r == 0 else 0, st.cuda_stream)      # the root offers too little
                res[r, "small"] = "no error"
            except fe.SslamError as e:
                res[r, "small"] = e.code
            res[r, "again"] = g.gather_dev(pay[r], nby[r], recv if r == 0 else None, recv.numel() if r == 0 else 0, st.cuda_stream)      # and the group still works
            g.close()
        except Exception as e:      # pragma: no cover
            res[r, "exc"] = repr(e)
    th = [threading.Thread(target=work, args=(r,)) for r in (0, 1)]
    [t.start() for t in th]; [t.join(120) for t in th]
    assert not any(k[1] == "exc" for k in res), res
    n0, n1 = pay[0].numel() - 16, pay[1].numel()
    assert list(res[0, "sizes"]) == [n0, n1] and res[1, "sizes"] is None and list(res[0, "again"]) == [n0, n1]
    assert res[0, "small"] == res[1, "small"] == fe.SSLAM_ERR_CAPACITY
    torch.cuda.synchronize()
    assert torch.equal(recv[:n0], pay[0][:n0]) and torch.equal(recv[n0:n0 + n1], pay[1])
