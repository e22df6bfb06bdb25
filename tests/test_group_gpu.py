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
