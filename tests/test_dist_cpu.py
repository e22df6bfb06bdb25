"""CPU suite: the N>1 path (frame sharding + the single final gather) with world_size 2 on gloo.
Each rank packs oracle-produced per-frame records for its shard; rank 0 must end up with exactly
the concatenation of the single-process results, in global frame order."""
import os, sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import pkg

HERE = os.path.dirname(os.path.abspath(__file__))
CAP, LCAP = 600, 60
NFRAMES = 4


def _frame_record(i):
    sys.path.insert(0, HERE)
    import oracle_lib
    from synth import synth_frame
    orc = oracle_lib.Oracle()
    img = synth_frame(3000 + i, w=256, h=192)
    kp, desc = orc.orb_extract(img, 300)
    kl, ld, fn, raw = orc.lines_extract(img, LCAP)
    n, nl = len(kp), len(kl)
    kpa = np.zeros((CAP, 28), np.uint8); kpa[:n] = kp.view(np.uint8).reshape(n, 28)
    da = np.zeros((CAP, 32), np.uint8); da[:n] = desc
    kla = np.zeros((LCAP, 68), np.uint8); kla[:nl] = kl.view(np.uint8).reshape(nl, 68)
    lda = np.zeros((LCAP, 32), np.uint8); lda[:nl] = ld
    return n, nl, kpa, da, kla, lda


def _pack(sh, idxs):
    recs = [_frame_record(i) for i in idxs]
    t = lambda k, dt: torch.from_numpy(np.stack([np.asarray(r[k]) for r in recs]).astype(dt))
    n = t(0, np.int32); nl = t(1, np.int32)
    kp = torch.from_numpy(np.stack([r[2] for r in recs])).view(torch.float32).reshape(len(idxs), CAP, 7)
    desc = torch.from_numpy(np.stack([r[3] for r in recs]))
    kl = torch.from_numpy(np.stack([r[4] for r in recs])).view(torch.float32).reshape(len(idxs), LCAP, 17)
    ld = torch.from_numpy(np.stack([r[5] for r in recs]))
    return sh.pack_records(n, nl, kp, desc, kl, ld)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = pkg._load("sslam_sharding", os.path.join(pkg.PKG_DIR, "sharding.py"))
    rec = _pack(sh, sh.shard_indices(NFRAMES, world, rank))
    out = sh.gather_to_root(dist, rec, world, rank)
    ag = sh.AsyncGather(dist, world, rank)             # the overlapped form used by bench.py: two steps back to back
    ag.submit(rec.clone()); ag.submit(rec)
    out2 = ag.result()
    if rank == 0:
        assert torch.equal(out, out2)
        q.put(out.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_indices():
    sh = pkg._load("sslam_sharding", os.path.join(pkg.PKG_DIR, "sharding.py"))
    assert sh.shard_indices(8, 8, 3) == [3]
    assert sh.shard_indices(10, 4, 1) == [1, 5, 9]
    assert sorted(sum((sh.shard_indices(10, 4, r) for r in range(4)), [])) == list(range(10))


@pytest.mark.timeout(300)
def test_gather_equals_single_process_concatenation():
    sh = pkg._load("sslam_sharding", os.path.join(pkg.PKG_DIR, "sharding.py"))
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    gathered = q.get(timeout=240)
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    single = _pack(sh, list(range(NFRAMES))).numpy()
    np.testing.assert_array_equal(gathered, single)
    lay = sh.record_layout(CAP, LCAP)
    assert gathered.shape == (NFRAMES, lay["size"])
    r0 = sh.unpack_record(gathered[1], CAP, LCAP)
    n, nl, kpa, da, kla, lda = _frame_record(1)
    assert r0["n"] == n and r0["nl"] == nl
    np.testing.assert_array_equal(r0["desc"], da[:n]); np.testing.assert_array_equal(r0["kl"], kla[:nl])
