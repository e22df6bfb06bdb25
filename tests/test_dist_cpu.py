"""CPU suite: the N>1 path (frame sharding + the single final gather) with world_size 2 on gloo.
Each rank builds the C ABI's record stream (sslam_record_header + keypoints, descriptors, keylines, LBD descriptors, line equations) for
the oracle-produced results of its shard; rank 0 gathers the streams and ingests them with the library's own sslam_unpack_records (host
code: it runs without a GPU).  What rank 0 ends up with must be exactly the single-process results, frame by frame, line equations
included, for an uneven shard split (5 frames over 2 ranks) too."""
import os, sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import pkg

HERE = os.path.dirname(os.path.abspath(__file__))
CAP, LCAP = 600, 60
NFRAMES = 5


def _frame_record(i):
    sys.path.insert(0, HERE)
    import oracle_lib
    from synth import synth_frame
    orc = oracle_lib.Oracle()
    img = synth_frame(3000 + i, w=256, h=192) if i != 2 else np.full((192, 256), 99, np.uint8)      # frame 2: nothing to find
    kp, desc = orc.orb_extract(img, 300)
    kl, ld, fn, raw = orc.lines_extract(img, LCAP)
    return (i, kp.view(np.uint8).reshape(len(kp), 28), desc, kl.view(np.uint8).reshape(len(kl), 68), ld, fn)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = pkg._load("sslam_sharding", os.path.join(pkg.PKG_DIR, "sharding.py"))
    stream = torch.from_numpy(sh.pack_stream_host([_frame_record(i) for i in sh.shard_indices(NFRAMES, world, rank)]))
    out, sizes = sh.gather_streams(dist, stream, world, rank)
    out2, sizes2 = sh.gather_streams(dist, stream.clone(), world, rank)          # a second step on the same group
    if rank == 0:
        assert torch.equal(out, out2) and sizes == sizes2 and len(sizes) == world and sum(sizes) == out.numel()
        q.put((out.numpy(), sizes))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_indices():
    sh = pkg._load("sslam_sharding", os.path.join(pkg.PKG_DIR, "sharding.py"))
    assert sh.shard_indices(8, 8, 3) == [3]
    assert sh.shard_indices(10, 4, 1) == [1, 5, 9]
    assert sorted(sum((sh.shard_indices(10, 4, r) for r in range(4)), [])) == list(range(10))


def test_host_stream_matches_the_c_abi_layout():
    """sharding.pack_stream_host writes what sslam_pack_records_dev writes: the library's unpack reads it back, malformed streams are refused"""
    fe = pkg.frontend()
    sh = pkg._load("sslam_sharding", os.path.join(pkg.PKG_DIR, "sharding.py"))
    recs = [_frame_record(i) for i in (0, 2)]
    stream = sh.pack_stream_host(recs)
    assert stream.size == sum(sh.record_bytes(len(r[1]), len(r[3])) for r in recs) == sum(fe.record_bytes(len(r[1]), len(r[3])) for r in recs)
    kp, desc, nk, kl, ld, fn, nl, nrec = fe.unpack_records(stream, 3, CAP, LCAP)
    assert nrec == 2 and nk[1] == -1 and nk[2] == 0 and nl[2] == 0 and nk[0] == len(recs[0][1]) > 100
    np.testing.assert_array_equal(fn[0, :nl[0]], recs[0][5])
    bad = stream.copy(); bad[12] ^= 0x10                                  # header.bytes no longer matches the counts
    with pytest.raises(fe.SslamError):
        fe.unpack_records(bad, 3, CAP, LCAP)
    with pytest.raises(fe.SslamError):
        fe.unpack_records(stream, 3, 10, LCAP)                            # capacity too small for frame 0


@pytest.mark.timeout(300)
def test_gather_equals_single_process_results():
    fe = pkg.frontend()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    gathered, sizes = q.get(timeout=240)
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    kp, desc, nk, kl, ld, fn, nl, nrec = fe.unpack_records(gathered, NFRAMES, CAP, LCAP)      # rank 0's ingest
    assert nrec == NFRAMES and len(sizes) == 2 and sizes[0] != sizes[1]                       # 3 frames on rank 0, 2 on rank 1
    for i in range(NFRAMES):
        _, okp, od, okl, old, ofn = _frame_record(i)
        assert nk[i] == len(okp) and nl[i] == len(okl), i
        np.testing.assert_array_equal(kp[i, :nk[i]].view(np.uint8).reshape(-1, 28), okp)
        np.testing.assert_array_equal(desc[i, :nk[i]], od)
        np.testing.assert_array_equal(kl[i, :nl[i]].view(np.uint8).reshape(-1, 68), okl)
        np.testing.assert_array_equal(ld[i, :nl[i]], old)
        np.testing.assert_array_equal(fn[i, :nl[i]], ofn)                                     # mvKeyLineFunctions reach rank 0
    assert nk[2] == 0


def test_sharded_batch_bookkeeping_without_devices():
    """sslam_frontend_batch_sharded's dealing of frames over GPUs (csrc/group.hip: sslam_shard_layout / _frame / _chunk_count, the very functions
    its worker threads call): every frame lands in exactly one (GPU, chunk, slot), on GPU frame % G, slots of a chunk are filled front to back
    (the worker processes slots 0 .. count-1), a GPU's frame numbers advance by G (the record header's frame0 + b * step), and uneven tails
    leave some GPUs with a short or empty last chunk.  No device needed: this is the part of the N > 1 path that can run here."""
    import ctypes as C
    L = C.CDLL(pkg.builder().build(force=False, verbose=False))
    for n in (0, 1, 7, 8, 9, 64, 1000, 4095, 4096, 4097, 8 * 512 + 3, 12288):
        for G in (1, 2, 3, 4, 8):
            slots, chunks = C.c_int(-1), C.c_int(-1)
            assert L.sslam_shard_layout(n, G, C.byref(slots), C.byref(chunks)) == 0
            Cs, K = slots.value, chunks.value
            assert 1 <= Cs <= 512 and K * Cs * G >= n and (K == 0) == (n == 0)
            seen = np.zeros(n, np.int32)
            for d in range(G):
                for ck in range(K):
                    c = L.sslam_shard_chunk_count(n, G, ck, d)
                    fr = [L.sslam_shard_frame(n, G, ck, d, j) for j in range(Cs)]
                    assert all(f >= 0 for f in fr[:c]) and all(f < 0 for f in fr[c:])          # a prefix of the slots
                    for j in range(c):
                        assert fr[j] % G == d and fr[j] == fr[0] + j * G                       # what sslam_pack_records_dev stamps: frame0 + b * step
                        seen[fr[j]] += 1
            assert (seen == 1).all()
            if n and n % G:        # uneven tail: the GPUs past the remainder hold one frame fewer in total
                per = [sum(L.sslam_shard_chunk_count(n, G, ck, d) for ck in range(K)) for d in range(G)]
                assert max(per) - min(per) == 1 and sum(per) == n
    assert L.sslam_shard_frame(10, 2, 0, 2, 0) == -1 and L.sslam_shard_frame(10, 2, 5, 0, 0) == -1 and L.sslam_shard_layout(-1, 2, None, None) != 0


def test_bench_gpus_n_launches_itself_or_refuses_in_one_line():
    """`python bench.py --gpus N` outside a launcher starts its own ranks (bench.py self_launch); with fewer GPUs than ranks -- none here -- it says so in one line and
    exits non-zero before any rank is started (the round-5 behaviour was "needs torch.distributed.run", exit 2, on every node)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["HIP_VISIBLE_DEVICES"] = ""      # whatever this host has: none visible to the count
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "--gpus 8 but 0 GPU(s) visible" in r.stderr and "Traceback" not in r.stderr
