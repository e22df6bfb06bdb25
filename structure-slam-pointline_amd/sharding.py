"""Frame sharding and the final result gather for the batch-of-frames mode -- harness side.

Frames are independent units: global frame i of a batch of n goes to rank i % world (round-robin), every rank runs the same
single-GPU pipeline on its shard, and ONE exchange step at the end gathers the per-frame records to rank 0.  No other collective
exists on the path.  The records are the C ABI's record stream (include/sslam_frontend.h: sslam_record_header + keypoints, descriptors,
keylines, LBD descriptors and the 24-byte line equations, compacted to the counts).

  * GroupGather: the product path -- sslam_pack_records_dev + sslam_group_gather_dev (RCCL ncclSend / ncclRecv inside the library),
    run from a side thread so that the exchange of step k overlaps the kernels of step k+1;
  * pack_stream_host / gather_streams: the same stream built with numpy and moved with torch.distributed (gloo) -- the CPU tests of the
    N > 1 bookkeeping (tests/test_dist_cpu.py), where no GPU and no RCCL exist."""
import threading
import numpy as np
import torch


def shard_indices(n_frames, world, rank):
    return list(range(rank, n_frames, world))


def record_bytes(nkp, nl):
    return (16 + nkp * 60 + nl * 124 + 15) & ~15


def pack_stream_host(records):
    """records: iterable of (frame, kp[n,28] u8, desc[n,32] u8, kl[m,68] u8, ldesc[m,32] u8, linefn[m,3] f64) -> uint8 stream"""
    parts = []
    for frame, kp, desc, kl, ld, fn in records:
        n, m = len(kp), len(kl)
        b = record_bytes(n, m)
        body = np.concatenate([np.array([n, m, frame, b], np.int32).view(np.uint8), np.ascontiguousarray(kp, np.uint8).reshape(-1),
                               np.ascontiguousarray(desc, np.uint8).reshape(-1), np.ascontiguousarray(kl, np.uint8).reshape(-1),
                               np.ascontiguousarray(ld, np.uint8).reshape(-1), np.ascontiguousarray(fn, np.float64).reshape(-1).view(np.uint8)])
        parts.append(np.concatenate([body, np.zeros(b - body.size, np.uint8)]))
    return np.concatenate(parts) if parts else np.zeros(0, np.uint8)


def gather_streams(dist, stream, world, rank):
    """stream: 1-D uint8 tensor of this rank (any length).  Returns on rank 0 the streams of all ranks back to back (rank order) and
    the per-rank lengths; (None, None) elsewhere.  Lengths first (all_gather of one int64), then one gather padded to the longest."""
    if world == 1:
        return stream, [int(stream.numel())]
    mine = torch.tensor([stream.numel()], dtype=torch.int64, device=stream.device)
    sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine)
    sizes = [int(s.item()) for s in sizes]
    longest = max(max(sizes), 1)
    padded = torch.zeros(longest, dtype=torch.uint8, device=stream.device); padded[:stream.numel()] = stream
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, bufs, dst=0)
    if rank != 0:
        return None, None
    return torch.cat([bufs[r][:sizes[r]] for r in range(world)]), sizes


class GroupGather:
    """The exchange step through the C ABI (one process per GPU): pack the step's results into the record stream on the device, then
    sslam_group_gather_dev on a side thread -- the gather of step k runs while the kernels of step k+1 execute; at most one is in flight."""

    def __init__(self, fe, ctx, group, pipe, device):
        self.fe, self.ctx, self.group, self.pipe = fe, ctx, group, pipe
        self.world, self.rank = group.size, group.rank
        cap_bytes = fe.record_stream_capacity(pipe.B, pipe.cap, pipe.lcap if pipe.with_lines else 0)
        self.cap_bytes = cap_bytes
        self.send = [torch.zeros(cap_bytes + 16, dtype=torch.uint8, device=device) for _ in range(2)]      # double buffered: step k+1 packs while step k travels
        self.total = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(2)]
        self.recv = torch.zeros(cap_bytes * self.world + 16, dtype=torch.uint8, device=device) if self.rank == 0 else None
        self.k = 0
        self.thread = None; self.sizes = None; self.error = None
        self.wait_s = 0.0; self.waits = 0

    def submit(self):
        """call on the stream the step's kernels were launched on (torch's current stream)"""
        self.wait()
        s = self.k & 1; self.k += 1
        p, c = self.pipe, self.pipe.feat["cur"]
        lines = p.with_lines
        # the pack runs on the pipeline's own (non-default) stream, ordered after this step's kernels of both branches and before the next
        # step's (stream handle 0 would mean "the context's stream" in the C ABI, which is not ordered with the pipeline at all)
        s1, _ = p._streams()
        cur = torch.cuda.current_stream(p.dev)
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            self.fe.pack_records_dev(self.ctx, p.B, self.rank, self.world, c["kp"], c["desc"], c["n"], p.cap,
                                     c["kl"] if lines else None, c["ldesc"] if lines else None, c["linefn"] if lines else None, c["nl"] if lines else None, p.lcap,
                                     self.send[s], self.cap_bytes, self.total[s], p._stream())
            ev = torch.cuda.Event(); ev.record(s1)
        cur.wait_stream(s1)
        self.last = s

        def run():
            try:
                ev.synchronize()
                self.sizes = self.group.gather_dev(self.send[s], self.total[s], self.recv, self.recv.numel() if self.recv is not None else 0)
            except Exception as e:          # reported by wait()
                self.error = e
        self.thread = threading.Thread(target=run); self.thread.start()

    def wait(self):
        if self.thread is not None:
            import time
            t0 = time.perf_counter()
            self.thread.join(); self.thread = None
            self.wait_s += time.perf_counter() - t0; self.waits += 1      # how long the pipeline stood still for the exchange (0 = fully hidden)
            if self.error is not None:
                e, self.error = self.error, None
                raise e

    def result(self):
        """(record streams of all ranks back to back as a device tensor, per-rank lengths) on rank 0, None elsewhere"""
        self.wait()
        if self.rank != 0 or self.sizes is None:
            return None
        return self.recv[:int(self.sizes.sum())], [int(x) for x in self.sizes]


class TorchGather:
    """The same exchange step through torch.distributed (backend "nccl" = RCCL): what bench.py falls back to when the library's own
    communicator cannot be created on some rank (the records are still packed by sslam_pack_records_dev).  Synchronous."""

    def __init__(self, fe, ctx, dist, pipe, device, world, rank):
        self.fe, self.ctx, self.dist, self.pipe, self.world, self.rank = fe, ctx, dist, pipe, world, rank
        self.cap_bytes = fe.record_stream_capacity(pipe.B, pipe.cap, pipe.lcap if pipe.with_lines else 0)
        self.send = torch.zeros(self.cap_bytes + 16, dtype=torch.uint8, device=device)
        self.total = torch.zeros(2, dtype=torch.int64, device=device)
        self.out = None
        self.wait_s = 0.0; self.waits = 0

    def submit(self):
        p, c = self.pipe, self.pipe.feat["cur"]
        lines = p.with_lines
        s1, _ = p._streams()
        cur = torch.cuda.current_stream(p.dev)
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            self.fe.pack_records_dev(self.ctx, p.B, self.rank, self.world, c["kp"], c["desc"], c["n"], p.cap,
                                     c["kl"] if lines else None, c["ldesc"] if lines else None, c["linefn"] if lines else None, c["nl"] if lines else None, p.lcap,
                                     self.send, self.cap_bytes, self.total, p._stream())
        cur.wait_stream(s1)
        n = int(self.total[0].item())
        if n < 0 or n > self.cap_bytes:      # the pack kernel's overflow sentinel is UINT64_MAX, which reads as -1 here: never slice with it
            raise RuntimeError("record stream overflow: the packed results of rank %d do not fit %d bytes" % (self.rank, self.cap_bytes))
        got, sizes = gather_streams(self.dist, self.send[:n], self.world, self.rank)
        self.out = (got, sizes) if self.rank == 0 else None

    def wait(self):
        pass

    def result(self):
        return self.out
