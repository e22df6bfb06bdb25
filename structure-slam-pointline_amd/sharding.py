"""Frame sharding and the final result gather for the batch-of-frames mode.

Frames are independent units: global frame i of a batch of n goes to rank i % world (round-robin),
every rank runs the same single-GPU pipeline on its shard, and ONE exchange step at the end gathers
the fixed-capacity per-frame records to rank 0.  No other collective exists on the path."""
import numpy as np
import torch


def shard_indices(n_frames, world, rank):
    return list(range(rank, n_frames, world))


def record_layout(cap, lcap):
    """byte offsets of one frame record: [n_kp i32][n_ln i32][kp cap*28][desc cap*32][kl lcap*68][ldesc lcap*32]"""
    o = {"n": 0, "nl": 4, "kp": 8}
    o["desc"] = o["kp"] + cap * 28
    o["kl"] = o["desc"] + cap * 32
    o["ldesc"] = o["kl"] + lcap * 68
    o["size"] = o["ldesc"] + lcap * 32
    return o


def pack_records(n, nl, kp, desc, kl, ldesc):
    """torch tensors [B], [B], [B,cap,7] f32, [B,cap,32] u8, [B,lcap,17] f32, [B,lcap,32] u8 -> [B, size] u8"""
    B = n.shape[0]
    parts = [n.view(torch.uint8).reshape(B, -1), nl.view(torch.uint8).reshape(B, -1), kp.view(torch.uint8).reshape(B, -1),
             desc.reshape(B, -1), kl.view(torch.uint8).reshape(B, -1), ldesc.reshape(B, -1)]
    return torch.cat(parts, dim=1).contiguous()


def unpack_record(rec, cap, lcap):
    """one frame record (1-D uint8 numpy) -> dict of numpy arrays trimmed to the counts"""
    lay = record_layout(cap, lcap)
    rec = np.ascontiguousarray(rec)
    n = int(rec[0:4].view(np.int32)[0]); nl = int(rec[4:8].view(np.int32)[0])
    kp = rec[lay["kp"]:lay["desc"]].reshape(cap, 28)[:n]
    desc = rec[lay["desc"]:lay["kl"]].reshape(cap, 32)[:n]
    kl = rec[lay["kl"]:lay["ldesc"]].reshape(lcap, 68)[:nl]
    ldesc = rec[lay["ldesc"]:lay["size"]].reshape(lcap, 32)[:nl]
    return {"n": n, "nl": nl, "kp": kp, "desc": desc, "kl": kl, "ldesc": ldesc}


def gather_to_root(dist, rec, world, rank):
    """rec: [B_local, size] uint8 (same B_local on every rank).  Returns on rank 0 the records in GLOBAL
    frame order (frame i lives on rank i % world at local slot i // world); None elsewhere."""
    if world == 1:
        return rec
    bufs = [torch.empty_like(rec) for _ in range(world)] if rank == 0 else None
    dist.gather(rec, bufs, dst=0)
    if rank != 0:
        return None
    stacked = torch.stack(bufs, dim=1)            # [B_local, world, size] -> global index = local*world + rank
    return stacked.reshape(-1, rec.shape[1])


class AsyncGather:
    """The one exchange step, overlapped with the next step's compute: gather(step k) runs on RCCL's stream
    while the kernels of step k+1 execute; at most one gather is in flight."""
    def __init__(self, dist, world, rank, always_collective=False):
        self.dist, self.world, self.rank = dist, world, rank
        self.local_only = world == 1 and not (always_collective and dist is not None)
        self.work = None; self.bufs = None; self.rec = None

    def submit(self, rec):
        self.wait()
        if self.local_only:
            self.rec = rec
            return
        if self.rank == 0 and (self.bufs is None or self.bufs[0].shape != rec.shape):
            self.bufs = [torch.empty_like(rec) for _ in range(self.world)]
        self.rec = rec                     # keep the source alive until the collective has consumed it
        self.work = self.dist.gather(rec, self.bufs if self.rank == 0 else None, dst=0, async_op=True)

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None

    def result(self):
        """records in global frame order on rank 0 (None elsewhere)"""
        self.wait()
        if self.local_only:
            return self.rec
        if self.rank != 0:
            return None
        return torch.stack(self.bufs, dim=1).reshape(-1, self.rec.shape[1])
