#!/usr/bin/env python3
"""bench.py — front-end frames/sec (ORB + LSD/LBD extract + match) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

A "step" is one pass of the hot path over one batch of B synthetic 640x480 frames that
are already resident in HBM: ORB extract (1000 kp, 8 levels), LSD+LBD extract (<=200
lines), and matching against the previous frame's features (SearchForInitialization +
dense knn-2 for ORB, knn-2 + MAD gate for lines)  == BASELINE.json configs[2].
With --gpus N (launched under torch.distributed.run, one rank per GPU) every rank processes
its own B frames (weak scaling, frames are independent units) and the per-frame results are
gathered to rank 0 over RCCL once per step; no other collective exists on the path.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel,
algorithmic bytes from SURVEY.md §8(d) / DESIGN.md over the HIP-event launch duration) and
`cpu_baseline` (the CPU oracle = restatement of the reference path, 1 core, bounded sample).
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 640, 480
NFEAT, NLINES = 1000, 200
WORKLOADS = {     # BASELINE.json configs; c3 (configs[2]) is the one the metric is quoted on and the default
    "c2": dict(w=640, h=480, nfeat=1000, nlines=0, match=False, batch=3072, name="BASELINE configs[1]: single synthetic 640x480 frame stream, ORB-only (1000 kp, 8 levels)"),
    "c3": dict(w=640, h=480, nfeat=1000, nlines=200, match=True, batch=6144, name="BASELINE configs[2]: 640x480 ORB(1000kp,8 levels)+LSD/LBD(<=200 lines) extract + Hamming match vs previous frame, inputs resident in HBM"),
    "c4": dict(w=1280, h=960, nfeat=2000, nlines=400, match=True, batch=3072, name="BASELINE configs[3]: 1280x960 ORB(2000kp)+LSD/LBD(<=400 lines) extract + match, inputs resident in HBM"),
}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def alg_bytes_per_frame(w, h, nkp, nln):
    """Algorithmic HBM bytes per frame per kernel (SURVEY.md §8(d) + Appendix B byte model,
    split per kernel in DESIGN.md §roofline).  P = pyramid level pixels."""
    import numpy as np
    P = []
    sc = np.float32(1.0)
    for l in range(8):
        isc = np.float32(1.0) / sc
        P.append(int(np.rint(np.float32(w) * isc)) * int(np.rint(np.float32(h) * isc)))
        sc = np.float32(sc * np.float32(1.2))
    sP = sum(P)
    s08 = int(round(w * 0.8)) * int(round(h * 0.8))
    return {
        "k_copy_level0": w * h + P[0],                      # input read + level-0 write
        "k_resize": sum(P[:-1]) + sum(P[1:]),               # resize reads of level l-1 + writes of level l
        "k_fast_cells": sP,                                 # FAST read of every level
        "k_octree": 0,                                      # candidate lists only (latency-bound, no image traffic)
        "k_describe": 2 * sP + nkp * (2 * 961 + 32 + 28),   # blur r/w folded into the per-keypoint patch stage + patch reads + outputs
        "k_blur7": 2 * w * h,                               # LSD pre-blur r/w
        "k_lsd_grad": (w * h + s08) + s08 + 8 * s08,        # 0.8x resample (fused: blurred read, scaled image) + gradient read, fp32 angle + int magnitude write
        "k_lsd_hist": 8 * s08, "k_lsd_scan": 0, "k_lsd_scatter": 4 * s08,    # ordered-list build
        "k_lsd_regions": 5 * s08,                           # region-grow reads (angle + magnitude + used)
        "k_nfa_count": 0, "k_nfa_eval": 0, "k_nfa_accept": 0, "k_nfa_finish": 0,      # rectangle validation: angle-map rows under ~400 rectangles, cache resident
        "k_keylines": nln * (16 + 68 + 24),
        "k_blur_sobel": w * h + 4 * w * h,                  # fused LBD pre-blur + Sobel: source read, {dx,dy} s16 pair write
        "k_lbd": nln * 63 * 100 * 4 + nln * 100,            # band reads at a nominal 100-px line + descriptor
        "k_search_init": 2 * 32 * nkp + 8 * nkp, "k_knn2_batch": 2 * 32 * nkp + 16 * nkp,
        "k_line_match": 2 * 32 * nln + 16 * nln,
        "k_zero_misc": 0,
    }


def cpu_baseline(frames_cur, frames_prev, budget_s=12.0, max_frames=600, with_lines=True, with_match=True):
    """The CPU oracle (a restatement of the reference's CPU path; kind "port") timed on this box's
    host cores, single thread like the reference's front-end (src/Frame.cc:86-87)."""
    import numpy as np
    import oracle_lib
    orc = oracle_lib.Oracle()
    t0 = time.perf_counter()
    n = 0
    prev_feat = None
    # prime "previous" features outside the timed loop
    pk, pd = orc.orb_extract(frames_prev[0], NFEAT)
    pl = orc.lines_extract(frames_prev[0], NLINES) if with_lines else None
    t0 = time.perf_counter()
    ref = []        # the oracle's outputs for the first pass over the distinct frames: compared with the GPU batch after the timing
    while n < max_frames and (time.perf_counter() - t0) < budget_s:
        cur = frames_cur[n % len(frames_cur)]
        kp, d = orc.orb_extract(cur, NFEAT)
        if with_lines:
            kl, ld, fn, raw = orc.lines_extract(cur, NLINES)
        if n < len(frames_cur):
            ref.append((kp, d, kl if with_lines else None, ld if with_lines else None))
        if with_match:
            pm = np.stack([pk["x"], pk["y"]], axis=1).astype(np.float32)
            orc.search_for_initialization(pk, pd, kp, d, pm, 100, 0.9, True, (0.0, float(W), 0.0, float(H)))
            orc.knn2(pd, d)
            if with_lines:
                orc.line_match(pl[1], ld, 0.5, False)
        n += 1
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of the same %dx%d workload (ORB %d%s%s), %.1f s, single thread" % (n, W, H, NFEAT, " + LSD/LBD %d" % NLINES if with_lines else "", " + matches" if with_match else "", dt)}
    # the same oracle as N independent single-threaded processes, one pinned per host core (informational; `value` stays the
    # single-thread figure, the reference's front-end being single-threaded).  A separate interpreter without torch; any failure just omits it.
    try:
        import subprocess
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_allcores.py"), str(W), str(H), str(NFEAT), str(NLINES if with_lines else 0), "6", "256"],
                           capture_output=True, text=True, timeout=120)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        out["all_cores"] = json.loads(line[-1]) if r.returncode == 0 and line else None
    except Exception:
        out["all_cores"] = None
    return out, ref


def parity_vs_gpu(ref, feat, with_lines):
    """The oracle outputs of the baseline leg against what the timed GPU batch holds for the same frames (batch slot i = distinct
    frame i): keypoints + descriptors byte for byte, keylines byte for byte except KeyLine.angle (<= 1 ulp), LBD bytes."""
    import numpy as np
    res = {"frames": len(ref), "orb_equal": 0, "lines_equal": 0 if with_lines else None}
    n = feat["n"][:len(ref)].cpu().numpy(); nl = feat["nl"][:len(ref)].cpu().numpy()
    for i, (kp, d, kl, ld) in enumerate(ref):
        gk = feat["kp"][i, :n[i]].cpu().numpy().view(np.uint8).reshape(-1, 28); gd = feat["desc"][i, :n[i]].cpu().numpy()
        res["orb_equal"] += int(n[i] == len(kp) and np.array_equal(gk, kp.view(np.uint8).reshape(-1, 28)) and np.array_equal(gd, d))
        if with_lines:
            gl = feat["kl"][i, :nl[i]].cpu().numpy().view(np.uint8).reshape(-1, 68).copy(); ol = kl.view(np.uint8).reshape(-1, 68).copy()
            gl[:, 0:4] = 0; ol[:, 0:4] = 0
            res["lines_equal"] += int(nl[i] == len(kl) and np.array_equal(gl, ol) and np.array_equal(feat["ldesc"][i, :nl[i]].cpu().numpy(), ld))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (default: the workload's)")
    ap.add_argument("--unique", type=int, default=8, help="distinct synthetic frames (tiled to the batch)")
    ap.add_argument("--overlap", action="store_true", help="run the point and line branches on two streams (off by default: the persistent LSD kernel wants every wave slot, sharing them costs a second round)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event per-kernel timing (used for rocprofv3 runs)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this host driver (already exported on the GPU boxes)
    import numpy as np
    import torch
    import pkg
    from synth import synth_frame, warp_prev
    global W, H, NFEAT, NLINES
    wl = WORKLOADS[args.workload]
    W, H, NFEAT, NLINES = wl["w"], wl["h"], wl["nfeat"], max(wl["nlines"], 1)
    if args.batch <= 0:
        args.batch = wl["batch"]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        print("bench.py: --gpus %d needs `python -m torch.distributed.run --nproc-per-node %d`" % (args.gpus, args.gpus), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; this framework has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    dist = None
    # SSLAM_FORCE_COLLECTIVE=1 (tests): run the RCCL exchange step even with a single rank, so the collective path is exercised on a 1-GPU box
    force_collective = os.environ.get("SSLAM_FORCE_COLLECTIVE") == "1" and "RANK" in os.environ
    if world > 1 or force_collective:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    fe = pkg.frontend()
    pipeline = pkg._load("sslam_pipeline", os.path.join(pkg.PKG_DIR, "pipeline.py"))
    ctx = fe.Context(local_rank)
    B = args.batch
    # synthetic frames: `unique` distinct scenes per rank (seeds 2000+), each with its warped previous frame
    U = max(1, min(args.unique, B))
    cur_np = [synth_frame(2000 + rank * 64 + i, W, H) for i in range(U)]
    prev_np = [warp_prev(f) for f in cur_np]
    reps = (B + U - 1) // U
    cur = torch.from_numpy(np.stack(cur_np)).to(dev).repeat(reps, 1, 1)[:B].contiguous()
    prev = torch.from_numpy(np.stack(prev_np)).to(dev).repeat(reps, 1, 1)[:B].contiguous()

    pipe = pipeline.FrontendBatch(fe, ctx, W, H, B, NFEAT, NLINES, dev, with_lines=wl["nlines"] > 0, with_match=wl["match"])
    # previous-frame features: extracted once, resident (the stream's t-1 state)
    pipe.extract(prev, "prev")
    torch.cuda.synchronize()

    sharding = pkg._load("sslam_sharding", os.path.join(pkg.PKG_DIR, "sharding.py"))

    gather = sharding.AsyncGather(dist, world, rank, always_collective=force_collective)

    def one_step():
        pipe.step(cur, overlap=args.overlap)
        if dist is not None:      # the one exchange step of the path: per-frame records to rank 0 over RCCL,
            gather.submit(pipe.packed_results())      # overlapped with the next step's kernels

    for _ in range(args.warmup):
        one_step()
    gather.wait()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    if not args.no_profile:
        fe.lib().sslam_profile_enable(ctx.h, 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    gather.wait()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fe.lib().sslam_profile_enable(ctx.h, 0)
    prof = pipeline.profile_drain(fe, ctx) if not args.no_profile else {}
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    gather_ok = None
    if rank == 0 and dist is not None:
        # the gathered records of the last step must be this rank's own packed results at every slot it owns (global frame i = local i // world on rank i % world);
        # reported in the JSON line (config.gather_check), never fatal for the measurement
        try:
            got = gather.result()
            mine = pipe.packed_results()
            gather_ok = bool(got.shape[0] == B * world and torch.equal(got[0::world], mine))
            del got, mine
        except Exception as e:
            gather_ok = "error: %s" % (str(e)[:200],)
    if rank == 0:
        counts = pipe.feat["cur"]["n"].cpu().numpy(); lcounts = pipe.feat["cur"]["nl"].cpu().numpy()
        nm = pipe.nmatch.cpu().numpy(); nlp = pipe.nlpairs.cpu().numpy()
        total_frames = B * world * args.steps
        fps = total_frames / dt
        out = {
            "metric": "front-end frames/sec (ORB+LSD extract+match) 640x480 @1000kp/200ln" if args.workload == "c3" else "front-end frames/sec, workload " + args.workload,
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl["name"],
                       "batch_per_gpu": B, "global_batch": B * world, "unique_frames": U,
                       "mean_keypoints": float(counts.mean()), "mean_lines": float(lcounts.mean()),
                       "mean_orb_matches": float(nm.mean()), "mean_line_matches": float(nlp.mean()),
                       "parallelism": "frames sharded %d/GPU, RCCL gather of results to rank 0 per step" % B if world > 1 else "single GPU",
                       "gather_check": gather_ok},
        }
        if prof:
            ab = alg_bytes_per_frame(W, H, NFEAT, NLINES)
            dom = max(prof.items(), key=lambda kv: kv[1][0])
            name, (ms, launches) = dom
            avg_s = ms / launches * 1e-3
            bytes_per_launch = ab.get(name, 0) * B
            ach = bytes_per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
            traffic = None
            try:      # HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json), scaled to this batch
                pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                pt = pj["kernels"][name]      # fetch side x2: gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, calibrated in profiles/)
                traffic = (pt["fetch_bytes_per_frame"] * pj.get("fetch_correction", 1.0) + pt["write_bytes_per_frame"]) * B
            except Exception:
                pass
            out["roofline"] = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                               "traffic": traffic, "avg_launch_ms": ms / launches, "launches": launches,
                               "alg_bytes_per_launch": bytes_per_launch,
                               "whole_pipeline": {"alg_bytes_per_frame": sum(ab.values()), "achieved": sum(ab.values()) * fps / world / 1e9,
                                                  "frac": sum(ab.values()) * fps / world / 1e9 / HBM_PEAK_GBS},
                               "kernels_ms_per_step": {k: v[0] / args.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], ref = cpu_baseline(cur_np, prev_np, with_lines=wl["nlines"] > 0, with_match=wl["match"])
            out["cpu_baseline"]["parity_vs_gpu"] = parity_vs_gpu(ref, pipe.feat["cur"], wl["nlines"] > 0)
        print(json.dumps(out))
    pipe.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
