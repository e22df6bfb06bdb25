#!/usr/bin/env python3
"""bench.py — front-end frames/sec (ORB + LSD/LBD extract + match) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3] [--batch B] [--unique U]

A "step" is one pass of the hot path over one batch of B synthetic 640x480 frames that are already resident in HBM: ORB extract
(1000 kp, 8 levels), LSD+LBD extract (<=200 lines), and matching against the previous frame's features (SearchForInitialization +
dense knn-2 for ORB, knn-2 + MAD gate for lines)  == BASELINE.json configs[2].  With --gpus N (launched under torch.distributed.run,
one rank per GPU) every rank processes its own B frames (weak scaling, frames are independent units) and the compacted per-frame records
are gathered to rank 0 once per step through the library's own RCCL exchange (sslam_group_gather_dev: ncclSend / ncclRecv), overlapped
with the next step's kernels; no other collective exists on the path.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline        dominant kernel: algorithmic bytes (SURVEY.md §8(d) / DESIGN.md) over the HIP-event launch duration; whole pipeline
                  with §8(d)'s own 18 566 506 B/frame next to the per-kernel table's sum
  cpu_baseline    the CPU oracle (restatement of the reference path, "port"), one core, bounded sample, -O3 -march=native built on this box
  latency         single-frame H2D -> kernels -> D2H through the host entry points (what the drop-in shim calls per frame), p50/p90
  pcie_inclusive  sslam_frontend_batch: host images in, host records out (pinned and pageable)
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 640, 480
NFEAT, NLINES = 1000, 200
WORKLOADS = {     # BASELINE.json configs; c3 (configs[2]) is the one the metric is quoted on and the default
    "c2": dict(w=640, h=480, nfeat=1000, nlines=0, match=False, batch=3072, unique=64, name="BASELINE configs[1]: single synthetic 640x480 frame stream, ORB-only (1000 kp, 8 levels)"),
    "c3": dict(w=640, h=480, nfeat=1000, nlines=200, match=True, batch=12288, unique=64, name="BASELINE configs[2]: 640x480 ORB(1000kp,8 levels)+LSD/LBD(<=200 lines) extract + Hamming match vs previous frame, inputs resident in HBM"),
    "c4": dict(w=1280, h=960, nfeat=2000, nlines=400, match=True, batch=3072, unique=16, name="BASELINE configs[3]: 1280x960 ORB(2000kp)+LSD/LBD(<=400 lines) extract + match, inputs resident in HBM"),
    "c5": dict(w=640, h=480, nfeat=1000, nlines=200, match=True, batch=8, unique=8, h2d=True, sync_gather=True,
               name="BASELINE configs[4]: 8 independent 640x480 frames sharded over the GPUs (8/N per GPU), H2D + extract + match + RCCL gather of keypoints/lines inside the timed step"),
}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def level_pixels(w, h):
    import numpy as np
    P, PP = [], []
    sc = np.float32(1.0)
    for l in range(8):
        isc = np.float32(1.0) / sc
        lw, lh = int(np.rint(np.float32(w) * isc)), int(np.rint(np.float32(h) * isc))
        P.append(lw * lh); PP.append((lw + 38) * (lh + 38))
        sc = np.float32(sc * np.float32(1.2))
    return P, PP


def survey_bytes_per_frame(w, h, nkp, nln):
    """SURVEY.md §8(d) / Appendix B byte model, verbatim (18 566 506 B at 640x480 / 1000 / 200; 59 595 532 B at 1280x960 / 2000 / 400)"""
    P, PP = level_pixels(w, h)
    s = int(round(w * 0.8)) * int(round(h * 0.8))
    orb = w * h + sum(PP) + sum(P[:-1]) + sum(P) + 2 * sum(P) + nkp * (2 * 961 + 32 + 28)
    lsd = 3 * w * h + s + s + 8 * s + 4 * s + 5 * s
    lbd = w * h + 4 * w * h + nln * 63 * 100 * 4 + nln * 100
    match = 2 * 32 * nkp + 8 * nkp + 2 * 32 * nln + 16 * nln
    return orb + lsd + lbd + match


def alg_bytes_per_frame(w, h, nkp, nln):
    """Algorithmic HBM bytes per frame per kernel: the same byte model split per kernel as THIS implementation moves the data.
    It differs from §8(d)'s total where the structure differs: no padded pyramid (-0.21 MB), an 8-byte/pixel counting-sort pass the survey's
    "ordered-list write" does not price (+1.57 MB), knn-2 outputs, the rectangle counter's angle-map rows."""
    P, PP = level_pixels(w, h)
    sP = sum(P)
    s08 = int(round(w * 0.8)) * int(round(h * 0.8))
    return {
        "k_copy_level0": w * h + P[0],                      # input read + level-0 write
        "k_resize": sum(P[:-1]) + sum(P[1:]),               # resize reads of level l-1 + writes of level l
        "k_fast_cells": sP,                                 # FAST read of every level
        "k_octree": 0,                                      # candidate lists only (latency-bound, no image traffic)
        "k_describe": 2 * sP + nkp * (2 * 961 + 32 + 28),   # blur r/w folded into the per-keypoint patch stage + patch reads + outputs
        "k_blur7": 0,                                       # LSD pre-blur: evaluated inside the gradient kernel since round 6 (k_lsd_grad_fused; 2 * w * h as a kernel of its own before, and for geometries off the fused path)
        "k_lsd_grad": (w * h + s08) + s08 + 8 * s08,        # source read + 0.8x resample (never stored) + gradient read, fp32 angle + int magnitude write
        "k_lsd_hist": 8 * s08, "k_lsd_scan": 0, "k_lsd_scatter": 4 * s08,    # ordered-list build
        "k_lsd_regions": 5 * s08,                           # region-grow reads (angle + magnitude + used)
        "k_nfa_count": 2 * nln * 100 * 6 * 4,               # angle-map rows under ~2 nln candidate rectangles of a nominal 100 x 6 pixels, 4 B each
        "k_nfa_eval": 0, "k_nfa_accept": 0, "k_nfa_finish": 0,      # rectangle records only
        "k_nfa_all": 0,                                     # round 4: the same stage as one launch in the batch form (its bytes are k_nfa_count's row above; never both in one run)
        "k_keylines": nln * (16 + 68 + 24),
        "k_blur_sobel": w * h + 4 * w * h,                  # fused LBD pre-blur + Sobel: source read, {dx,dy} s16 pair write
        "k_lbd": nln * 63 * 100 * 4 + nln * 100,            # band reads at a nominal 100-px line + descriptor
        "k_search_init": 2 * 32 * nkp + 8 * nkp, "k_knn2_batch": 2 * 32 * nkp + 16 * nkp,
        "k_knn2_expand": 0,                                 # the train rows as int8 matrix-core operands: implementation traffic (8x the descriptors), not algorithmic bytes
        "k_line_match": 2 * 32 * nln + 16 * nln,
        "k_zero_misc": 0,
    }


SIMDS, VALU_CYCLES, CLOCK_HZ = 1024, 4, 2.4e9      # MI355X: 256 CUs x 4 SIMDs, one VALU wave-instruction per SIMD per 4 cycles, 2.4 GHz peak engine clock (MI355X_MICROARCH.md)
CORE_STAGINGS_PER_FRAME = 13054                     # dependent gather round trips of the sequential core per frame (profiles/r05_core_phase_table_B12288.txt)
CORE_RESIDENT_WAVES = 6144                          # 6 waves x 1 024 SIMDs


def other_walls(kernel, avg_launch_ms, B, ms_per_step, workload, hbm_frac):
    """The walls other than HBM bytes that this path can stand against, for the dominant kernel and for the whole step:
      issue          VALU wave-instructions (profiles/sq_instr.json: rocprofv3 --pmc SQ_INSTS_VALU per kernel, per frame) x frames x 4 cycles / (1 024 SIMDs x clock):
                     the time the vector pipes need if they never idle
      random_sector  tools/gather_probe run NOW on this GPU: random 64-B sectors per second with many loads in flight (the fabric's request rate) and as a dependent
                     chain at the core's residency (the round trip one staging of region growing waits for); against the kernel's counted L2 read requests and its
                     dependent stagings
    `bound` = the wall the dominant kernel is closest to.  Counter-derived inputs come from committed profiles (counters need their own rocprofv3 passes)."""
    import subprocess
    issue = None; rs = None
    fr = {"hbm": hbm_frac}
    sq = None
    if workload == "c3":
        try: sq = json.load(open(os.path.join(ROOT, "profiles", "sq_instr.json")))
        except Exception: sq = None
    if sq:
        k = sq["kernels"].get(kernel, {})
        fl = lambda v: v * B * VALU_CYCLES / (SIMDS * CLOCK_HZ) * 1e3
        issue = {"unit": "ms", "clock_ghz": CLOCK_HZ / 1e9, "source": "profiles/sq_instr.json: " + sq.get("source", ""),
                 "kernel": {"valu_wave_instructions_per_frame": k.get("valu_per_frame"), "salu_wave_instructions_per_frame": k.get("salu_per_frame"),
                            "floor_ms_per_launch": fl(k.get("valu_per_frame", 0.0)), "frac": fl(k.get("valu_per_frame", 0.0)) / avg_launch_ms if avg_launch_ms > 0 else None},
                 "step": {"valu_wave_instructions_per_frame": sq["valu_per_frame_total"], "floor_ms_per_step": fl(sq["valu_per_frame_total"]), "ms_per_step": ms_per_step,
                          "frac": fl(sq["valu_per_frame_total"]) / ms_per_step if ms_per_step > 0 else None}}
        fr["valu_issue"] = issue["kernel"]["frac"] or 0.0
    probe = None; probe_src = None
    exe = os.path.join(ROOT, "tools", "gather_probe")
    if os.path.exists(exe):
        try:
            r = subprocess.run([exe, "16", "6"], capture_output=True, text=True, timeout=120)
            probe = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]); probe_src = "tools/gather_probe run after the timed region on this GPU"
        except Exception:
            probe = None
    if probe is None:
        try: probe = json.load(open(os.path.join(ROOT, "profiles", "random_sector.json"))); probe_src = "profiles/random_sector.json (tools/gather_probe on another box: the binary is not built here)"
        except Exception: probe = None
    if probe:
        rs = {"probe": probe, "probe_source": probe_src}
        k = (sq or {}).get("kernels", {}).get(kernel, {})
        if k.get("l2_reads_per_frame") and avg_launch_ms > 0:
            rate = k["l2_reads_per_frame"] * B / (avg_launch_ms * 1e-3) / 1e9
            rs["kernel_l2_read_requests_per_frame"] = k["l2_reads_per_frame"]; rs["kernel_gsectors_per_s"] = rate
            rs["frac_of_request_rate"] = rate / probe["independent"]["gsectors_per_s"]
            fr["random_sector_rate"] = rs["frac_of_request_rate"]
        if kernel.startswith("k_lsd_regions") and avg_launch_ms > 0:
            resident = 4096 if B > 4096 else CORE_RESIDENT_WAVES      # the guest form of the two-stream step: 16 persistent workgroups per compute unit
            rounds = max(1.0, B / resident)
            rt = probe.get("unloaded", probe["dependent"])["round_trip_us"]      # the round trip when nothing queues: a lower bound on what a staging waits
            floor = rounds * CORE_STAGINGS_PER_FRAME * rt * 1e-3
            rs["dependent_chain"] = {"stagings_per_frame": CORE_STAGINGS_PER_FRAME, "round_trip_us": rt, "frames_per_wave_slot": rounds,
                                     "floor_ms_per_launch": floor, "frac": floor / avg_launch_ms,
                                     "note": "every staging of region growing is one dependent gather: a wave cannot issue the next before this one answered; 6 144 waves are resident, so a launch is `frames_per_wave_slot` frames deep"}
            fr["dependent_gather_latency"] = rs["dependent_chain"]["frac"]
    bound = max(fr.items(), key=lambda kv: kv[1])[0]
    note = "fractions of each wall for %s: %s; achieved / peak / frac above stay the HBM figures of the contract (algorithmic bytes over 8 TB/s)" % (kernel, ", ".join("%s %.3f" % kv for kv in sorted(fr.items(), key=lambda kv: -kv[1])))
    return {"bound": bound, "note": note, "issue": issue, "random_sector": rs}


def synth_frames(w, h, U, rank):
    """U distinct seeded scenes per rank (+ their warped previous frames): object count and noise vary from frame to frame, so the LSD core's
    content-dependent work varies too.  Cached in /tmp (generation costs ~0.2 s per 640x480 frame)."""
    import numpy as np
    from synth import synth_frame, warp_prev
    cache = "/tmp/sslam_bench_frames_%dx%d_%d_r%d.npz" % (w, h, U, rank)
    if os.path.exists(cache):
        z = np.load(cache)
        return list(z["cur"]), list(z["prev"])
    area = (w * h) / (640 * 480)
    cur = []
    for i in range(U):
        k = i % 8      # eight kinds of scene: the default density (the round-1 frames), sparser / denser, more noise
        ns, nst, noise = [(60, 40, 2.0), (35, 25, 2.0), (90, 60, 2.0), (60, 40, 3.5), (45, 70, 1.0), (75, 30, 2.5), (60, 40, 2.0), (50, 50, 2.0)][k]
        cur.append(synth_frame(2000 + rank * 1000 + i, w, h, nshapes=int(ns * area), nstrokes=int(nst * area), noise=noise))
    prev = [warp_prev(f) for f in cur]
    try:
        np.savez(cache, cur=np.stack(cur), prev=np.stack(prev))
    except Exception:
        pass
    return cur, prev


def cpu_baseline(frames_cur, frames_prev, budget_s=12.0, max_frames=600, with_lines=True, with_match=True, nref=8):
    """The CPU oracle (a restatement of the reference's CPU path; kind "port") timed on this box's host cores, single thread like the
    reference's front-end (src/Frame.cc:86-87), built here with the reference's flags -O3 -march=native (CMakeLists.txt:10-11)."""
    import numpy as np
    import oracle_lib
    orc = oracle_lib.Oracle(native=True)
    pk, pd = orc.orb_extract(frames_prev[0], NFEAT)      # prime "previous" features outside the timed loop
    pl = orc.lines_extract(frames_prev[0], NLINES) if with_lines else None
    ref = []        # the oracle's outputs for the first distinct frames: compared with the GPU batch after the timing

    def one_frame(i):
        cur = frames_cur[i % len(frames_cur)]
        kp, d = orc.orb_extract(cur, NFEAT)
        kl = ld = None
        if with_lines:
            kl, ld, fn, raw = orc.lines_extract(cur, NLINES)
        if with_match:
            pm = np.stack([pk["x"], pk["y"]], axis=1).astype(np.float32)
            orc.search_for_initialization(pk, pd, kp, d, pm, 100, 0.9, True, (0.0, float(W), 0.0, float(H)))
            orc.knn2(pd, d)
            if with_lines:
                orc.line_match(pl[1], ld, 0.5, False)
        return kp, d, kl, ld

    # SURVEY.md §8(d)'s protocol: 5 warm-up frames, then per-frame times of >= 50 frames (bounded by the budget), median + p10 / p90 beside the mean rate
    WARM = 5
    for i in range(WARM):
        r = one_frame(i)
        if i < min(nref, len(frames_cur)): ref.append(r)
    per = []
    t0 = time.perf_counter()
    n = 0
    while n < max_frames and ((time.perf_counter() - t0) < budget_s or n < 50):
        ta = time.perf_counter()
        r = one_frame(WARM + n)
        per.append(time.perf_counter() - ta)
        if WARM + n < min(nref, len(frames_cur)): ref.append(r)
        n += 1
    dt = time.perf_counter() - t0
    per = np.array(per) * 1e3
    out = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "ms_per_frame": {"median": float(np.median(per)), "p10": float(np.percentile(per, 10)), "p90": float(np.percentile(per, 90)), "mean": float(per.mean())},
           "frames_per_s_median": float(1e3 / np.median(per)), "frames": n, "warmup_frames": WARM,
           "flags": "-O3 -march=native -ffp-contract=off, built on this host" if orc.native else "-O3 generic x86-64 (the native build failed on this host)",
           "sample": "%d frames (after %d warm-up frames) of the same %dx%d workload (ORB %d%s%s), %.1f s, single thread; value = frames / total time, ms_per_frame = per-frame median / p10 / p90 (SURVEY.md 8(d))" % (n, WARM, W, H, NFEAT, " + LSD/LBD %d" % NLINES if with_lines else "", " + SearchForInitialization + dense knn-2 + line match" if with_match else "", dt),
           "note": "the workload's dense 1000x1000 knn-2 (SURVEY.md §8(d) config 3) is part of both sides; the reference itself only runs the windowed search per frame"}
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


def latency_leg(fe, ctx, frames, with_lines, nframes=120):
    """SURVEY §8(d) "GPU timing protocol": per frame H2D -> kernels -> D2H through the synchronous host entry points sslam_orb_extract +
    sslam_lines_extract -- exactly what Frame::ExtractORB / ExtractLSD (src/Frame.cc:150-161) call through the shim, one frame at a time.
    HIP events on the context's stream bracket each call (the call's own copies and kernels run on that stream); wall clock beside it."""
    import numpy as np, torch
    ox = fe.OrbExtractor(ctx, NFEAT); lx = fe.LineExtractor(ctx, NLINES) if with_lines else None
    st = torch.cuda.ExternalStream(ctx.stream)
    for f in frames[:3]:
        ox(f)
        if lx: lx(f)
    ev = []; wall = []
    for i in range(nframes):
        f = frames[i % len(frames)]
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        t0 = time.perf_counter()
        e0.record(st); ox(f); e1.record(st)
        if lx: lx(f)
        e2.record(st)
        wall.append((time.perf_counter() - t0) * 1e3); ev.append((e0, e1, e2))
    torch.cuda.synchronize()
    orb = np.array([a.elapsed_time(b) for a, b, c in ev]); lin = np.array([b.elapsed_time(c) for a, b, c in ev]); wall = np.array(wall)
    ox.close()
    if lx: lx.close()
    pct = lambda a: {"p50": float(np.percentile(a, 50)), "p90": float(np.percentile(a, 90)), "max": float(a.max())}
    out = {"unit": "ms", "frames": nframes, "path": "sslam_orb_extract + sslam_lines_extract per frame (host image in, host results out)",
           "lsd_core": "cluster form (one main wave + helper waves on several compute units; lsd_cluster.h), NFA stage next to it (k_nfa_stream)" if (W * 0.8 <= 2048 and H * 0.8 <= 1024 and os.environ.get("SSLAM_LSD_CLUSTER", "1") != "0" and os.environ.get("SSLAM_LSD_FLAVOUR", "c")[0] == "c") else "lone wave per frame (lsd_regions.h, k_lsd_regions<true>)",
           "orb_extract_hipEvent": pct(orb), "lines_extract_hipEvent": pct(lin) if with_lines else None, "frame_hipEvent": pct(orb + lin), "frame_wall": pct(wall),
           "frames_per_s_one_at_a_time": float(1e3 / np.median(wall))}
    if with_lines and not os.environ.get("SSLAM_LSD_FLAVOUR"):
        # the same call with the sequential core forced to its lone-wave form (what a single frame got before the multi-wave form existed)
        os.environ["SSLAM_LSD_FLAVOUR"] = "lat"
        try:
            lx = fe.LineExtractor(ctx, NLINES); lone = []
            for i in range(3 + min(nframes, 48)):
                f = frames[i % len(frames)]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); lx(f); e1.record(st); torch.cuda.synchronize()
                if i >= 3: lone.append(e0.elapsed_time(e1))
            lx.close()
            out["lines_extract_lone_wave_hipEvent"] = pct(np.array(lone))
        finally:
            del os.environ["SSLAM_LSD_FLAVOUR"]
    return out


def nfa_stream_leg():
    """Child process of the default run (`bench.py --nfa-stream-leg`, prints one JSON object): the one-frame-at-a-time line extraction of the latency leg's 64 frames
    on the default path (round 5: the NFA stage NEXT TO the cluster form of the core, csrc/lsd_nfa.h k_nfa_stream) and with SSLAM_NFA_STREAM=0 (the stage behind the core,
    the round-4 default) -- HIP events around sslam_lines_extract, and every frame's keylines / LBD bytes / line functions compared between the two."""
    import numpy as np, torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pkg
    torch.cuda.set_device(0)
    fe = pkg.frontend(); ctx = fe.Context(0)
    frames, _ = synth_frames(640, 480, 64, 0)
    st = torch.cuda.ExternalStream(ctx.stream)
    pct = lambda a: {"p50": float(np.percentile(a, 50)), "p90": float(np.percentile(a, 90)), "max": float(a.max())}
    res = {"unit": "ms", "frames": 128, "path": "sslam_lines_extract per frame (host image in, host results out), HIP events"}
    outs = {}
    for mode in ("default", "nfa_behind_core"):
        os.environ.pop("SSLAM_NFA_STREAM", None)
        if mode == "nfa_behind_core": os.environ["SSLAM_NFA_STREAM"] = "0"
        lx = fe.LineExtractor(ctx, 200)
        for f in frames[:4]: lx(f)
        t = []; outs[mode] = []
        for i in range(128):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); r = lx(frames[i % 64]); e1.record(st); torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1))
            if i < 64: outs[mode].append(tuple(np.ascontiguousarray(a).tobytes() for a in r))
        lx.close()
        res[mode] = pct(np.array(t))
    os.environ.pop("SSLAM_NFA_STREAM", None)
    res["frames_differing_from_default_path"] = int(sum(a != b for a, b in zip(outs["default"], outs["nfa_behind_core"])))
    print(json.dumps(res))


def pcie_leg(fe, ctx, frames, with_lines, n=18432, chunk=0, prev_frames=None):
    """sslam_frontend_batch: host images in, host records out (extract only: the API of SURVEY §8(b)).  n covers three chunks of the default
    size (6144 frames = every wave slot of the sequential LSD core), so that uploads, kernels and downloads of neighbouring chunks overlap;
    the result arrays exist before the timed call (a caller streams into its own buffers)."""
    import numpy as np, torch
    ox = fe.OrbExtractor(ctx, NFEAT); lx = fe.LineExtractor(ctx, NLINES) if with_lines else None
    U = len(frames)
    if prev_frames is not None:      # a sequence: previous frame, current frame, previous frame, ... so that the match stage has something to match
        host = np.stack([(prev_frames if i % 2 == 0 else frames)[(i // 2) % U] for i in range(min(n, 4 * U))])
    else:
        host = np.stack([frames[i % U] for i in range(min(n, 4 * U))])
    host = np.ascontiguousarray(np.tile(host, ((n + len(host) - 1) // len(host), 1, 1))[:n])
    out = {"entry": "sslam_frontend_batch", "frames": n, "chunk": chunk or "default (6144)",
           "note": "extract only (ORB + LSD/LBD); H2D of chunk k+1 and D2H of chunk k-1 under the kernels of chunk k, point and line branch on two streams inside the library; result arrays allocated before the timed call"}
    lcap = NLINES
    warm = n      # one whole untimed call first: BOTH slots of the bounce ring (device buffers, pinned staging) exist before the timed call (a warm-up of one chunk left the second slot's allocation inside it)
    res_pg = fe.frontend_batch_alloc(n, ox.cap, lcap, pinned=False)
    fe.frontend_batch_raw(ox, lx, host[:warm], tuple(a[:warm] for a in res_pg), chunk=chunk)          # workspace / staging allocation outside the timing
    t0 = time.perf_counter(); fe.frontend_batch_raw(ox, lx, host, res_pg, chunk=chunk); out["pageable_frames_per_s"] = n / (time.perf_counter() - t0)
    pin = torch.empty(host.shape, dtype=torch.uint8, pin_memory=True); pin.numpy()[:] = host
    res_pin = fe.frontend_batch_alloc(n, ox.cap, lcap, pinned=True)
    fe.frontend_batch_raw(ox, lx, pin.numpy()[:warm], tuple(a[:warm] for a in res_pin), chunk=chunk)
    t0 = time.perf_counter(); fe.frontend_batch_raw(ox, lx, pin.numpy(), res_pin, chunk=chunk); out["pinned_frames_per_s"] = n / (time.perf_counter() - t0)
    out["results_equal"] = bool(np.array_equal(res_pg[2], res_pin[2]) and np.array_equal(res_pg[6], res_pin[6]))
    # the same with the match stage (sslam_frontend_batch_match: frame i against frame i-1 -- SearchForInitialization + dense 2-NN + line matcher,
    # the three matchers of the resident-frame step), i.e. BASELINE configs[2] end to end through host buffers
    m_pg = fe.frontend_batch_match_alloc(n, ox.cap, lcap, pinned=False)
    fe.frontend_batch_match_raw(ox, lx, host[:warm], tuple(a[:warm] for a in res_pg), tuple(a[:warm] for a in m_pg), chunk=chunk)
    t0 = time.perf_counter(); fe.frontend_batch_match_raw(ox, lx, host, res_pg, m_pg, chunk=chunk); out["pageable_with_match_frames_per_s"] = n / (time.perf_counter() - t0)
    m_pin = fe.frontend_batch_match_alloc(n, ox.cap, lcap, pinned=True)
    fe.frontend_batch_match_raw(ox, lx, pin.numpy()[:warm], tuple(a[:warm] for a in res_pin), tuple(a[:warm] for a in m_pin), chunk=chunk)
    t0 = time.perf_counter(); fe.frontend_batch_match_raw(ox, lx, pin.numpy(), res_pin, m_pin, chunk=chunk); out["pinned_with_match_frames_per_s"] = n / (time.perf_counter() - t0)
    out["match_results_equal"] = bool(np.array_equal(m_pg[1], m_pin[1]) and np.array_equal(m_pg[5], m_pin[5]))
    out["match_entry"] = "sslam_frontend_batch_match (SearchForInitialization window 100 + dense Hamming 2-NN + LSD line matcher against the previous frame of the sequence)"
    out["matches_per_frame"] = float(m_pin[1][1::2].mean()) if prev_frames is not None else float(m_pin[1][1:].mean())      # (odd frames: a frame against its own predecessor)
    ox.close()
    if lx: lx.close()
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks here, one per GPU, under torch.distributed.run on 127.0.0.1 (what the driver's own command
    line does), hand their output through -- rank 0 prints the ONE JSON line -- and return the launcher's exit code: non-zero only when a rank really failed."""
    import subprocess, socket, ctypes
    n = args.gpus
    try:      # a clear message instead of N tracebacks when the node has fewer GPUs than asked for
        hip = ctypes.CDLL("libamdhip64.so"); cnt = ctypes.c_int(0)
        if hip.hipGetDeviceCount(ctypes.byref(cnt)) != 0: cnt.value = 0
        if cnt.value < n:
            print("bench.py: --gpus %d but %d GPU(s) visible%s" % (n, cnt.value, "; this framework has no CPU fallback" if cnt.value == 0 else ""), file=sys.stderr)
            return 2
    except OSError:
        pass
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    argv = [a for a in sys.argv[1:] if a != "--self-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["SSLAM_BENCH_SELF_LAUNCHED"] = "1"
    if n == 1:
        env.setdefault("SSLAM_FORCE_COLLECTIVE", "1")      # one rank: the exchange step still runs through RCCL (ncclSend / ncclRecv to itself)
    print("bench.py: launching %d rank(s): %s" % (n, " ".join(cmd)), file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (default: the workload's)")
    ap.add_argument("--unique", type=int, default=0, help="distinct synthetic frames per rank, tiled to the batch (default: the workload's, 64 for c3)")
    ap.add_argument("--no-overlap", action="store_true", help="run the point branch (ORB extract + match) and the line branch (LSD/LBD extract + match) one after the other on one stream instead of on two HIP streams (default: two streams -- the sequential LSD core is latency-bound and leaves issue slots that the VALU-bound ORB kernels fill)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the latency and PCIe-inclusive legs")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event per-kernel timing (used for rocprofv3 runs)")
    ap.add_argument("--no-pcie", action="store_true", help="latency leg only (no PCIe-inclusive host-batch leg)")
    ap.add_argument("--nfa-stream-leg", action="store_true", help="(child process of the default run) single-frame line extraction with SSLAM_NFA_STREAM=0 (the NFA stage behind the core) beside the default path")
    ap.add_argument("--lsd-nfa-variant", type=int, default=-1, help="decision D11 (sslam_lines_set_nfa_variant): 0 = log_gamma(n + 1), the default of rounds 1-4; 1 = (double(n) + 1), the default since round 5; -1: the library's default")
    ap.add_argument("--self-launch", action="store_true", help="start the ranks from this process even for --gpus 1 (what --gpus N > 1 does by itself when no launcher set RANK): the N = 1 test of the multi-rank path, RCCL exchange included")
    ap.add_argument("--no-other-workloads", action="store_true", help="the default run appends a short pass of BASELINE configs[3] (1280x960 / 2000 kp / 400 lines) as other_workloads.c4; this skips it")
    args = ap.parse_args()
    if args.nfa_stream_leg:
        nfa_stream_leg()
        return
    if "RANK" not in os.environ and (args.gpus > 1 or args.self_launch):
        sys.exit(self_launch(args))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this host driver (already exported on the GPU boxes)
    import numpy as np
    import torch
    import pkg
    global W, H, NFEAT, NLINES
    wl = WORKLOADS[args.workload]
    W, H, NFEAT, NLINES = wl["w"], wl["h"], wl["nfeat"], max(wl["nlines"], 1)
    with_lines = wl["nlines"] > 0

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        print("bench.py: --gpus %d under a launcher that set WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; this framework has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    if args.batch <= 0:
        args.batch = max(1, wl["batch"] // world) if args.workload == "c5" else wl["batch"]
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    dist = None
    # SSLAM_FORCE_COLLECTIVE=1 (tests): run the RCCL exchange step even with a single rank, so the collective path is exercised on a 1-GPU box
    force_collective = os.environ.get("SSLAM_FORCE_COLLECTIVE") == "1" and "RANK" in os.environ
    if world > 1 or force_collective:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))      # bootstrap, barrier and the max-over-ranks timing only

    fe = pkg.frontend()
    pipeline = pkg._load("sslam_pipeline", os.path.join(pkg.PKG_DIR, "pipeline.py"))
    sharding = pkg._load("sslam_sharding", os.path.join(pkg.PKG_DIR, "sharding.py"))
    ctx = fe.Context(local_rank)
    B = args.batch
    U = max(1, min(args.unique if args.unique > 0 else wl["unique"], B))
    cur_np, prev_np = synth_frames(W, H, U, rank)
    reps = (B + U - 1) // U
    cur = torch.from_numpy(np.stack(cur_np)).to(dev).repeat(reps, 1, 1)[:B].contiguous()
    prev = torch.from_numpy(np.stack(prev_np)).to(dev).repeat(reps, 1, 1)[:B].contiguous()
    host_cur = None
    if wl.get("h2d"):
        host_cur = torch.empty((B, H, W), dtype=torch.uint8, pin_memory=True); host_cur.copy_(cur)

    pipe = pipeline.FrontendBatch(fe, ctx, W, H, B, NFEAT, NLINES, dev, with_lines=with_lines, with_match=wl["match"])
    if args.lsd_nfa_variant >= 0 and pipe.lines is not None: pipe.lines.set_nfa_variant(args.lsd_nfa_variant)
    pipe.extract(prev, "prev")      # previous-frame features: extracted once, resident (the stream's t-1 state)
    torch.cuda.synchronize()

    gather = None
    if dist is not None:
        # the library's own communicator (RCCL through the C ABI): rank 0 makes the id, torch.distributed only carries it to the others
        uid = torch.from_numpy(fe.Group.unique_id()).to(dev) if rank == 0 else torch.zeros(128, dtype=torch.uint8, device=dev)
        dist.broadcast(uid, src=0)
        if force_collective and world == 1:
            os.environ["SSLAM_GROUP_SELF_SENDRECV"] = "1"      # one rank: route its own records through ncclSend / ncclRecv so that RCCL moves the bytes
        # every rank must end up on the same exchange path: the library's communicator if ALL ranks got one and one trial gather went
        # through everywhere, else torch.distributed (also RCCL) -- a measurement run must not die on the communicator bootstrap
        group = None; gather_impl = "sslam_group_gather_dev"; why = ""
        def agreed(ok):      # every rank learns whether ALL ranks succeeded
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev); dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1
        ok = True
        try:
            if os.environ.get("SSLAM_BENCH_FAIL_GROUP"):      # test knob: exercise the fallback
                raise RuntimeError("SSLAM_BENCH_FAIL_GROUP")
            group = fe.Group(device=local_rank, rank=rank, nranks=world, uid=uid.cpu().numpy())
            gather = sharding.GroupGather(fe, ctx, group, pipe, dev)
        except Exception as e:      # noqa: BLE001
            ok = False; why = str(e)[:160]
        ok = agreed(ok)
        if ok:                       # one trial exchange, agreed upon again, before anything is timed
            try:
                pipe.step(cur, overlap=not args.no_overlap); gather.submit(); gather.wait()
            except Exception as e:      # noqa: BLE001
                ok = False; why = str(e)[:160]
            ok = agreed(ok)
        if not ok:
            print("bench.py: rank %d falls back to torch.distributed for the gather (%s)" % (rank, why or "another rank failed"), file=sys.stderr)
            gather = sharding.TorchGather(fe, ctx, dist, pipe, dev, world, rank); group = None
            gather_impl = "torch.distributed gather (backend nccl = RCCL); library communicator unavailable: " + (why or "on another rank")

    def one_step():
        if host_cur is not None:
            cur.copy_(host_cur, non_blocking=True)             # configs[4]: the H2D of the frames is part of the step
        pipe.step(cur, overlap=not args.no_overlap)
        if gather is not None:      # the one exchange step of the path: compacted per-frame records to rank 0 over RCCL,
            gather.submit()         # overlapped with the next step's kernels (configs[4]: waited for inside the step)
            if wl.get("sync_gather"):
                gather.wait()

    for _ in range(args.warmup):
        one_step()
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    if not args.no_profile:
        fe.lib().sslam_profile_enable(ctx.h, 1)
    if gather is not None:
        gather.wait_s = 0.0; gather.waits = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fe.lib().sslam_profile_enable(ctx.h, 0)
    prof = pipeline.profile_drain(fe, ctx) if not args.no_profile else {}
    per_rank = None
    if dist is not None:
        # every rank's own wall time and how long it stood still for the exchange: rank 0 prints them (a slow rank or an exchange that is
        # not hidden behind the next step shows here, not in the max-over-ranks figure)
        mine = torch.tensor([dt, gather.wait_s if gather is not None else 0.0], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "wall_s": float(a[0].item()), "frames_per_s": B * args.steps / float(a[0].item()), "gather_wait_ms_per_step": float(a[1].item()) * 1e3 / max(args.steps, 1)}
                    for r, a in enumerate(allr)]
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # the dominant kernel alone: two more steps on ONE stream (kernels back to back), so that the roofline block also carries a duration that
    # is not stretched by the other branch's kernels sharing the chip (outside the timed region; rank 0's own GPU)
    prof_iso = {}
    if not args.no_profile and not args.no_overlap and gather is None:
        fe.lib().sslam_profile_enable(ctx.h, 1)
        for _ in range(2):
            pipe.step(cur, overlap=False)
        torch.cuda.synchronize()
        fe.lib().sslam_profile_enable(ctx.h, 0)
        prof_iso = pipeline.profile_drain(fe, ctx)

    gather_info = None
    if rank == 0 and gather is not None:
        # what RCCL delivered for the last step: rank 0's own records must be in it byte for byte (global frame i = local i // world on rank i % world);
        # reported in the JSON line (config.gather_check), never fatal for the measurement
        try:
            got, sizes = gather.result()
            mine = pipe.packed_stream()
            rk, rd, rn, rl, rld, rfn, rnl, nrec = fe.unpack_records(got.cpu().numpy(), B * world, pipe.cap, pipe.lcap, with_lines)
            ok = nrec == B * world and sizes[0] == mine.numel()
            c = pipe.feat["cur"]
            n_own = c["n"].cpu().numpy()
            ok = ok and bool((rn[0::world] == n_own).all()) and bool(np.array_equal(rd[0::world][0, :n_own[0]], c["desc"][0, :n_own[0]].cpu().numpy()))
            if with_lines:
                nl_own = c["nl"].cpu().numpy()
                ok = ok and bool((rnl[0::world] == nl_own).all()) and bool(np.array_equal(rfn[0::world][-1, :nl_own[-1]], c["linefn"][-1, :nl_own[-1]].cpu().numpy()))
            gather_info = {"ok": bool(ok), "impl": gather_impl, "rccl_ranks": int(group.size) if group is not None else int(dist.get_world_size()), "self_launched": os.environ.get("SSLAM_BENCH_SELF_LAUNCHED") == "1",
                           "records": int(nrec), "bytes_per_rank": sizes, "bytes_per_frame": float(sum(sizes)) / max(nrec, 1),
                           "per_rank": per_rank, "note": "gather_wait_ms_per_step = time the pipeline waited for the previous step's exchange (0: hidden behind the kernels)"}
        except Exception as e:
            gather_info = {"ok": False, "impl": gather_impl, "error": str(e)[:200], "per_rank": per_rank}
    if rank == 0:
        counts = pipe.feat["cur"]["n"].cpu().numpy(); lcounts = pipe.feat["cur"]["nl"].cpu().numpy()
        nm = pipe.nmatch.cpu().numpy(); nlp = pipe.nlpairs.cpu().numpy()
        total_frames = B * world * args.steps
        fps = total_frames / dt
        stat = lambda a: {"min": int(a[:U].min()), "mean": float(a[:U].mean()), "max": int(a[:U].max())}
        out = {
            "metric": "front-end frames/sec (ORB+LSD extract+match) 640x480 @1000kp/200ln" if args.workload == "c3" else "front-end frames/sec, workload " + args.workload,
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if args.workload != "c5" else "strong", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl["name"],
                       "batch_per_gpu": B, "global_batch": B * world, "unique_frames": U,
                       "keypoints_per_frame": stat(counts), "lines_per_frame": stat(lcounts),
                       "mean_keypoints": float(counts.mean()), "mean_lines": float(lcounts.mean()),
                       "mean_orb_matches": float(nm.mean()), "mean_line_matches": float(nlp.mean()),
                       "line_decisions": {"lsd_nfa_variant": (1 if args.lsd_nfa_variant < 0 else args.lsd_nfa_variant), "lbd_bit_order": 1, "lsd_resize": 0, "seed_order": 0,
                                          "note": "include/sslam_frontend.h sslam_lines_set_*: D11 / D12 default to the OpenCV-as-recalled forms since round 5 (rounds 1-4: 0 / 0; other_workloads.c3_lsd_nfa_variant_0 is the step under the old D11)"},
                       "streams": "point branch and line branch on two HIP streams" if not args.no_overlap else "one stream",
                       "parallelism": ("frames sharded %d/GPU, RCCL gather (sslam_group_gather_dev) of compacted records to rank 0 per step" % B) if gather is not None else "single GPU",
                       "gather_check": gather_info["ok"] if gather_info else None, "gather": gather_info},
        }
        if prof:
            ab = alg_bytes_per_frame(W, H, NFEAT, NLINES if with_lines else 0)
            dom = max(prof.items(), key=lambda kv: kv[1][0])
            name, (ms, launches) = dom
            avg_s = ms / launches * 1e-3
            bytes_per_launch = ab.get(name, 0) * B
            ach = bytes_per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
            traffic, traffic_src = None, None
            try:      # HBM bytes per launch from the committed PMC passes (separate rocprofv3 --pmc runs cannot share a process with this timing run), scaled to this batch
                pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_c4.json" if args.workload == "c4" else "pmc_traffic.json")))
                pt = pj["kernels"][name]      # fetch side x2: gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, calibrated in profiles/)
                # lines.hip's dispatch rule: cluster form, lone waves, throughput form (six waves per SIMD), guest form (four, persistent: the two-stream step from two rounds of its grid on)
                core_now = ("k_lsd_regions_cl" if B <= 64 else "k_lsd_regions<true, 4>" if B < 1024 else "k_lsd_regions<false, 4>, one workgroup per frame" if B <= 4096 else
                            "k_lsd_regions<false, 4>, guest form (persistent grid)" if (B >= 8192 and not args.no_overlap) else "k_lsd_regions<false, 6>")      # (tools/make_pmc_traffic.py lsd_core: the same rule)
                same_form = pj.get("lsd_core") == core_now or (args.workload == "c4" and pj.get("lsd_core", "").startswith("k_lsd_regions<false, 4>, one workgroup"))      # (c4: counted at 1 024 frames, timed at 3 072: the same instantiation)
                if "fetch_correction" not in pj or not same_form:
                    # the counted kernel must be the instantiation this run times, and the gfx950 fetch correction must be on record in the file
                    traffic_src = "refused: the PMC file counted %s at batch %s (fetch_correction %s), this run times %s" % (pj.get("lsd_core"), pj.get("batch"), pj.get("fetch_correction"), core_now)
                else:
                    traffic = (pt["fetch_bytes_per_frame"] * pj["fetch_correction"] + pt["write_bytes_per_frame"]) * B
                    traffic_src = "profiles/%s: fetch x %.1f + write per frame x this batch (%s)" % ("pmc_traffic_c4.json" if args.workload == "c4" else "pmc_traffic.json", pj["fetch_correction"], pj.get("source", ""))
            except Exception:
                pass
            sv = survey_bytes_per_frame(W, H, NFEAT, NLINES if with_lines else 0)
            per_gpu_fps = fps / world
            walls = other_walls(name, ms / launches, B, dt / args.steps * 1e3, args.workload, ach / HBM_PEAK_GBS)
            out["roofline"] = {"kernel": name, "bound": walls["bound"], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                               "bound_note": walls["note"], "issue": walls["issue"], "random_sector": walls["random_sector"],
                               "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": ms / launches, "launches": launches,
                               "alg_bytes_per_launch": bytes_per_launch,
                               "isolated": ({"avg_launch_ms": prof_iso[name][0] / prof_iso[name][1], "frac": bytes_per_launch / (prof_iso[name][0] / prof_iso[name][1] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "note": "the same kernel in two extra steps on one stream after the timed region: no other kernel shares the chip"} if name in prof_iso else None),
                               "whole_pipeline": {"survey_8d_bytes_per_frame": sv, "achieved": sv * per_gpu_fps / 1e9, "frac": sv * per_gpu_fps / 1e9 / HBM_PEAK_GBS,
                                                  "per_kernel_table_bytes_per_frame": sum(ab.values()), "per_kernel_table_frac": sum(ab.values()) * per_gpu_fps / 1e9 / HBM_PEAK_GBS,
                                                  "note": "frac uses SURVEY.md §8(d)'s own byte total; the per-kernel table (DESIGN.md §4) is what this implementation moves"},
                               "kernels_ms_per_step": {k: v[0] / args.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])},
                               "kernels_note": "HIP-event durations per kernel; with the two branches on two streams kernels of different branches overlap, so the durations include the sharing and their sum exceeds ms_per_step (run with --no-overlap for isolated durations)" if not args.no_overlap else "kernels run back to back on one stream"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], ref = cpu_baseline(cur_np, prev_np, with_lines=with_lines, with_match=wl["match"])
            out["cpu_baseline"]["parity_vs_gpu"] = parity_vs_gpu(ref, pipe.feat["cur"], with_lines)
        if world == 1 and not args.no_extras and args.workload in ("c3", "c2", "c4"):
            # the extra legs build their own extractors: the resident batch (175 GB at the default size) is released first
            pipe.close(); pipe = None
            del cur, prev
            torch.cuda.empty_cache()
            try:
                out["latency"] = latency_leg(fe, ctx, cur_np, with_lines)
                if not args.no_pcie:
                    out["pcie_inclusive"] = pcie_leg(fe, ctx, cur_np, with_lines, n=18432 if W == 640 else 3072, chunk=0 if W == 640 else 1024, prev_frames=prev_np)
            except Exception as e:
                out["latency_error"] = str(e)[:300]
        if world == 1 and args.workload == "c3" and not args.no_extras and not args.no_other_workloads:
            # BASELINE configs[3] (1280x960, 2000 kp, 400 lines: the "rocprof HBM-GB/s run") in the driver's own bench run: a short pass of the
            # same harness in a child process (this process holds no batch any more), its headline numbers appended to this line
            import subprocess
            try:
                child_env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                                               "SSLAM_FORCE_COLLECTIVE", "TORCHELASTIC_RUN_ID")}      # a plain single-process run, whatever launched this one
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "c4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-pcie"],
                                   capture_output=True, text=True, timeout=900, env=child_env)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                c4 = json.loads(line[-1])
                out["other_workloads"] = {"c4": {"workload": c4["config"]["workload"], "value": c4["value"], "unit": c4["unit"], "ms_per_step": c4["ms_per_step"], "steps": c4["steps"],
                                                 "batch": c4["config"].get("batch_per_gpu"), "mean_keypoints": c4["config"].get("mean_keypoints"), "mean_lines": c4["config"].get("mean_lines"),
                                                 "roofline": {k: c4["roofline"].get(k) for k in ("kernel", "frac", "achieved", "avg_launch_ms", "traffic")} if "roofline" in c4 else None,
                                                 "whole_pipeline_frac": c4.get("roofline", {}).get("whole_pipeline", {}).get("frac"),
                                                 "latency": {k: c4.get("latency", {}).get(k) for k in ("orb_extract_hipEvent", "lines_extract_hipEvent", "frames_per_s_one_at_a_time")}}}
            except Exception as e:
                out["other_workloads"] = {"c4": {"error": str(e)[:300]}}
            try:      # trend continuity: the same step under decision D11's variant 0 (nfa()'s first term = log_gamma(n + 1): the default of rounds 1-4, whose bench lines r01-r04 are)
                if args.lsd_nfa_variant < 0:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extras", "--no-other-workloads", "--lsd-nfa-variant", "0"],
                                       capture_output=True, text=True, timeout=600, env=child_env)
                    v0 = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                    out["other_workloads"]["c3_lsd_nfa_variant_0"] = {
                        "workload": "the headline step with sslam_lines_set_nfa_variant(0) -- log_gamma(n + 1), the default of rounds 1-4 (BENCH_r01..r04 were measured on it); the headline's default, variant 1, accepts 2.0-2.7 x the segments, so its NFA stage is shorter and its frames hold more lines",
                        "value": v0["value"], "unit": v0["unit"], "ms_per_step": v0["ms_per_step"], "steps": v0["steps"], "mean_lines": v0["config"].get("mean_lines"), "mean_line_matches": v0["config"].get("mean_line_matches"),
                        "kernels_ms_per_step": {k: v0.get("roofline", {}).get("kernels_ms_per_step", {}).get(k) for k in ("k_lsd_regions", "k_nfa_all", "k_lbd", "k_keylines", "k_line_match")}}
            except Exception as e:
                out.setdefault("other_workloads", {})["c3_lsd_nfa_variant_0"] = {"error": str(e)[:300]}
            try:      # the alternative of the single-frame default, in a child process of its own (a fault there cannot touch this line): the NFA stage behind the core instead of next to it
                import subprocess
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--nfa-stream-leg"], capture_output=True, text=True, timeout=240,
                                   env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
                out["latency_nfa_behind_core"] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:
                out["latency_nfa_behind_core"] = {"error": str(e)[:300]}
        print(json.dumps(out))
    if gather is not None:
        gather.wait()
        if group is not None:
            group.close()
    if pipe is not None:
        pipe.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
