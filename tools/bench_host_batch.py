"""PCIe-inclusive batch rate: frames in HOST memory through sslam_frontend_batch (H2D + ORB + LSD/LBD + D2H, chunks overlapped, two branch
streams), results back in host arrays that exist before the timed call.  usage: python tools/bench_host_batch.py [n] [chunk] [one-stream 0/1]"""
import sys, os, time; sys.path.insert(0, 'tests')
import numpy as np, torch, pkg
torch.cuda.init()          # before the library's own context: torch's lazy initialisation fails after it on this stack
from synth import synth_frame
n = int(sys.argv[1]) if len(sys.argv) > 1 else 18432
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if len(sys.argv) > 3 and sys.argv[3] == "1": os.environ["SSLAM_BATCH_ONE_STREAM"] = "1"
fe = pkg.frontend(); ctx = fe.Context(0)
orb = fe.OrbExtractor(ctx, 1000); lines = fe.LineExtractor(ctx, 200)
if os.environ.get("HOST_BATCH_BENCH_FRAMES"):      # the bench's own sequence: 64 scenes of varied density, previous / current frame alternating (bench.py pcie_leg)
    sys.path.insert(0, '.'); import bench
    cur, prev = bench.synth_frames(640, 480, 64, 0)
    u = np.stack([(prev if i % 2 == 0 else cur)[(i // 2) % 64] for i in range(256)])
else:
    u = np.stack([synth_frame(2000 + i) for i in range(8)])
frames = np.ascontiguousarray(np.tile(u, ((n + len(u) - 1) // len(u), 1, 1))[:n])
pin = torch.empty(frames.shape, dtype=torch.uint8, pin_memory=True); pin.numpy()[:] = frames
outs = {False: fe.frontend_batch_alloc(n, orb.cap, 200, pinned=False), True: fe.frontend_batch_alloc(n, orb.cap, 200, pinned=True)}
fe.frontend_batch_raw(orb, lines, frames[:min(n, 6144)], tuple(a[:min(n, 6144)] for a in outs[False]), chunk=chunk)                # warm-up: allocations, first-touch
for name, src, pinned in (("pageable", frames, False), ("pinned", pin.numpy(), True)):
    for rep in range(2):
        t0 = time.perf_counter(); out = fe.frontend_batch_raw(orb, lines, src, outs[pinned], chunk=chunk); dt = time.perf_counter() - t0
    print("sslam_frontend_batch: %d frames 640x480, %s host memory in and out, chunk %s: %.1f ms, %.0f frames/s PCIe-inclusive; frame 5: %d keypoints, %d lines"
          % (n, name, chunk or "default", dt * 1e3, n / dt, out[2][5], out[6][5]), flush=True)
a, b = outs[False], outs[True]
assert np.array_equal(a[2], b[2]) and np.array_equal(a[6], b[6]) and np.array_equal(a[1][:16, :900], b[1][:16, :900]), "pinned path differs from the pageable one"
ref = fe.frontend_batch(orb, lines, frames[:16], chunk=8)
assert all(np.array_equal(r[1], a[1][i, :len(r[1])]) and np.array_equal(r[3], a[4][i, :len(r[3])]) for i, r in enumerate(ref)), "chunked reference differs"
print("results equal across pageable / pinned / small-chunk calls")
