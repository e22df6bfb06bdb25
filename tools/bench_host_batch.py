"""PCIe-inclusive batch rate: n frames in host memory through sslam_frontend_batch (pinned staging, H2D / kernels / D2H overlapped),
results back in host arrays.  usage: python tools/bench_host_batch.py [n] [chunk]"""
import sys, time; sys.path.insert(0, 'tests')
import numpy as np, torch, pkg
torch.cuda.init()          # before the library's own context: torch's lazy initialisation fails after it on this stack
from synth import synth_frame
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
fe = pkg.frontend(); ctx = fe.Context(0)
orb = fe.OrbExtractor(ctx, 1000); lines = fe.LineExtractor(ctx, 200)
u = np.stack([synth_frame(2000 + i) for i in range(8)])
frames = np.ascontiguousarray(np.tile(u, (n // 8, 1, 1)))
pin = torch.empty(frames.shape, dtype=torch.uint8, pin_memory=True); pin.numpy()[:] = frames
fe.frontend_batch(orb, lines, frames[:chunk], chunk=chunk)                # warm-up: allocations, first-touch
for name, src, pinned in (("pageable", frames, False), ("pinned", pin.numpy(), True)):
    for rep in range(2):
        t0 = time.perf_counter(); out = fe.frontend_batch(orb, lines, src, chunk=chunk, pinned=pinned); dt = time.perf_counter() - t0
    print("sslam_frontend_batch: %d frames 640x480, %s host memory in and out, chunk %d: %.1f ms, %.0f frames/s PCIe-inclusive; frame 5: %d keypoints, %d lines"
          % (n, name, chunk, dt * 1e3, n / dt, len(out[5][0]), len(out[5][2])))
ref = fe.frontend_batch(orb, lines, frames[:16], chunk=8)
assert all(np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3]) for a, b in zip(ref, out[:16])), "pinned path differs"
