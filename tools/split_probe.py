"""What-if: the batch as S sub-batches, each on its own HIP stream and free-running over K steps, so that the issue-bound
LSD core of one sub-batch overlaps the streaming kernels of the others.  Prints frames/s per configuration.
usage: python tools/split_probe.py B S [K] [stagger]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import pkg
from synth import synth_frame, warp_prev

B, S = int(sys.argv[1]), int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 3
stagger = int(sys.argv[4]) if len(sys.argv) > 4 else 1
fe = pkg.frontend(); pipeline = pkg._load("sslam_pipeline", os.path.join(pkg.PKG_DIR, "pipeline.py"))
ctx = fe.Context(0); dev = "cuda:0"
U = 8
cur_np = [synth_frame(2000 + i, 640, 480) for i in range(U)]; prev_np = [warp_prev(f) for f in cur_np]
sub = B // S
cur = torch.from_numpy(np.stack(cur_np)).to(dev).repeat((sub + U - 1) // U, 1, 1)[:sub].contiguous()
prev = torch.from_numpy(np.stack(prev_np)).to(dev).repeat((sub + U - 1) // U, 1, 1)[:sub].contiguous()
pipes = [pipeline.FrontendBatch(fe, ctx, 640, 480, sub, 1000, 200, dev) for _ in range(S)]
outer = [torch.cuda.Stream(dev) for _ in range(S)]
for p in pipes:
    p.extract(prev, "prev")
torch.cuda.synchronize()

def run(k):
    main = torch.cuda.current_stream()
    for i, (p, sx) in enumerate(zip(pipes, outer)):
        sx.wait_stream(main)
    for _ in range(k):
        for i, (p, sx) in enumerate(zip(pipes, outer)):
            with torch.cuda.stream(sx):
                p.step(cur, lines_first=bool(stagger and (i & 1)))
    for sx in outer:
        main.wait_stream(sx)

run(1); torch.cuda.synchronize()
t0 = time.perf_counter(); run(K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("B=%d split=%d stagger=%d: %.0f frames/s  (%.1f ms per %d frames)" % (B, S, stagger, sub * S * K / dt, dt / K * 1e3, sub * S), flush=True)
n0 = pipes[0].feat["cur"]["n"].cpu().numpy(); n1 = pipes[-1].feat["cur"]["n"].cpu().numpy()
assert (n0 == n1).all()
for p in pipes: p.close()
ctx.close()
