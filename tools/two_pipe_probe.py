"""What-if: the batch as P independent pipelines (own extractor handles, own pair of streams) stepping without joins, so that the sequential
LSD core of one sub-batch runs under the VALU-bound kernels of the others.  python tools/two_pipe_probe.py <pipes> <frames per pipe> [steps]"""
import sys, os, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch, pkg, bench
P = int(sys.argv[1]); B = int(sys.argv[2]); K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
torch.cuda.set_device(0)
fe = pkg.frontend(); ctx = fe.Context(0)
pipeline = pkg._load("sslam_pipeline", os.path.join(pkg.PKG_DIR, "pipeline.py"))
cur_np, prev_np = bench.synth_frames(640, 480, 64, 0)
cur = torch.from_numpy(np.stack(cur_np)).cuda().repeat(B // 64, 1, 1).contiguous()
prev = torch.from_numpy(np.stack(prev_np)).cuda().repeat(B // 64, 1, 1).contiguous()
pipes = [pipeline.FrontendBatch(fe, ctx, 640, 480, B, 1000, 200, "cuda:0") for _ in range(P)]
for p in pipes: p.extract(prev, "prev")
torch.cuda.synchronize()
def run(k):
    for _ in range(k):
        for p in pipes: p.step(cur, overlap=True, join=False)
run(1); torch.cuda.synchronize()
t0 = time.perf_counter(); run(K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("pipes", P, "x", B, "frames:", round(P * B * K / dt), "frames/s", round(dt / K * 1e3, 1), "ms per round; lines", float(pipes[0].feat["cur"]["nl"].float().mean()))
