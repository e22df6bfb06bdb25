"""What-if: the batch as P pipelines (own extractor handles, own pair of streams) that take turns on the sequential LSD core.
   python tools/two_pipe_probe.py <pipes> <frames per pipe> [steps] [gate: 1 = the cores take turns (sslam_lines_set_core_gate), 0 = free running]
Free running (round 2, tools/split_probe.py): the halves' cores overlap each other and lose.  Gated: one half's VALU-bound kernels behind its
core meet the other half's bandwidth-bound kernels in front of its core."""
import sys, os, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch, pkg, bench
P = int(sys.argv[1]); B = int(sys.argv[2]); K = int(sys.argv[3]) if len(sys.argv) > 3 else 4; GATE = int(sys.argv[4]) if len(sys.argv) > 4 else 1
torch.cuda.set_device(0)
fe = pkg.frontend(); ctx = fe.Context(0)
pipeline = pkg._load("sslam_pipeline", os.path.join(pkg.PKG_DIR, "pipeline.py"))
cur_np, prev_np = bench.synth_frames(640, 480, 64, 0)
cur = torch.from_numpy(np.stack(cur_np)).cuda().repeat(B // 64, 1, 1).contiguous()
prev = torch.from_numpy(np.stack(prev_np)).cuda().repeat(B // 64, 1, 1).contiguous()
pipes = [pipeline.FrontendBatch(fe, ctx, 640, 480, B, 1000, 200, "cuda:0") for _ in range(P)]
for p in pipes: p.extract(prev, "prev")
torch.cuda.synchronize()
done = [torch.cuda.Event() for _ in range(P)]
for e in done: e.record()          # (recording creates the hipEvent_t)
torch.cuda.synchronize()
if GATE and P > 1:
    for i, p in enumerate(pipes):
        p.lines.set_core_gate(done[(i - 1) % P].cuda_event, done[i].cuda_event)      # my core starts when the previous pipeline's core is through
def run(k):
    for _ in range(k):
        for p in pipes: p.step(cur, overlap=True, join=False)
run(1); torch.cuda.synchronize()
t0 = time.perf_counter(); run(K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("pipes", P, "x", B, "frames, gate", GATE, ":", round(P * B * K / dt), "frames/s", round(dt / K * 1e3, 1), "ms per round; lines", float(pipes[0].feat["cur"]["nl"].float().mean()))
