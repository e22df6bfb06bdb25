"""Line-extraction kernel times for the bench's 64 varied frames (B = 6144) under the environment knobs given on the command line
(NAME=VALUE ...); prints the top kernels.  Example: python tools/lsd_env_probe.py SSLAM_COUNT_WAVES=2"""
import sys, os
for a in sys.argv[1:]:
    k, v = a.split("=", 1); os.environ[k] = v
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch, pkg
import bench
fe = pkg.frontend(); ctx = fe.Context(0)
pipeline = pkg._load("sslam_pipeline", os.path.join(pkg.PKG_DIR, "pipeline.py"))
cur, prev = bench.synth_frames(640, 480, 64, 0)
B = 6144
imgs = torch.from_numpy(np.stack(cur)).cuda().repeat(B // 64, 1, 1).contiguous()
pipe = pipeline.FrontendBatch(fe, ctx, 640, 480, B, 1000, 200, "cuda:0", with_match=False)
f = pipe.feat["cur"]
with torch.cuda.stream(torch.cuda.Stream()):
    st = torch.cuda.current_stream().cuda_stream
    pipe.lines.extract_batch_dev(imgs, 640, 480, 640, 640 * 480, B, f["kl"], f["ldesc"], f["linefn"], f["nl"], 200, st)
    torch.cuda.synchronize()
    fe.lib().sslam_profile_enable(ctx.h, 1)
    for _ in range(2):
        pipe.lines.extract_batch_dev(imgs, 640, 480, 640, 640 * 480, B, f["kl"], f["ldesc"], f["linefn"], f["nl"], 200, st)
    torch.cuda.synchronize()
prof = pipeline.profile_drain(fe, ctx)
print(" ".join(sys.argv[1:]) or "default", {k: round(v[0] / 2, 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get('TOPK', '6'))]}, "sum %.2f" % (sum(v[0] for v in prof.values()) / 2),
      "cksum", int(f["ldesc"][:64].to(torch.int64).sum().item()))
