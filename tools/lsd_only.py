"""Times the line-extraction kernels alone on a full batch (HIP events per kernel): python tools/lsd_only.py [B] [unique] [reps]"""
import sys, os
sys.path.insert(0, 'tests')
import numpy as np, torch, pkg
from synth import synth_frame
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
U = int(sys.argv[2]) if len(sys.argv) > 2 else 8
R = int(sys.argv[3]) if len(sys.argv) > 3 else 3
fe = pkg.frontend(); ctx = fe.Context(0)
pipeline = pkg._load("sslam_pipeline", os.path.join(pkg.PKG_DIR, "pipeline.py"))
cache = "/tmp/lsd_only_%d.npy" % U
if os.path.exists(cache): frames = np.load(cache)
else:
    frames = np.stack([synth_frame(2000 + i) for i in range(U)]); np.save(cache, frames)
imgs = torch.from_numpy(frames).cuda().repeat((B + U - 1) // U, 1, 1)[:B].contiguous()
pipe = pipeline.FrontendBatch(fe, ctx, 640, 480, B, 1000, 200, "cuda:0", with_match=False)
f = pipe.feat["cur"]
with torch.cuda.stream(torch.cuda.Stream()):
    st = torch.cuda.current_stream().cuda_stream
    pipe.lines.extract_batch_dev(imgs, 640, 480, 640, 640 * 480, B, f["kl"], f["ldesc"], f["linefn"], f["nl"], 200, st)
    torch.cuda.synchronize()
    fe.lib().sslam_profile_enable(ctx.h, 1)
    for _ in range(R):
        pipe.lines.extract_batch_dev(imgs, 640, 480, 640, 640 * 480, B, f["kl"], f["ldesc"], f["linefn"], f["nl"], 200, st)
    torch.cuda.synchronize()
prof = pipeline.profile_drain(fe, ctx)
nl = f["nl"].cpu().numpy()
print(os.environ.get("SSLAM_LIB", "product")[-40:], "B", B, "lines/frame %.1f" % nl.mean(), {k: round(v[0] / R, 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("LSD_ONLY_TOP", "4"))]},
      "sum %.2f" % (sum(v[0] for v in prof.values()) / R), "cksum", int(f["ldesc"].to(torch.int64).sum().item()))
