import sys, os; sys.path.insert(0, 'tests')
import numpy as np, torch, pkg
from synth import synth_frame, warp_prev, noise_frame, const_frame
fe = pkg.frontend(); pipeline = pkg._load("sslam_pipeline", os.path.join(pkg.PKG_DIR, "pipeline.py"))
ctx = fe.Context(0)
frames = [synth_frame(2000 + i) for i in range(8)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
imgs = torch.from_numpy(np.stack(frames)).cuda().repeat(B // 8, 1, 1).contiguous()
pipe = pipeline.FrontendBatch(fe, ctx, 640, 480, B, 1000, 200, "cuda:0")
pipe.extract(imgs, "prev")
def snap():
    torch.cuda.synchronize()
    c = pipe.feat["cur"]
    return {k: c[k].cpu().numpy().copy() for k in ("n", "nl", "kp", "desc", "kl", "ldesc")}
ref = None
for it in range(6):
    pipe.step(imgs, overlap=(it % 2 == 1))
    s = snap()
    # frames repeat with period 8: every replica must equal replica 0
    for k in ("n", "nl"):
        a = s[k].reshape(B // 8, 8)
        bad = (a != a[0]).any(axis=1).sum()
        print(it, 'overlap' if it % 2 else 'serial', k, 'replicas differing:', int(bad), a[0])
    if ref is not None:
        for k in s:
            print('   vs first run', k, 'equal' if np.array_equal(s[k], ref[k]) else 'DIFF')
    else:
        ref = s
