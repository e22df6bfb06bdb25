import sys, os; sys.path.insert(0, 'tests')
import numpy as np, torch, pkg
from synth import synth_frame, warp_prev, noise_frame, const_frame
fe = pkg.frontend(); pipeline = pkg._load("sslam_pipeline", os.path.join(pkg.PKG_DIR, "pipeline.py"))
ctx = fe.Context(0)
frames = [synth_frame(2000), noise_frame(9), const_frame(), synth_frame(2001), warp_prev(synth_frame(2000))]
B = len(frames)
imgs = torch.from_numpy(np.stack(frames)).cuda()
pipe = pipeline.FrontendBatch(fe, ctx, 640, 480, B, 1000, 200, "cuda:0")
ox = fe.OrbExtractor(ctx, 1000); lx = fe.LineExtractor(ctx, 200)
single = []
for f in frames:
    kp, desc = ox(f); kl, ld, fn = lx(f)
    kp2, desc2 = ox(f); kl2, ld2, fn2 = lx(f)
    print('single repeat equal:', np.array_equal(kp.view(np.uint8), kp2.view(np.uint8)), np.array_equal(desc, desc2), np.array_equal(kl.view(np.uint8), kl2.view(np.uint8)), np.array_equal(ld, ld2), len(kp), len(kl))
    single.append((kp, desc, kl, ld))
for it in range(6):
    ov = it % 2 == 1
    pipe.step(imgs, overlap=ov)
    torch.cuda.synchronize()
    c = pipe.feat["cur"]
    n = c["n"].cpu().numpy(); nl = c["nl"].cpu().numpy()
    kp = c["kp"].cpu().numpy(); desc = c["desc"].cpu().numpy(); kl = c["kl"].cpu().numpy(); ld = c["ldesc"].cpu().numpy()
    for i in range(B):
        skp, sdesc, skl, sld = single[i]
        ok_n = n[i] == len(skp); ok_nl = nl[i] == len(skl)
        ok_kp = ok_n and np.array_equal(kp[i][:n[i]].view(np.uint8).reshape(-1), skp.view(np.uint8).reshape(-1))
        ok_d = ok_n and np.array_equal(desc[i][:n[i]], sdesc)
        ok_kl = ok_nl and np.array_equal(kl[i][:nl[i]].view(np.uint8).reshape(-1), skl.view(np.uint8).reshape(-1))
        ok_ld = ok_nl and np.array_equal(ld[i][:nl[i]], sld)
        print(it, 'ov' if ov else 'se', 'frame', i, 'n', n[i], len(skp), 'nl', nl[i], len(skl), 'kp', ok_kp, 'desc', ok_d, 'kl', ok_kl, 'ld', ok_ld)
