import sys, os; sys.path.insert(0, 'tests')
import numpy as np, torch, pkg
from synth import synth_frame, warp_prev, noise_frame, const_frame
fe = pkg.frontend(); pipeline = pkg._load("sslam_pipeline", os.path.join(pkg.PKG_DIR, "pipeline.py"))
sh = pkg._load("sslam_sharding", os.path.join(pkg.PKG_DIR, "sharding.py"))
ctx = fe.Context(0)
frames = [synth_frame(2000), noise_frame(9), const_frame(), synth_frame(2001), warp_prev(synth_frame(2000))]
B = len(frames)
for rep in range(3):
    pipe = pipeline.FrontendBatch(fe, ctx, 640, 480, B, 1000, 200, "cuda:0")
    imgs = torch.from_numpy(np.stack(frames)).cuda()
    prev = torch.from_numpy(np.stack([warp_prev(f) for f in frames])).cuda()
    if rep != 1:
        pipe.extract(prev, "prev")
    pipe.step(imgs, overlap=True)
    torch.cuda.synchronize()
    c = pipe.feat["cur"]
    print('rep', rep, 'direct n', c["n"].cpu().numpy(), 'nl', c["nl"].cpu().numpy())
    rec = pipe.packed_results().cpu().numpy()
    ox = fe.OrbExtractor(ctx, 1000); lx = fe.LineExtractor(ctx, 200)
    for i, f in enumerate(frames):
        r = sh.unpack_record(rec[i], pipe.cap, pipe.lcap)
        kp, desc = ox(f); kl, ld, fn = lx(f)
        print('  frame', i, 'rec n', r["n"], 'single', len(kp), 'rec nl', r["nl"], 'single', len(kl),
              'kp', r["n"] == len(kp) and np.array_equal(r["kp"], kp.view(np.uint8).reshape(len(kp), 28)),
              'ld', r["nl"] == len(kl) and np.array_equal(r["ldesc"], ld))
    ox.close(); lx.close(); pipe.close()
