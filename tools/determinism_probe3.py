"""One batch through the two-stream pipeline three times (with and without priming the previous frame), its packed record stream
(sslam_pack_records_dev) unpacked by the library's own sslam_unpack_records and compared frame by frame with single-frame extraction:
python tools/determinism_probe3.py   (GPU)"""
import sys, os; sys.path.insert(0, 'tests')
import numpy as np, torch, pkg
from synth import synth_frame, warp_prev, noise_frame, const_frame
fe = pkg.frontend(); pipeline = pkg._load("sslam_pipeline", os.path.join(pkg.PKG_DIR, "pipeline.py"))
ctx = fe.Context(0)
frames = [synth_frame(2000), noise_frame(9), const_frame(), synth_frame(2001), warp_prev(synth_frame(2000))]
B = len(frames)
bad = 0
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
    kp_r, d_r, n_r, kl_r, ld_r, fn_r, nl_r, nrec = fe.unpack_records(pipe.packed_stream().cpu().numpy(), B, pipe.cap, pipe.lcap, True)
    assert nrec == B
    ox = fe.OrbExtractor(ctx, 1000); lx = fe.LineExtractor(ctx, 200)
    for i, f in enumerate(frames):
        kp, desc = ox(f); kl, ld, fn = lx(f)
        okp = n_r[i] == len(kp) and np.array_equal(kp_r[i, :n_r[i]].view(np.uint8).reshape(-1, 28), kp.view(np.uint8).reshape(len(kp), 28)) and np.array_equal(d_r[i, :n_r[i]], desc)
        old = nl_r[i] == len(kl) and np.array_equal(ld_r[i, :nl_r[i]], ld) and np.array_equal(fn_r[i, :nl_r[i]], fn)
        bad += (not okp) + (not old)
        print('  frame', i, 'rec n', n_r[i], 'single', len(kp), 'rec nl', nl_r[i], 'single', len(kl), 'kp', okp, 'lines', old)
    ox.close(); lx.close(); pipe.close()
print("determinism_probe3:", "OK" if bad == 0 else "%d MISMATCHES" % bad)
sys.exit(1 if bad else 0)
