"""Cross-check of the drift-bounded shortcut in the LSD core (csrc/lsd_regions.h): needs a library built with
SSLAM_EXTRA_FLAGS=-DSSLAM_LSD_DRIFT_VERIFY.  Every shortcut decision is compared in-kernel with the exact test; prints the number of
disagreements (must be 0), and how many decisions took the shortcut."""
import sys, os, ctypes as C
sys.path.insert(0, 'tests')
import numpy as np, torch, pkg
from synth import synth_frame, noise_frame
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
fe = pkg.frontend(); ctx = fe.Context(0)
pipeline = pkg._load("sslam_pipeline", os.path.join(pkg.PKG_DIR, "pipeline.py"))
rng = np.random.default_rng(5)
frames = []
for i in range(B):
    k = i % 4
    if k == 0: frames.append(synth_frame(9000 + i))
    elif k == 1: frames.append(synth_frame(9000 + i, nshapes=int(rng.integers(5, 150)), nstrokes=int(rng.integers(0, 120)), noise=float(rng.uniform(0, 8))))
    elif k == 2: frames.append(synth_frame(9000 + i, noise=float(rng.uniform(4, 12))))
    else: frames.append(noise_frame(i) if i % 16 == 3 else synth_frame(9000 + i, nshapes=200, nstrokes=200))
imgs = torch.from_numpy(np.stack(frames)).cuda()
pipe = pipeline.FrontendBatch(fe, ctx, 640, 480, B, 1000, 200, "cuda:0", with_match=False)
with torch.cuda.stream(torch.cuda.Stream()):
    st = torch.cuda.current_stream().cuda_stream
    f = pipe.feat["cur"]
    pipe.lines.extract_batch_dev(imgs, 640, 480, 640, 640 * 480, B, f["kl"], f["ldesc"], f["linefn"], f["nl"], 200, st)
torch.cuda.synchronize()
out = (C.c_longlong * 8)()
bad = dec = short = 0
for s in range(B):
    fe.lib().sslam_lines_debug_cycles(pipe.lines.h, s, out)
    bad += out[6]; dec += out[7] & 0xFFFFFFFF; short += out[7] >> 32
print("frames %d  decisions outside the exact path %d  shortcuts %d (%.1f %%)  disagreements %d" % (B, dec, short, 100.0 * short / max(dec, 1), bad))
assert bad == 0
