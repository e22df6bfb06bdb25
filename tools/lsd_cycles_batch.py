"""Stage clocks of the sequential LSD core in THROUGHPUT mode (a full batch, every wave slot taken): needs a library built with
-DSSLAM_LSD_CYCLES (tools/build_variant.sh cyc -DSSLAM_LSD_CYCLES; SSLAM_LIB=...).  Prints the mean share of region growing / region2rect /
refine per frame, the split of growing into "waiting for a staging's gather" and "accept loop", and the kernel time."""
import sys, os, ctypes as C
sys.path.insert(0, 'tests')
import numpy as np, torch, pkg
from synth import synth_frame
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
U = int(sys.argv[2]) if len(sys.argv) > 2 else 8
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
    for rep in range(2):
        fe.lib().sslam_profile_enable(ctx.h, 1)
        pipe.lines.extract_batch_dev(imgs, 640, 480, 640, 640 * 480, B, f["kl"], f["ldesc"], f["linefn"], f["nl"], 200, st)
        torch.cuda.synchronize()
        prof = pipeline.profile_drain(fe, ctx)
print("B", B, {k: round(v[0], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:3]})
out = (C.c_longlong * 8)()
rows = []
for s in range(0, B, max(1, B // 128)):
    fe.lib().sslam_lines_debug_cycles(pipe.lines.h, s, out)
    rows.append([out[i] for i in range(8)])
A = np.array(rows, dtype=np.float64)
tot = A[:, 4]
print("frames sampled", len(A), "total cycles mean %.3e (min %.3e max %.3e)" % (tot.mean(), tot.min(), tot.max()))
for i, nm in enumerate(["grow (first growth)", "region2rect (first)", "refine block (re-grow + rects + radius)", "  of which reduce_region_radius"]):
    print("%-45s %5.1f %%" % (nm, 100 * (A[:, i] / tot).mean()))
print("%-45s %5.1f %%" % ("seed scan + emit + rest", 100 * (1 - (A[:, 0] + A[:, 1] + A[:, 2]) / tot).mean()))
print("all growths: waiting for the staging gather %5.1f %%, accept loops %5.1f %% of the total; %.0f stagings/frame, %.0f cycles wait and %.0f cycles accept loop per staging"
      % (100 * (A[:, 5] / tot).mean(), 100 * (A[:, 6] / tot).mean(), A[:, 7].mean(), (A[:, 5] / A[:, 7]).mean(), (A[:, 6] / A[:, 7]).mean()))
