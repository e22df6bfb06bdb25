"""Line extraction of small resident batches (1 .. 16 frames per call): up to 8 frames take the cluster form of the LSD core (one XCD's worth of
workgroups per frame), more the multi-wave form.  Prints ms per call (p50 over the bench's frames) and frames/s.  SSLAM_LSD_FLAVOUR=mw forces
the multi-wave form for comparison."""
import sys, os, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch, pkg, bench
torch.cuda.set_device(0)
fe = pkg.frontend(); ctx = fe.Context(0)
cur, prev = bench.synth_frames(640, 480, 64, 0)
for nf in ([int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64, 96]):      # frames per call (arguments: other sizes, e.g. 128 512 1024 3072)
    ex = fe.LineExtractor(ctx, 200)
    cap = 256
    d_kl = torch.zeros(nf * cap * 68, dtype=torch.uint8, device="cuda"); d_ld = torch.zeros(nf * cap * 32, dtype=torch.uint8, device="cuda")
    d_fn = torch.zeros(nf * cap * 3, dtype=torch.float64, device="cuda"); d_n = torch.zeros(nf, dtype=torch.int32, device="cuda")
    ts = []
    for rep in range(max(64 // nf, 4) + 2):
        fr = [cur[(rep * nf + i) % 64] for i in range(nf)]
        dev = torch.from_numpy(np.stack(fr)).cuda(); torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            a.record(st)
            ex.extract_batch_dev(dev, 640, 480, 640, 640 * 480, nf, d_kl, d_ld, d_fn, d_n, cap, st.cuda_stream)
            b.record(st)
        st.synchronize()
        if rep >= 2: ts.append(a.elapsed_time(b))
    ts = np.array(ts)
    print("%2d frames per call: %.2f ms p50 (%.2f p90)  %.0f frames/s" % (nf, np.percentile(ts, 50), np.percentile(ts, 90), nf / np.percentile(ts, 50) * 1e3))
    ex.close()
