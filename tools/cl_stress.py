"""Race hunt for the cluster form of the LSD core: the same frames extracted over and over (every schedule of main wave / helpers / feeder is
different), alone and with a batch of ORB extraction running on another stream at the same time (helper workgroups then start late or not at
all: the main wave's own claims, the bounded waits).  Every run must equal the oracle's segments and descriptors.
usage: cl_stress.py [repetitions per frame = 300] [frames = 6]"""
import sys, os, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, ctypes as C, torch, pkg, bench, oracle_lib
from synth import synth_frame, noise_frame
torch.cuda.set_device(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 6
fe = pkg.frontend(); ctx = fe.Context(0); orc = oracle_lib.Oracle()
cur, prev = bench.synth_frames(640, 480, 64, 0)
frames = [cur[i * 7 % 64] for i in range(nfr - 2)] + [synth_frame(1235, w=1280, h=960), noise_frame(3, w=320, h=240)]
lx = fe.LineExtractor(ctx, 400)
orb = fe.OrbExtractor(ctx, 1000)
busy = torch.from_numpy(np.stack(cur)).cuda()
d_kp = torch.zeros(64 * orb.cap * 28, dtype=torch.uint8, device="cuda"); d_desc = torch.zeros(64 * orb.cap * 32, dtype=torch.uint8, device="cuda"); d_n = torch.zeros(64, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()
bad = 0; total = 0; expired = 0; t0 = time.time()
for fi, f in enumerate(frames):
    okl, old, ofn, oraw = orc.lines_extract(f, 400)
    for r in range(reps):
        noisy = r % 3 == 2
        if noisy:
            with torch.cuda.stream(side):
                orb.extract_batch_dev(busy, 640, 480, 640, 640 * 480, 64, d_kp, d_desc, d_n, orb.cap, side.cuda_stream)
        kl, ld, fn = lx(f)
        same = np.array_equal(lx.debug_segments(0), oraw) and np.array_equal(ld, old) and np.array_equal(fn, ofn)
        out = (C.c_longlong * 8)(); fe.lib().sslam_lines_debug_cycles(lx.h, 0, out); expired += int(out[7])
        bad += int(not same); total += 1
        if not same: print("frame", fi, f.shape, "rep", r, "noisy" if noisy else "", "DIFFERS")
    side.synchronize()
print("cl_stress: %d extractions of %d frames (every third with 64 frames of ORB extraction on another stream) in %.1f s; bounded waits expired %d; %d mismatches" % (total, len(frames), time.time() - t0, expired, bad))
