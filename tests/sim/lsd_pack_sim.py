"""Design study (test infrastructure, not product): work statistics of the sequential LSD core on the benchmark's synthetic frames, from the
CPU oracle's own main loop (tests/sim/lsd_trace.cpp).  Used to size the frames-per-wave packing of k_lsd_regions."""
import ctypes as C, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from synth import synth_frame

so = os.path.join(HERE, "libtrace.so")
src = os.path.join(HERE, "lsd_trace.cpp")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, src,
                           os.path.join(HERE, "..", "..", "oracle", "orb_oracle.cpp")] if False else
                          ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, src])
L = C.CDLL(so)
names = ["nDefined", "seeds", "accepted", "regionsGE", "refines", "reduceIters", "rectCalls", "rectPoints", "regrowAccepted",
         "stag1", "stag2", "stag4", "stag8", "tick1", "tick2", "tick4", "tick8"] + ["size<=2^%d" % i for i in range(12)] + ["acc%d" % i for i in range(9)]
rows = []
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for i in range(n):
    img = synth_frame(2000 + i)
    out = (C.c_longlong * 64)()
    k = L.lsd_trace_frame(C.c_void_p(img.ctypes.data), 640, 480, out, 64)
    rows.append([out[j] for j in range(k)])
A = np.array(rows)
for j, nm in enumerate(names):
    print("%-16s mean %10.1f  min %8d  max %8d" % (nm, A[:, j].mean(), A[:, j].min(), A[:, j].max()))
