"""Cycle breakdown of the sequential LSD core + single-frame latencies.  The stage clocks are compiled out by default:
build with SSLAM_EXTRA_FLAGS=-DSSLAM_LSD_CYCLES python structure-slam-pointline_amd/build.py --force for the breakdown (the latencies need nothing)."""
import sys, ctypes as C; sys.path.insert(0,'tests')
import numpy as np, pkg
from synth import synth_frame
fe = pkg.frontend(); ctx = fe.Context(0)
ex = fe.LineExtractor(ctx, 200)
img = synth_frame(2000)
for rep in range(2):
    kl, ld, fn = ex(img)
out = (C.c_longlong*8)()
fe.lib().sslam_lines_debug_cycles(ex.h, 0, out)
tot = out[4] or 1
print('nfa count %.1f%% math %.1f%%' % (100*out[5]/tot, 100*out[6]/tot)); print('lines', len(kl), 'cycles: grow %.1f%% rect %.1f%% refine %.1f%% (of which reduce_region_radius %.1f%%) total %d (%.2f ms @2.4GHz?)' % (100*out[0]/tot, 100*out[1]/tot, 100*out[2]/tot, 100*out[3]/tot, tot, tot/2.4e6))
import time
t=time.time(); 
for _ in range(5): ex(img)
print('single-frame lines latency ms', (time.time()-t)/5*1e3)
ox = fe.OrbExtractor(ctx)
ox(img); t=time.time()
for _ in range(10): ox(img)
print('single-frame ORB latency ms', (time.time()-t)/10*1e3)
