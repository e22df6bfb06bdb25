"""FETCH_SIZE calibration on this device (run under `rocprofv3 --pmc FETCH_SIZE`): known byte counts in the two access
patterns the library uses most.  Kernel names in the PMC summary: k_probe_stream16 (4 GiB, 16 B/lane coalesced, read once)
and k_probe_gather16 (4 GiB buffer, 2^28 scattered 16-B gathers = 4 GiB requested, 128-B lines touched = 32 GiB)."""
import sys, ctypes as C; sys.path.insert(0, 'tests')
import pkg
fe = pkg.frontend(); ctx = fe.Context(0)
for mode in (0, 1):
    req = C.c_longlong(0)
    rc = fe.testing_lib().sslam_selftest_fetch_probe(ctx.h, C.c_size_t(4 << 30), mode, C.byref(req))
    print("mode", mode, "rc", rc, "bytes requested", req.value)
