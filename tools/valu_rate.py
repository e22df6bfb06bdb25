"""Issue rate of the vector pipes (sslam_selftest_valu_rate): python tools/valu_rate.py   (GPU)"""
import sys, ctypes as C; sys.path.insert(0, 'tests')
import pkg
fe = pkg.frontend(); ctx = fe.Context(0)
for kind, name in enumerate(("v_add_u32", "v_fma_f32", "v_add_f64", "v_bcnt_u32_b32")):
    g = C.c_double(0)
    rc = fe.testing_lib().sslam_selftest_valu_rate(ctx.h, kind, C.byref(g))
    print("%-16s %8.1f G wave-instructions/s  = %.2f cycles per wave-instruction per SIMD at 2.4 GHz (1024 SIMDs)" % (name, g.value, 1024 * 2.4 / g.value) if rc == 0 else name + " failed")
