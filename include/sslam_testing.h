/* Test-only entry points -- NOT part of the drop-in boundary (include/sslam_frontend.h is) and NOT in libsslam_frontend.so: they exist in
 * libsslam_frontend_testing.so, the same sources compiled with -DSSLAM_TESTING (structure-slam-pointline_amd/build.py builds both; the testing library exports everything the
 * product library does plus what is declared here).  tests/ and tools/ call them; a product caller has no reason to. */
#ifndef SSLAM_TESTING_H
#define SSLAM_TESTING_H
#include "sslam_frontend.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Multi-GPU group code (sslam_group_create / _create_rank, reference: the batch-of-frames mode of BASELINE configs[4]) on a box with ONE GPU:
 * while `on` is nonzero, NEW groups bind an in-process stand-in for the RCCL entry points (host-mediated device-to-device copies with NCCL's matching rules;
 * same process only) and may hold more members than GPUs are visible -- so that the group code's multi-member paths (a host thread per member, uneven tails,
 * the collective error agreement, grouped send / receive to the root) execute with real device buffers.  It says nothing about xGMI.  Process-wide; returns the
 * previous setting.  (Round 4 selected the stand-in with the environment variable SSLAM_GROUP_FAKE_RCCL; an explicit call cannot be set by accident.) */
int sslam_testing_use_rccl_standin(int on);

/* Self-test of the table-based exact integer division of the NFA binomial tail against the hardware IEEE division:
 * `pairs` random quotients a/b with 1 <= a,b < n; *mismatches_out must come back 0. */
int sslam_selftest_exact_div(sslam_ctx* ctx, int n, long long pairs, long long* mismatches_out);
/* The log-gamma / log(p) / reciprocal tables of the NFA stage as the library evaluates them on the host with the reference's own libm
 * expressions (opencv lsd.cpp log_gamma_windschitl / log_gamma_lanczos, reached from src/ExtractLineSegment.cpp:38-40): out[2n + 48].
 * Host-only; lets a test pin the table bits (a libm that rounds differently would otherwise go unnoticed until a rectangle flips). */
int sslam_debug_nfa_tables(int n, double* out);
/* Self-test of the guarded fp32 early-exit test used in the NFA tail loop (the reference's `err < tolerance * ...` test,
 * opencv lsd.cpp nfa(), reached from src/ExtractLineSegment.cpp:38-43): random inputs, half on the decision boundary.
 * disagree_out must be 0; ambiguous_out = cases that fall back to the fp64 expression. */
int sslam_selftest_tail_test(sslam_ctx* ctx, long long samples, long long* disagree_out, long long* ambiguous_out);
/* The region-growing core replaces the IEEE division inside cv::fastAtan2 by the hardware's refinement sequence without its
 * scaling / special-case steps (identity on the value range of a region's direction sums) and the quadrant compares by sign-bit
 * arithmetic: `samples` random and adversarial sums, bit-compared with the `/` operator and the straight form.
 * mismatches_out[0] = divisions that differ, mismatches_out[1] = angles that differ. */
int sslam_selftest_region_div(sslam_ctx* ctx, long long samples, long long mismatches_out[2]);

/* k_describe steers the rBRIEF pattern with (float)cos / (float)sin of the keypoint angle evaluated in double (src/ORBextractor.cc:108-147 computeOrbDescriptor) through a
 * reduction of its own for [0, 6.5] instead of the library's sincos: every float of the range, both results after the rounding to float; *mismatches_out must be 0. */
int sslam_selftest_sincos(sslam_ctx* ctx, long long* mismatches_out);
/* k_lsd_hist_sort bins a pixel's |g|^2 (LSD's 1 024 gradient bins, opencv lsd.cpp ll_angle, reached from src/ExtractLineSegment.cpp:38-40) in fp32 where fp32 decides and with
 * the reference's fp64 expression otherwise: every s in [0, max_s] (max_s = the frame's largest |g|^2, < 2^24) against the fp64 expression; *mismatches_out must be 0. */
int sslam_selftest_lsd_bin(sslam_ctx* ctx, int max_s, long long* mismatches_out);
/* k_lbd's walk rounds a coordinate with ONE conversion (v_cvt_rpi_i32_f32 = floor(x + 0.5)) where BinaryDescriptor::computeLBD has (short)round(x) under a clamp to the image
 * (OpenCV line_descriptor binary_descriptor.cpp, reached from src/ExtractLineSegment.cpp:53): every float bit pattern of the coordinate range, against the previous
 * instruction sequence under the clamps (mismatches_out[0]) and against roundf for x >= 0 (mismatches_out[1]).  Both must be 0. */
int sslam_selftest_lbd_round(sslam_ctx* ctx, long long mismatches_out[2]);

/* Profiling aid (no reference counterpart): the chip's issue rate for one kind of vector instruction (0: v_add_u32, 1: v_fma_f32,
 * 2: v_add_f64, 3: v_bcnt_u32_b32), 16 independent instructions per lane and round with 8 waves per SIMD resident: wave-instructions per
 * second in units of 1e9.  What the SQ utilisation figures of profiles/README.md are priced against. */
int sslam_selftest_valu_rate(sslam_ctx* ctx, int kind, double* ginst_per_s_out);
/* Profiling aid (no reference counterpart): reads a known number of bytes in one of the library's two dominant access
 * patterns (mode 0: 16 B/lane coalesced stream, mode 1: scattered 16-B gathers) so that rocprofv3's FETCH_SIZE can be
 * calibrated on this device (tools/fetch_probe.py, profiles/README.md). */
int sslam_selftest_fetch_probe(sslam_ctx* ctx, size_t bytes, int mode, long long* bytes_requested_out);
/* counters of the cluster form of the sequential core (one frame at a time, helper waves on several compute units) for frame `frame` of
 * the last call; meaningful in builds with -DSSLAM_CL_CYCLES only (tools/cl_probe.py). */
int sslam_lines_debug_cluster(sslam_lines* ln, int frame, long long* out8);


#ifdef __cplusplus
}
#endif
#endif
