// Internal declarations shared by the HIP translation units of libsslam_frontend.so.
// gfx950 (MI355X) only: wave64, no CUDA-compat layer, no dual paths.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include <mutex>
#include "../../include/sslam_frontend.h"

namespace sslam {

void set_error(const char* fmt, ...);

#define SSLAM_HIP(expr)                                                                     \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            sslam::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return SSLAM_ERR_HIP;                                                           \
        }                                                                                   \
    } while (0)

// growable device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return SSLAM_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        SSLAM_HIP(hipMalloc(&p, want));
        cap = want;
        return SSLAM_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

struct HostPinned {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return SSLAM_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        SSLAM_HIP(hipHostMalloc(&p, bytes + 256, hipHostMallocDefault));
        cap = bytes + 256;
        return SSLAM_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

}  // namespace sslam

struct sslam_prof_rec { const char* name; hipEvent_t a, b; };

struct sslam_ctx {
    int device = 0;
    bool profEnabled = false;              // per-kernel HIP-event timing (sslam_profile_*)
    std::vector<sslam_prof_rec> prof;
    hipStream_t stream = nullptr;
    std::recursive_mutex mu;       // every entry point serialises on the context (SURVEY §8b threading); recursive: the host forms call the *_batch_dev forms
    sslam::DevBuf scratch[8];      // matcher staging
    sslam::DevBuf knnExpand;       // sslam_hamming_knn2_batch_dev: the train rows as int8 matrix-core operands (match_knn.h)
    hipEvent_t knnDone = nullptr;  // recorded behind the kernel that reads knnExpand: a call on ANOTHER stream waits for it before it overwrites the buffer
    void* knnLastStream = nullptr;
    sslam::DevBuf recordOffsets[4];   // sslam_pack_records_dev: per-frame offsets of the record stream, one buffer per stream that packs
    void* recordOffsetsStream[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned long recordOffsetsUse[4] = {0, 0, 0, 0}, recordOffsetsClock = 0;      // least-recently-used recycling of the four slots
    sslam::HostPinned pinned[4];
    int num_cus = 0;
    void* batchCache = nullptr;                  // sslam_frontend_batch: staging buffers, streams, events kept between calls (batch.hip)
    void (*batchCacheFree)(void*) = nullptr;
};

// device-resident copy of one Frame's features: keypoints or keylines, 32-byte descriptors, optional
// mvuRight, image bounds.  Matchers that take a frame handle skip the per-call upload.
struct sslam_frame {
    sslam_ctx* ctx = nullptr;
    int kind = 0, n = 0;           // kind 0: sslam_keypoint rows, 1: sslam_keyline rows
    float bounds[4] = {0, 0, 0, 0};
    bool hasUright = false;
    sslam::DevBuf feats, desc, uright;
};

// device-resident DBoW2 vocabulary tree: CSR children lists, 32-byte node descriptors, word id
// and weight per node (leaves).
struct sslam_vocab {
    sslam_ctx* ctx = nullptr;
    int nnodes = 0, levels = 0;
    int k = 0, nwords = 0, scoring = 0 /* L1_NORM */, weighting = 0 /* TF_IDF */;      // DBoW2 BowVector.h:36-53
    sslam::DevBuf childPtr, children, desc, wordId, weight;
};

// owner context of an extractor handle (the handle types are private to their translation units)
extern "C" __attribute__((visibility("hidden"))) sslam_ctx* sslam_orb_context(sslam_orb* orb);
extern "C" __attribute__((visibility("hidden"))) sslam_ctx* sslam_lines_context(sslam_lines* lines);

namespace sslam {
// roctx ranges (the reference has no tracing; this is the hook the survey proposed): with SSLAM_ROCTX=1 every stage
// scope below also pushes / pops a roctx range named after the kernel, so that `rocprofv3 --marker-trace` shows the host-side launch
// sequence of a call next to the kernel trace.  libroctx64 is bound at run time (dlopen): no link-time dependency, no cost when unset.
void roctx_push(const char* name);      // ctx.hip
void roctx_pop();
// RAII stage timer: records a HIP event pair on the launch stream around one kernel launch.
struct ProfScope {
    sslam_ctx* c; hipStream_t st; sslam_prof_rec r; bool on; bool marked;
    ProfScope(sslam_ctx* ctx, const char* name, hipStream_t s) : c(ctx), st(s), on(ctx->profEnabled), marked(false) {
        static const bool kRoctx = getenv("SSLAM_ROCTX") != nullptr;
        if (kRoctx) { roctx_push(name); marked = true; }
        if (!on) return;
        r.name = name;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(r.a, st);
    }
    ~ProfScope() { if (on) { (void)hipEventRecord(r.b, st); c->prof.push_back(r); } if (marked) roctx_pop(); }
};
}  // namespace sslam

// ---- device helpers (wave64) ---------------------------------------------------
#ifdef __HIPCC__
namespace sslam {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Workgroups are handed to the eight XCDs round-robin (linear id % 8).  With one workgroup per frame a batch whose content
// repeats with a period of 8 (or drifts slowly) would give every XCD the same kind of frame and the kernel would last as long
// as the heaviest kind.  Rotating the assignment inside each group of eight frames by a digit sum of the group index gives every
// XCD (and every CU / SIMD behind it) one frame of every group and a mix of residue classes; the map is a bijection on [0, n)
// (the tail group is left alone).
__device__ __forceinline__ int xcd_mix_frame(int id, int n) {
    const int g = id >> 3;
    const int rot = g + (g >> 3) + (g >> 6) + (g >> 9);       // no power-of-two period: CUs and SIMDs are dealt out round-robin as well
    return ((g << 3) + 8 <= n) ? ((g << 3) | ((id + rot) & 7)) : id;
}

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int mbcnt(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over the wave as a wave-uniform value without an LDS round trip: four DPP row_shr steps leave each row's total in its last lane,
// the four totals are read with v_readlane (wave_sum's six __shfl_xor steps are six dependent ds_bpermute round trips)
__device__ __forceinline__ int wave_sum_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);      // row_shr:8
    return __builtin_amdgcn_readlane(v, 15) + __builtin_amdgcn_readlane(v, 31) + __builtin_amdgcn_readlane(v, 47) + __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (unsigned)__shfl_xor((int)v, o, 64));
    return v;
}
// lane permutations inside a row of 16 run on the DPP path (no LDS round trip): quad swaps, then row rotations by 4 and 8;
// only the two cross-row steps go through ds_bpermute
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)v, CTRL, 0xF, 0xF, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, false);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    unsigned long long w;
    w = dpp_u64<0xB1>(v); v = w < v ? w : v;          // quad_perm [1,0,3,2]
    w = dpp_u64<0x4E>(v); v = w < v ? w : v;          // quad_perm [2,3,0,1]
    w = dpp_u64<0x124>(v); v = w < v ? w : v;         // row_ror:4
    w = dpp_u64<0x128>(v); v = w < v ? w : v;         // row_ror:8
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, o, 64);
        unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), o, 64);
        w = ((unsigned long long)hi << 32) | lo;
        v = w < v ? w : v;
    }
    return v;
}
// inclusive prefix sum across the wave: four DPP row_shr steps inside each row of 16 (zero fill at the row start), then the
// three row totals are added with v_readlane; no LDS round trip
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);      // row_shr:8
    const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
    const int row = lane_id() >> 4;
    return v + (row > 0 ? t0 : 0) + (row > 1 ? t1 : 0) + (row > 2 ? t2 : 0);
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

// cv::fastAtan2 (scalar atan_f32) with explicit round-to-nearest fp32 ops, no FMA
// contraction (oracle decision D4).  Degrees in [0,360).
template <bool BRANCHY = false>
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    // both branches of the reference divide the smaller magnitude by the larger one.  Default: one branch-free instance (fewer
    // instructions, what a SIMD shared by several waves wants); BRANCHY keeps the two-sided form, whose wave-uniform branch
    // is the shorter dependent chain for a lone wave.
    const float ax = fabsf(x), ay = fabsf(y);
    const bool steep = !(ax >= ay);
    float a;
    if (BRANCHY) {
        if (!steep) {
            const float c = __fdiv_rn(ay, __fadd_rn(ax, eps)), c2 = __fmul_rn(c, c);
            a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
        } else {
            const float c = __fdiv_rn(ax, __fadd_rn(ay, eps)), c2 = __fmul_rn(c, c);
            a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
        }
    } else {
        const float c = __fdiv_rn(steep ? ax : ay, __fadd_rn(steep ? ay : ax, eps));
        const float c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
        if (steep) a = __fsub_rn(90.f, a);
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// cvRound(float): round half to even
__device__ __forceinline__ int cv_roundf(float v) { return __float2int_rn(v); }

// sin and cos of x in [0, 6.5] in double, for k_describe's steering angle (a float: the keypoint's angle in degrees times pi / 180).  The library's sincos carries the argument
// reduction for every magnitude and cost ~150 of the kernel's ~700 vector instructions per keypoint (all 64 lanes of the keypoint's wave evaluate the same value) and its
// highest register count; here: quadrant by one multiply (k = 0 .. 4), two-term Cody-Waite reduction (k * pi/2 is exact inside the FMA: k has three bits, the head 53), the fdlibm
// kernels on [-pi/4, pi/4].  sslam_selftest_sincos compares (float)sin, (float)cos with the library's for EVERY float in the range: what k_describe consumes is identical.
__device__ __forceinline__ void sincos_0_2pi(double x, double& sn, double& cs) {
    const int k = (int)__builtin_fma(x, 0.63661977236758134308, 0.5);                  // round(x * 2 / pi): x >= 0
    const double kd = (double)k;
    double r = __builtin_fma(-kd, 1.57079632679489655800e+00, x);                       // pi/2 head
    r = __builtin_fma(-kd, 6.12323399573676603587e-17, r);                              // pi/2 tail
    const double z = r * r;
    // __kernel_sin(r, 0)
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double sr = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double s = r + (z * r) * (S1 + z * sr);
    // __kernel_cos(r, 0)
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double cr = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double ar = fabs(r);
    double c;
    if (ar < 0.3) c = 1.0 - (0.5 * z - z * cr);
    else {
        const double qx = ar > 0.78125 ? 0.28125 : __hiloint2double(__double2hiint(ar) - 0x00200000, 0);      // |r| / 4, truncated
        const double hz = 0.5 * z - qx, a = 1.0 - qx;
        c = a - (hz - z * cr);
    }
    switch (k & 3) {
        case 0: sn = s; cs = c; break;
        case 1: sn = c; cs = -s; break;
        case 2: sn = -s; cs = -c; break;
        default: sn = -c; cs = s; break;
    }
}

}  // namespace sslam
#endif
