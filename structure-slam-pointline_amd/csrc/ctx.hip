// Context, error reporting and ABI housekeeping of libsslam_frontend.so.
#include "common.h"
#include <cstdarg>
#include <dlfcn.h>

namespace sslam {
// roctx, bound at run time (SSLAM_ROCTX=1): roctxRangePushA / roctxRangePop of libroctx64 (or the rocprofiler-sdk build of it)
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
    Roctx() {
        void* h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA"); pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
Roctx& roctx() { static Roctx r; return r; }
}  // namespace
void roctx_push(const char* name) { Roctx& r = roctx(); if (r.push) (void)r.push(name); }
void roctx_pop() { Roctx& r = roctx(); if (r.pop) (void)r.pop(); }

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace sslam

using namespace sslam;

extern "C" const char* sslam_last_error(void) { return g_err; }

extern "C" const char* sslam_status_str(int s) {
    switch (s) {
        case SSLAM_OK: return "ok";
        case SSLAM_ERR_INVALID: return "invalid argument";
        case SSLAM_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
        case SSLAM_ERR_CAPACITY: return "output buffer too small";
        case SSLAM_ERR_HIP: return "HIP runtime error";
        case SSLAM_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown status";
    }
}

extern "C" int sslam_abi_version(void) { return 1; }

extern "C" int sslam_ctx_create(int device, sslam_ctx** out) {
    if (!out) return SSLAM_ERR_INVALID;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device visible (%s); libsslam_frontend has no CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return SSLAM_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) { set_error("device %d out of range (have %d)", device, n); return SSLAM_ERR_INVALID; }
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice(%d) failed", device); return SSLAM_ERR_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { set_error("hipGetDeviceProperties failed"); return SSLAM_ERR_NO_DEVICE; }
    sslam_ctx* c = new sslam_ctx();
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; set_error("hipStreamCreate failed"); return SSLAM_ERR_NO_DEVICE; }
    *out = c;
    return SSLAM_OK;
}

extern "C" int sslam_ctx_destroy(sslam_ctx* c) {
    if (!c) return SSLAM_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->batchCache && c->batchCacheFree) { c->batchCacheFree(c->batchCache); c->batchCache = nullptr; }
    for (auto& b : c->scratch) b.release();
    c->knnExpand.release();
    if (c->knnDone) (void)hipEventDestroy(c->knnDone);
    for (auto& b : c->recordOffsets) b.release();
    for (auto& b : c->pinned) b.release();
    (void)hipStreamDestroy(c->stream);
    delete c;
    return SSLAM_OK;
}

extern "C" int sslam_ctx_synchronize(sslam_ctx* c) {
    if (!c) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(c->device));
    SSLAM_HIP(hipStreamSynchronize(c->stream));
    return SSLAM_OK;
}

extern "C" void* sslam_ctx_stream(sslam_ctx* c) { return c ? (void*)c->stream : nullptr; }

// ---- per-kernel profiling (HIP events on the launch stream) -------------------
#include <map>
extern "C" int sslam_profile_enable(sslam_ctx* c, int on) {
    if (!c) return SSLAM_ERR_INVALID;
    c->profEnabled = on != 0;
    return SSLAM_OK;
}

// Drains the recorded event pairs: per distinct kernel name, total milliseconds and launch count.
// names_out receives pointers to static strings.  Returns the number of distinct kernels (<= cap).
extern "C" int sslam_profile_drain(sslam_ctx* c, const char** names_out, double* ms_out, int* launches_out, int cap) {
    if (!c) return SSLAM_ERR_INVALID;
    if (hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return SSLAM_ERR_HIP;
    std::map<std::string, std::pair<double, int>> agg;
    std::map<std::string, const char*> nm;
    for (auto& r : c->prof) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { auto& e = agg[r.name]; e.first += ms; e.second += 1; nm[r.name] = r.name; }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    c->prof.clear();
    int n = 0;
    for (auto& kv : agg) {
        if (n < cap) { if (names_out) names_out[n] = nm[kv.first]; if (ms_out) ms_out[n] = kv.second.first; if (launches_out) launches_out[n] = kv.second.second; }
        ++n;
    }
    return n;
}
