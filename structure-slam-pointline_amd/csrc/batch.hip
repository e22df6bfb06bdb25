// Host-buffer batch mode: the `sslam_frontend_batch` of SURVEY.md §8(b) -- n frames of one size in host memory go through
// Frame::ExtractORB + Frame::ExtractLSD (src/Frame.cc:150-161) and come back as per-frame host records.  Pure host orchestration over the
// *_batch_dev entry points.
//
// Round 3 (the entry ran at 17 k frames/s against 60 k for resident frames):
//   * a chunk is as large as the sequential LSD core wants its launches -- 6144 frames fill every wave slot of the chip (24 single-wave
//     workgroups per CU); chunks of 1024 ran the core at one wave per SIMD.  Default chunk = min(n, 6144), bounded by free device memory;
//   * the point branch and the line branch of a chunk run on two HIP streams, the point branch released by the event the library records
//     right before the sequential core (sslam_lines_set_core_event) -- the alignment pipeline.py uses for resident batches, now inside the
//     library;
//   * pageable caller memory goes through a pinned bounce ring (two chunk-sized buffers each way) fed by a pool of host copy threads that lives in the per-context
//     cache: chunk k is staged WHILE chunk k-2's results are copied out (two task groups on the same pool), both while the GPU works on chunk k-1.  Rounds 3-5 ran the two
//     copies one after the other on the calling thread with at most 8 helpers (2.6 GB of memcpy per 6 144-frame chunk against 85-95 ms of GPU time per chunk: the
//     host was the bottleneck, 43.8 k frames/s against 63.4 k from pinned memory);
//   * uploads, kernels and downloads of neighbouring chunks overlap as before (two slots);
//   * sslam_frontend_batch_match adds the match stage of BASELINE configs[2] ("extract + Hamming match vs previous frame"): frame i is
//     matched against frame i-1 of the call -- ORBmatcher::SearchForInitialization, the dense Hamming 2-NN and the LSD line matcher, the
//     three launches pipeline.py times for resident frames.  The device arrays of a chunk carry one extra frame in front (the last frame
//     of the chunk before it), so "previous" and "current" are the same arrays one frame apart and the *_batch_dev matchers run unchanged.
#include "common.h"
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

using namespace sslam;

extern "C" int sslam_orb_batch_status_dev(sslam_orb* orb, int cap, int32_t* d_status4, void* stream);
extern "C" int sslam_lines_batch_status_dev(sslam_lines* lines, int cap, int32_t* d_status4, void* stream);
extern "C" int sslam_lines_set_core_event(sslam_lines* lines, void* hip_event);
extern "C" int sslam_lines_core_guest_form(sslam_lines* lines, int nframes);
extern "C" int sslam_orb_set_gate_event(sslam_orb* orb, void* hip_event);

// vbPrevMatched of SearchForInitialization starts at F1's keypoint positions (src/Tracking.cc:340-342): the first two floats of each record
__global__ __launch_bounds__(256) void k_prev_matched_init(const sslam_keypoint* __restrict__ kp, size_t rows, float2* __restrict__ pm) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < rows) { const float* k = (const float*)(kp + i); pm[i] = make_float2(k[0], k[1]); }
}

namespace {
int copy_threads() {
    static const int n = [] {
        if (const char* e = getenv("SSLAM_BATCH_THREADS")) return std::max(1, atoi(e));
        const unsigned hc = std::thread::hardware_concurrency();
        return (int)std::max(1u, std::min(12u, hc ? hc / 2 : 4u));      // measured with 8, 16, 24 and 48 threads on a 256-thread host: 61.1 / 59.7 / 57.2 / 60.8 k frames per second (profiles/r06e_*) -- what counts is that staging and copy-out run side by side, not the thread count
    }();
    return n;
}
// The host copy threads of the pageable path.  A TaskGroup is one parallel copy (the staging of a chunk, or the copy-out of a chunk's results); several groups may be in
// flight on the pool at once, and the caller waits for the group it needs.
struct TaskGroup {
    std::mutex m; std::condition_variable cv; int left = 0;
    void wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return left == 0; }); }
};
struct CopyPool {
    std::vector<std::thread> th;
    std::mutex m; std::condition_variable cv;
    std::deque<std::pair<TaskGroup*, std::function<void()>>> q;
    bool stop = false;
    explicit CopyPool(int n) {
        for (int i = 0; i < n; ++i) th.emplace_back([this] {
            for (;;) {
                std::pair<TaskGroup*, std::function<void()>> job;
                { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return stop || !q.empty(); }); if (q.empty()) return; job = std::move(q.front()); q.pop_front(); }
                job.second();
                { std::lock_guard<std::mutex> lk(job.first->m); if (--job.first->left == 0) job.first->cv.notify_all(); }
            }
        });
    }
    ~CopyPool() { { std::lock_guard<std::mutex> lk(m); stop = true; } cv.notify_all(); for (auto& t : th) t.join(); }
    // f(a, b) over [0, items) in pieces, asynchronously: g.wait() returns when all of them ran
    template <class F>
    void parallel_for(TaskGroup& g, int items, F f) {
        if (items <= 0) return;
        const int pieces = std::max(1, std::min(items, (int)th.size() * 2)), per = (items + pieces - 1) / pieces;
        int npieces = 0;
        for (int a = 0; a < items; a += per) ++npieces;
        { std::lock_guard<std::mutex> lk(g.m); g.left += npieces; }
        { std::lock_guard<std::mutex> lk(m); for (int a = 0; a < items; a += per) { const int b = std::min(items, a + per); q.emplace_back(&g, [=] { f(a, b); }); } }
        cv.notify_all();
    }
};
struct Slot {
    DevBuf dIn, dKp, dDesc, dN, dKl, dLd, dFn, dNl, dStatus;
    DevBuf dPm, dM12, dNm, dKnnI, dKnnD, dLp, dNlp;      // match stage: vbPrevMatched, vnMatches12, counts, 2-NN, line pairs
    HostPinned hIn, hOut, hStatus;
    hipEvent_t evIn = nullptr, evPoint = nullptr, evLines = nullptr, evOut = nullptr;
    int first = 0, count = 0;            // frames of the chunk in flight
    void release() {
        dIn.release(); dKp.release(); dDesc.release(); dN.release(); dKl.release(); dLd.release(); dFn.release(); dNl.release(); dStatus.release(); dPm.release(); dM12.release(); dNm.release(); dKnnI.release(); dKnnD.release(); dLp.release(); dNlp.release(); hIn.release(); hOut.release(); hStatus.release();
        for (hipEvent_t* e : {&evIn, &evPoint, &evLines, &evOut}) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; }
    }
};
// what a context keeps between calls: staging / device buffers of the two slots, the copy and branch streams, the core event
struct BatchCache {
    Slot slot[2];
    hipStream_t cp = nullptr, cpOut = nullptr, stLines = nullptr;
    hipEvent_t evCore = nullptr;
    CopyPool* pool = nullptr;
    ~BatchCache() {
        delete pool;
        for (auto& s : slot) s.release();
        for (hipStream_t* st : {&cp, &cpOut, &stLines}) { if (*st) (void)hipStreamDestroy(*st); *st = nullptr; }
        if (evCore) (void)hipEventDestroy(evCore);
    }
};
void free_batch_cache(void* p) { delete (BatchCache*)p; }

}  // namespace

namespace {
int batch_impl(const char* fn, sslam_orb* orb, sslam_lines* lines, const uint8_t* images, int n, int w, int h, size_t stride, size_t image_stride, int chunk,
               sslam_keypoint* kp_out, uint8_t* desc_out, int32_t* nkp_out, int cap,
               sslam_keyline* kl_out, uint8_t* ldesc_out, double* linefn_out, int32_t* nl_out, int lcap, const sslam_batch_match* M) {
    sslam_ctx* ctx = orb ? sslam_orb_context(orb) : nullptr;
    if (!orb || !ctx || n < 0 || w <= 0 || h <= 0 || stride < (size_t)w || cap <= 0 || (n > 0 && (!images || !kp_out || !desc_out || !nkp_out)) ||
        (lines && (sslam_lines_context(lines) != ctx || lcap <= 0 || (n > 0 && (!kl_out || !ldesc_out || !linefn_out || !nl_out)))) || (n > 1 && image_stride < stride * (size_t)(h - 1) + (size_t)w) ||
        (M && n > 0 && (!M->init_matches12 || !M->init_nmatches || (M->knn_idx == nullptr) != (M->knn_dist == nullptr) || (lines && (!M->line_pairs || !M->line_npairs))))) {
        set_error("%s: invalid arguments", fn); return SSLAM_ERR_INVALID;
    }
    if (n == 0) return SSLAM_OK;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    const size_t fpx = (size_t)w * h;
    const bool knn = M && M->knn_idx, lmatch = M && lines;
    int C = std::min(n, chunk > 0 ? chunk : 6144);
    if (chunk <= 0) {      // the default follows the core's wave slots, but never asks for more than a third of the free device memory
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
            // per frame: the ORB workspace (pyramid + three candidate planes ~ 4.2 B per pixel), the LSD / LBD workspace (0.64 x 28 B of planes, order list and
            // region spill per pixel, 4 B of Sobel pairs, ~1.4 MB of rectangle / NFA records), the two slots of inputs and outputs
            const size_t perFrame = 6 * fpx + (lines ? 25 * fpx + 1500000 : 0) + 2 * fpx + (size_t)cap * (60 * 3 + (M ? 12 + 20 + (knn ? 32 + 256 : 0) : 0)) + (lines ? (size_t)lcap * 124 * 3 : 0) + 65536;
            const size_t fit = freeB / 3 / std::max<size_t>(perFrame, 1);
            if ((size_t)C > fit) C = (int)std::max<size_t>(fit, 1);      // little free memory (another process on the GPU, very large frames): smaller chunks, down to one frame; only then can an allocation fail
        }
    }
    // host staging layout of one chunk's results (pageable callers)
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    const size_t oKp = take(sizeof(sslam_keypoint) * (size_t)C * cap), oDesc = take(32 * (size_t)C * cap), oN = take(4 * (size_t)C),
                 oKl = take(lines ? sizeof(sslam_keyline) * (size_t)C * lcap : 0), oLd = take(lines ? 32 * (size_t)C * lcap : 0), oFn = take(lines ? 24 * (size_t)C * lcap : 0),
                 oNl = take(lines ? 4 * (size_t)C : 0), oM12 = take(M ? 4 * (size_t)C * cap : 0), oNm = take(M ? 4 * (size_t)C : 0),
                 oKi = take(knn ? 8 * (size_t)C * cap : 0), oKd = take(knn ? 8 * (size_t)C * cap : 0), oLp = take(lmatch ? 8 * (size_t)C * lcap : 0), oNlp = take(lmatch ? 4 * (size_t)C : 0);
    const size_t outBytes = o;
    // pinned (hipHostMalloc / hipHostRegister) caller memory is copied from / to directly; pageable memory goes through the pinned staging
    auto is_pinned = [](const void* q) {
        hipPointerAttribute_t a;
        if (!q || hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }
        return a.type == hipMemoryTypeHost;
    };
    const bool inDirect = stride == (size_t)w && (n == 1 || image_stride == fpx) && is_pinned(images);
    const bool outDirect = is_pinned(kp_out) && is_pinned(desc_out) && is_pinned(nkp_out) &&
                           (!lines || (is_pinned(kl_out) && is_pinned(ldesc_out) && is_pinned(linefn_out) && is_pinned(nl_out))) &&
                           (!M || (is_pinned(M->init_matches12) && is_pinned(M->init_nmatches) && (!knn || (is_pinned(M->knn_idx) && is_pinned(M->knn_dist))) &&
                                   (!lmatch || (is_pinned(M->line_pairs) && is_pinned(M->line_npairs)))));
    if (!ctx->batchCache) { ctx->batchCache = new BatchCache(); ctx->batchCacheFree = free_batch_cache; }
    BatchCache& B = *(BatchCache*)ctx->batchCache;
    Slot* slot = B.slot;
    int rc = SSLAM_OK;
    if ((!B.cp && hipStreamCreateWithFlags(&B.cp, hipStreamNonBlocking) != hipSuccess) || (!B.cpOut && hipStreamCreateWithFlags(&B.cpOut, hipStreamNonBlocking) != hipSuccess) ||
        (!B.stLines && hipStreamCreateWithFlags(&B.stLines, hipStreamNonBlocking) != hipSuccess) || (!B.evCore && hipEventCreateWithFlags(&B.evCore, hipEventDisableTiming) != hipSuccess)) {
        set_error("%s: stream / event creation failed", fn); return SSLAM_ERR_HIP;
    }
    hipStream_t cp = B.cp, cpOut = B.cpOut, stP = ctx->stream, stL = B.stLines;      // H2D and D2H on separate streams: the next chunk's upload must not queue behind this chunk's download
    // Feature arrays hold C + 1 frames: slot 0 is the frame in front of the chunk (the match stage's "previous frame" of the chunk's first
    // frame), the chunk's own frames follow.  Without the match stage the extra frame is simply never read.
    const size_t F1 = (size_t)C + 1;
    for (int i = 0; i < 2; ++i) {
        Slot& s = slot[i];
        s.count = 0;
        if ((rc = s.dIn.ensure(fpx * C)) || (rc = s.dKp.ensure(sizeof(sslam_keypoint) * F1 * cap)) || (rc = s.dDesc.ensure(32 * F1 * cap)) ||
            (rc = s.dN.ensure(4 * F1)) || (rc = s.dStatus.ensure(32)) || (rc = s.hStatus.ensure(32)) || (!inDirect && (rc = s.hIn.ensure(fpx * C))) || (!outDirect && (rc = s.hOut.ensure(outBytes)))) return rc;
        if (lines && ((rc = s.dKl.ensure(sizeof(sslam_keyline) * (size_t)C * lcap)) || (rc = s.dLd.ensure(32 * F1 * lcap)) ||
                      (rc = s.dFn.ensure(24 * (size_t)C * lcap)) || (rc = s.dNl.ensure(4 * F1)))) return rc;
        if (M && ((rc = s.dPm.ensure(8 * (size_t)C * cap)) || (rc = s.dM12.ensure(4 * (size_t)C * cap)) || (rc = s.dNm.ensure(4 * (size_t)C)) ||
                  (knn && ((rc = s.dKnnI.ensure(8 * (size_t)C * cap)) || (rc = s.dKnnD.ensure(8 * (size_t)C * cap)))) ||
                  (lmatch && ((rc = s.dLp.ensure(8 * (size_t)C * lcap)) || (rc = s.dNlp.ensure(4 * (size_t)C)))))) return rc;
        for (hipEvent_t* e : {&s.evIn, &s.evPoint, &s.evLines, &s.evOut})
            if (!*e && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) { set_error("%s: hipEventCreate failed", fn); return SSLAM_ERR_HIP; }
        if (n <= C) break;                 // one chunk: the second slot is never used
    }
    if (!B.pool && !(inDirect && outDirect)) B.pool = new CopyPool(copy_threads());
    CopyPool* pool = B.pool;
    TaskGroup gStage, gDrain[2];                       // the staging of the chunk being submitted; the copy-out of each slot's finished chunk
    int firstStatus = SSLAM_OK;                        // a truncated / unsupported frame does not stop the batch; it is reported at the end
    // results of a finished chunk: pinned staging -> the caller's arrays.  drain() waits for the chunk's D2H, reads its status words and hands the copy to the pool
    // (task group gDrain[slot]); drained() waits for that copy -- it must have finished before the slot's staging buffer receives another chunk's results.
    auto drained = [&](int si) { gDrain[si].wait(); };
    auto drain = [&](int si) -> int {
        Slot& s = slot[si];
        if (s.count == 0) return SSLAM_OK;
        if (hipEventSynchronize(s.evOut) != hipSuccess) { set_error("%s: D2H failed", fn); return SSLAM_ERR_HIP; }
        // per-chunk status words (sslam_orb_batch_status_dev / sslam_lines_batch_status_dev): the conditions the single-frame calls report
        const int* S = s.hStatus.as<int>();
        int status = SSLAM_OK;
        if (lines && S[6]) { set_error("%s: frame %d produced more than 8192 LSD candidate rectangles", fn, s.first + S[7]); status = SSLAM_ERR_UNSUPPORTED; }
        else if (S[0]) { set_error("%s: %d frame(s) of chunk %d.. hold more keypoints than cap %d (first: frame %d)", fn, S[0], s.first, cap, s.first + S[1]); status = SSLAM_ERR_CAPACITY; }
        else if (lines && S[4]) { set_error("%s: %d frame(s) of chunk %d.. hold more lines than lcap %d (first: frame %d)", fn, S[4], s.first, lcap, s.first + S[5]); status = SSLAM_ERR_CAPACITY; }
        if (status != SSLAM_OK && firstStatus == SSLAM_OK) firstStatus = status;
        if (outDirect) { s.count = 0; return SSLAM_OK; }
        const uint8_t* H = s.hOut.as<uint8_t>();
        const size_t f0 = (size_t)s.first;
        pool->parallel_for(gDrain[si], s.count, [=](int a, int b) {      // frames [a, b) of the chunk
            const size_t c = (size_t)(b - a), g = f0 + a;
            std::memcpy(kp_out + g * cap, H + oKp + sizeof(sslam_keypoint) * (size_t)a * cap, sizeof(sslam_keypoint) * c * cap);
            std::memcpy(desc_out + 32 * g * cap, H + oDesc + 32 * (size_t)a * cap, 32 * c * cap);
            std::memcpy(nkp_out + g, H + oN + 4 * (size_t)a, 4 * c);
            if (lines) {
                std::memcpy(kl_out + g * lcap, H + oKl + sizeof(sslam_keyline) * (size_t)a * lcap, sizeof(sslam_keyline) * c * lcap);
                std::memcpy(ldesc_out + 32 * g * lcap, H + oLd + 32 * (size_t)a * lcap, 32 * c * lcap);
                std::memcpy(linefn_out + 3 * g * lcap, H + oFn + 24 * (size_t)a * lcap, 24 * c * lcap);
                std::memcpy(nl_out + g, H + oNl + 4 * (size_t)a, 4 * c);
            }
            if (M) {
                std::memcpy(M->init_matches12 + g * cap, H + oM12 + 4 * (size_t)a * cap, 4 * c * cap);
                std::memcpy(M->init_nmatches + g, H + oNm + 4 * (size_t)a, 4 * c);
                if (knn) { std::memcpy(M->knn_idx + 2 * g * cap, H + oKi + 8 * (size_t)a * cap, 8 * c * cap); std::memcpy(M->knn_dist + 2 * g * cap, H + oKd + 8 * (size_t)a * cap, 8 * c * cap); }
                if (lmatch) { std::memcpy(M->line_pairs + 2 * g * lcap, H + oLp + 8 * (size_t)a * lcap, 8 * c * lcap); std::memcpy(M->line_npairs + g, H + oNlp + 4 * (size_t)a, 4 * c); }
            }
        });
        s.count = 0;
        return SSLAM_OK;
    };
    const bool twoStreams = lines != nullptr && getenv("SSLAM_BATCH_ONE_STREAM") == nullptr;
    if (lines) (void)sslam_lines_set_core_event(lines, twoStreams ? (void*)B.evCore : nullptr);
    int k = 0;
    for (int f0 = 0; f0 < n && rc == SSLAM_OK; f0 += C, ++k) {
        Slot& s = slot[k & 1];
        const int c = std::min(C, n - f0);
        const uint8_t* hin = images + (size_t)f0 * fpx;
        if (!inDirect) {      // this chunk's frames -> the slot's pinned input buffer (its previous upload, chunk k-2, finished long ago: the kernels of k-2 ran behind it and its results are in)
            uint8_t* stage = s.hIn.as<uint8_t>();
            if (s.count && hipEventSynchronize(s.evIn) != hipSuccess) { set_error("%s: H2D failed", fn); rc = SSLAM_ERR_HIP; break; }
            pool->parallel_for(gStage, c, [=](int a, int b) {               // tight rows in the staging buffer
                for (int i = a; i < b; ++i) {
                    const uint8_t* src = images + (size_t)(f0 + i) * image_stride;
                    if (stride == (size_t)w) std::memcpy(stage + i * fpx, src, fpx);
                    else for (int y = 0; y < h; ++y) std::memcpy(stage + i * fpx + (size_t)y * w, src + (size_t)y * stride, w);
                }
            });
            hin = stage;
        }
        if ((rc = drain(k & 1))) { gStage.wait(); break; }                 // the slot's previous chunk (k-2): its copy-out runs on the pool beside the staging above
        gStage.wait();
        hipStream_t stLn = twoStreams ? stL : stP;
        if (hipMemcpyAsync(s.dIn.p, hin, fpx * c, hipMemcpyHostToDevice, cp) != hipSuccess || hipEventRecord(s.evIn, cp) != hipSuccess ||
            hipStreamWaitEvent(stP, s.evIn, 0) != hipSuccess || (twoStreams && hipStreamWaitEvent(stL, s.evIn, 0) != hipSuccess)) { set_error("%s: H2D failed", fn); rc = SSLAM_ERR_HIP; break; }
        // the chunk's frames start one frame into the feature arrays
        sslam_keypoint* dKp = s.dKp.as<sslam_keypoint>() + cap; uint8_t* dDesc = s.dDesc.as<uint8_t>() + 32 * (size_t)cap; int32_t* dN = s.dN.as<int32_t>() + 1;
        uint8_t* dLd = lines ? s.dLd.as<uint8_t>() + 32 * (size_t)lcap : nullptr; int32_t* dNl = lines ? s.dNl.as<int32_t>() + 1 : nullptr;
        if (M) {
            // frame 0 of the array = the frame in front of this chunk: the last frame of the chunk before (the other slot, same streams: ordered
            // behind its extraction), or no frame at all for the first chunk of the call (count 0: nothing matches)
            bool ok = true;
            if (k == 0) {
                ok = hipMemsetAsync(s.dN.p, 0, 4, stP) == hipSuccess && (!lines || hipMemsetAsync(s.dNl.p, 0, 4, stLn) == hipSuccess);
            } else {
                const Slot& q = slot[(k + 1) & 1]; const size_t last = (size_t)C;      // every chunk but the last one is full
                ok = hipMemcpyAsync(s.dKp.p, q.dKp.as<sslam_keypoint>() + last * cap, sizeof(sslam_keypoint) * (size_t)cap, hipMemcpyDeviceToDevice, stP) == hipSuccess &&
                     hipMemcpyAsync(s.dDesc.p, q.dDesc.as<uint8_t>() + 32 * last * cap, 32 * (size_t)cap, hipMemcpyDeviceToDevice, stP) == hipSuccess &&
                     hipMemcpyAsync(s.dN.p, q.dN.as<int32_t>() + last, 4, hipMemcpyDeviceToDevice, stP) == hipSuccess;
                if (ok && lines) ok = hipMemcpyAsync(s.dLd.p, q.dLd.as<uint8_t>() + 32 * last * lcap, 32 * (size_t)lcap, hipMemcpyDeviceToDevice, stLn) == hipSuccess &&
                                      hipMemcpyAsync(s.dNl.p, q.dNl.as<int32_t>() + last, 4, hipMemcpyDeviceToDevice, stLn) == hipSuccess;
            }
            if (!ok) { set_error("%s: carrying the previous frame failed", fn); rc = SSLAM_ERR_HIP; break; }
        }
        // line branch first: its call records the core event; the point branch then waits for that event and runs under the latency-bound core
        if (lines && (rc = sslam_lines_extract_batch_dev(lines, s.dIn.as<uint8_t>(), w, h, (size_t)w, fpx, c, s.dKl.as<sslam_keyline>(), dLd, s.dFn.as<double>(), dNl, lcap, stLn))) break;
        if (lines && (rc = sslam_lines_batch_status_dev(lines, lcap, s.dStatus.as<int32_t>() + 4, stLn))) break;
        if (lmatch && (rc = sslam_line_match_batch_dev(ctx, s.dLd.as<uint8_t>(), s.dNl.as<int32_t>(), dLd, dNl, lcap, c, M->line_gate_scale, M->line_ratio_mode,
                                                       s.dLp.as<int32_t>(), s.dNlp.as<int32_t>(), stLn))) break;
        // guest form of the core (chunks above 16 workgroups per CU): the pyramid goes ahead, FAST .. matching wait for the core event inside the ORB call; otherwise the whole point branch waits
        const bool guestForm = twoStreams && sslam_lines_core_guest_form(lines, c) != 0;
        (void)sslam_orb_set_gate_event(orb, guestForm ? (void*)B.evCore : nullptr);
        if (twoStreams && (hipEventRecord(s.evLines, stL) != hipSuccess || (!guestForm && hipStreamWaitEvent(stP, B.evCore, 0) != hipSuccess))) { set_error("%s: event failed", fn); rc = SSLAM_ERR_HIP; break; }
        if ((rc = sslam_orb_extract_batch_dev(orb, s.dIn.as<uint8_t>(), w, h, (size_t)w, fpx, c, dKp, dDesc, dN, cap, stP))) break;
        if ((rc = sslam_orb_batch_status_dev(orb, cap, s.dStatus.as<int32_t>(), stP))) break;
        if (M) {      // previous frame = F1 (query), current frame = F2 (train), as Tracking::MonocularInitialization calls it (src/Tracking.cc:330-345)
            const size_t rows = (size_t)c * cap;
            hipLaunchKernelGGL(k_prev_matched_init, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stP, s.dKp.as<sslam_keypoint>(), rows, s.dPm.as<float2>());
            if (hipGetLastError() != hipSuccess) { set_error("%s: launch failed", fn); rc = SSLAM_ERR_HIP; break; }
            if ((rc = sslam_orb_search_for_initialization_batch_dev(ctx, s.dKp.as<sslam_keypoint>(), s.dDesc.as<uint8_t>(), s.dN.as<int32_t>(), dKp, dDesc, dN, cap, c, s.dPm.as<float>(),
                                                                    s.dM12.as<int32_t>(), s.dNm.as<int32_t>(), M->window_size, M->nnratio, M->check_orientation, M->bounds, stP))) break;
            if (knn && (rc = sslam_hamming_knn2_batch_dev(ctx, s.dDesc.as<uint8_t>(), s.dN.as<int32_t>(), dDesc, dN, cap, c, s.dKnnI.as<int32_t>(), s.dKnnD.as<int32_t>(), stP))) break;
        }
        if (hipEventRecord(s.evPoint, stP) != hipSuccess || hipStreamWaitEvent(cpOut, s.evPoint, 0) != hipSuccess ||
            (twoStreams && hipStreamWaitEvent(cpOut, s.evLines, 0) != hipSuccess)) { set_error("%s: event failed", fn); rc = SSLAM_ERR_HIP; break; }
        // the next chunk's upload into the OTHER slot may start at once; this slot's input is overwritten only two chunks later
        drained(k & 1);                                                     // (the slot's result staging is about to be named as a D2H target again)
        uint8_t* H = outDirect ? nullptr : s.hOut.as<uint8_t>();
        const size_t g = (size_t)f0;
        bool ok = true;
        auto down = [&](void* direct, size_t stageOff, const void* dev, size_t bytes) {
            if (ok && bytes) ok = hipMemcpyAsync(outDirect ? direct : (void*)(H + stageOff), dev, bytes, hipMemcpyDeviceToHost, cpOut) == hipSuccess;
        };
        down(kp_out + g * cap, oKp, dKp, sizeof(sslam_keypoint) * (size_t)c * cap);
        down(desc_out + 32 * g * cap, oDesc, dDesc, 32 * (size_t)c * cap);
        down(nkp_out + g, oN, dN, 4 * (size_t)c);
        if (ok) ok = hipMemcpyAsync(s.hStatus.p, s.dStatus.p, 32, hipMemcpyDeviceToHost, cpOut) == hipSuccess;
        if (lines) {
            down(kl_out + g * lcap, oKl, s.dKl.p, sizeof(sslam_keyline) * (size_t)c * lcap);
            down(ldesc_out + 32 * g * lcap, oLd, dLd, 32 * (size_t)c * lcap);
            down(linefn_out + 3 * g * lcap, oFn, s.dFn.p, 24 * (size_t)c * lcap);
            down(nl_out + g, oNl, dNl, 4 * (size_t)c);
        }
        if (M) {
            down(M->init_matches12 + g * cap, oM12, s.dM12.p, 4 * (size_t)c * cap);
            down(M->init_nmatches + g, oNm, s.dNm.p, 4 * (size_t)c);
            if (knn) { down(M->knn_idx + 2 * g * cap, oKi, s.dKnnI.p, 8 * (size_t)c * cap); down(M->knn_dist + 2 * g * cap, oKd, s.dKnnD.p, 8 * (size_t)c * cap); }
            if (lmatch) { down(M->line_pairs + 2 * g * lcap, oLp, s.dLp.p, 8 * (size_t)c * lcap); down(M->line_npairs + g, oNlp, s.dNlp.p, 4 * (size_t)c); }
        }
        if (!ok || hipEventRecord(s.evOut, cpOut) != hipSuccess) { set_error("%s: D2H failed", fn); rc = SSLAM_ERR_HIP; break; }
        s.first = f0; s.count = c;
    }
    if (rc == SSLAM_OK) rc = drain(k & 1);          // older chunk first
    if (rc == SSLAM_OK) rc = drain((k + 1) & 1);
    gStage.wait(); drained(0); drained(1);           // (also on the error paths: no task may outlive the caller's buffers)
    (void)hipStreamSynchronize(cp); (void)hipStreamSynchronize(cpOut); (void)hipStreamSynchronize(stL); (void)hipStreamSynchronize(stP);
    if (lines) (void)sslam_lines_set_core_event(lines, nullptr);
    (void)sslam_orb_set_gate_event(orb, nullptr);
    for (int i = 0; i < 2; ++i) slot[i].count = 0;
    return rc != SSLAM_OK ? rc : firstStatus;
}
}  // namespace

extern "C" int sslam_frontend_batch(sslam_orb* orb, sslam_lines* lines, const uint8_t* images, int n, int w, int h, size_t stride, size_t image_stride, int chunk,
                                    sslam_keypoint* kp_out, uint8_t* desc_out, int32_t* nkp_out, int cap,
                                    sslam_keyline* kl_out, uint8_t* ldesc_out, double* linefn_out, int32_t* nl_out, int lcap) {
    return batch_impl("sslam_frontend_batch", orb, lines, images, n, w, h, stride, image_stride, chunk, kp_out, desc_out, nkp_out, cap, kl_out, ldesc_out, linefn_out, nl_out, lcap, nullptr);
}

extern "C" int sslam_frontend_batch_match(sslam_orb* orb, sslam_lines* lines, const uint8_t* images, int n, int w, int h, size_t stride, size_t image_stride, int chunk,
                                          sslam_keypoint* kp_out, uint8_t* desc_out, int32_t* nkp_out, int cap,
                                          sslam_keyline* kl_out, uint8_t* ldesc_out, double* linefn_out, int32_t* nl_out, int lcap, const sslam_batch_match* match) {
    if (!match) { set_error("sslam_frontend_batch_match: invalid arguments"); return SSLAM_ERR_INVALID; }
    return batch_impl("sslam_frontend_batch_match", orb, lines, images, n, w, h, stride, image_stride, chunk, kp_out, desc_out, nkp_out, cap, kl_out, ldesc_out, linefn_out, nl_out, lcap, match);
}

// Releases the staging buffers, streams and events sslam_frontend_batch keeps per context between calls (they are also released with the context).
extern "C" int sslam_frontend_batch_release(sslam_ctx* ctx) {
    if (!ctx) return SSLAM_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    if (ctx->batchCache && ctx->batchCacheFree) ctx->batchCacheFree(ctx->batchCache);
    ctx->batchCache = nullptr;
    return SSLAM_OK;
}
