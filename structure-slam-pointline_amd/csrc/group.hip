// Multi-GPU batch mode (SURVEY.md §8(b),(e); north_star: "shards independent images across the 8 GPUs of one node with RCCL over
// xGMI only for the final keypoint/line gather").  Frames are independent units, so the path has exactly one exchange step: the
// compacted per-frame records of every GPU go to the root GPU.  This file holds
//   * the record stream (pack kernels, host-side unpack),
//   * the group handle in its two forms (one process driving G devices / one process per GPU) over RCCL bound with dlopen,
//   * the gather (sizes first, then grouped ncclSend / ncclRecv),
//   * sslam_frontend_batch_sharded, the host-buffer batch entry point over all GPUs of a single-process group.
// xGMI is point to point (7 links per GPU): a gather to one root is bound by the root's ingest, G-1 links x ~50 GB/s effective.  At
// ~80 KB per 640x480 frame that is >4 M frames/s of ingest, two orders of magnitude above what eight GPUs extract; the exchange is
// latency, not bandwidth.
#include "common.h"
#include <atomic>
#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <thread>

using namespace sslam;

namespace {

// ------------------------------------------------------------------ RCCL, bound at run time
typedef struct ncclComm* ncclComm_t;
struct NcclUid { char internal[128]; };
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(NcclUid*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, NcclUid, int) = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
constexpr int kNcclUint8 = 1, kNcclUint64 = 5;      // ncclDataType_t (rccl.h)

Rccl* rccl_real() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy that is already in the process (PyTorch ships its own librccl.so.1) wins: two RCCL instances in one process would not share state
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        R.h = h;
        R.GetUniqueId = (decltype(R.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        R.CommInitRank = (decltype(R.CommInitRank))dlsym(h, "ncclCommInitRank");
        R.CommInitAll = (decltype(R.CommInitAll))dlsym(h, "ncclCommInitAll");
        R.CommDestroy = (decltype(R.CommDestroy))dlsym(h, "ncclCommDestroy");
        R.GroupStart = (decltype(R.GroupStart))dlsym(h, "ncclGroupStart");
        R.GroupEnd = (decltype(R.GroupEnd))dlsym(h, "ncclGroupEnd");
        R.Send = (decltype(R.Send))dlsym(h, "ncclSend");
        R.Recv = (decltype(R.Recv))dlsym(h, "ncclRecv");
        R.AllGather = (decltype(R.AllGather))dlsym(h, "ncclAllGather");
        R.GetErrorString = (decltype(R.GetErrorString))dlsym(h, "ncclGetErrorString");
        if (!R.GetUniqueId || !R.CommInitRank || !R.CommInitAll || !R.CommDestroy || !R.GroupStart || !R.GroupEnd || !R.Send || !R.Recv || !R.AllGather) R.h = nullptr;
    });
    return R.h ? &R : nullptr;
}

#ifdef SSLAM_TESTING      // libsslam_frontend_testing.so only
// ------------------------------------------------------------------ sslam_testing_use_rccl_standin(1) (include/sslam_testing.h): an in-process stand-in for the RCCL entry points
// N > 1 has never run on hardware here (one GPU per box), so the group code's multi-member paths -- a host thread per device, uneven tails,
// the collective error agreement, grouped send / receive to the root -- had no execution at all.  With this table selected at group creation
// the "devices" of a group are contexts (streams) of whatever GPUs are visible, dealt round-robin, and the collectives are host-mediated
// device-to-device copies with NCCL's matching rules: a send to p pairs with p's receive from the sender in posting order, calls between
// GroupStart / GroupEnd are issued together, an all-gather is a rendezvous of all ranks.  Same process only (ranks are threads).  It moves
// real bytes between real device buffers through the same code paths; it says nothing about xGMI -- the scaling run stays the driver's.
struct FakeWorld {
    int nranks = 0, refs = 0;
    std::mutex mu; std::condition_variable cv;
    struct Msg { unsigned long id; int src, dst; const void* ptr; size_t bytes; bool taken; };
    std::vector<Msg> box;                           // posted sends, in posting order (entries are named by id: the vector shifts when a sender clears its own)
    unsigned long nextId = 1;
    std::vector<const void*> agPtr; int agArrived = 0, agLeft = 0, agGen = 0;
};
struct FakeComm { FakeWorld* w; int rank; };
struct FakeOp { int kind; const void* sptr; void* rptr; size_t bytes; int peer; FakeComm* c; hipStream_t st; unsigned long id; };      // 0 send, 1 recv
// every wait of the stand-in is bounded: a protocol error of the group code must fail a test, not hang the suite
constexpr std::chrono::seconds kFakeWait(60);
thread_local int tFakeDepth = 0;
thread_local std::vector<FakeOp> tFakeOps;
std::mutex gFakeMu;
std::vector<std::pair<NcclUid, FakeWorld*>> gFakeWorlds;      // worlds being assembled by ncclCommInitRank, keyed by unique id
int gFakeIdCounter = 0;

size_t fake_dtype_bytes(int dt) { return dt == kNcclUint64 ? 8 : 1; }
int fake_flush_ops(std::vector<FakeOp>& ops);
int fake_flush() {
    std::vector<FakeOp> ops; ops.swap(tFakeOps);
    const int rc = fake_flush_ops(ops);
    if (rc != 0)      // an error leaves nothing behind: the rank's own posted sends (device pointers that may die with the caller's buffers) are withdrawn, or a later receive could match them
        for (FakeOp& o : ops) if (o.kind == 0 && o.id) {
            std::lock_guard<std::mutex> lk(o.c->w->mu);
            for (size_t i = 0; i < o.c->w->box.size(); ++i) if (o.c->w->box[i].id == o.id) { o.c->w->box.erase(o.c->w->box.begin() + i); break; }
            o.c->w->cv.notify_all();
        }
    return rc;
}
int fake_flush_ops(std::vector<FakeOp>& ops) {
    // sends first: publish (the data must be final: drain the sender's stream), then receives (wait for the partner's publication, copy,
    // acknowledge), then wait until every own send was taken -- a rank that posts both directions in one group cannot block itself
    for (FakeOp& o : ops) if (o.kind == 0) {
        if (hipStreamSynchronize(o.st) != hipSuccess) return 1;
        std::lock_guard<std::mutex> lk(o.c->w->mu);
        o.id = o.c->w->nextId++;
        o.c->w->box.push_back({o.id, o.c->rank, o.peer, o.sptr, o.bytes, false});
        o.c->w->cv.notify_all();
    }
    auto find = [](FakeWorld* w, unsigned long id) -> FakeWorld::Msg* { for (auto& m : w->box) if (m.id == id) return &m; return nullptr; };
    for (FakeOp& o : ops) if (o.kind == 1) {
        FakeWorld* w = o.c->w; const void* src = nullptr; unsigned long id = 0; size_t bytes = 0;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            // the oldest untaken send of that peer to this rank (receives of one rank are issued by one thread, one after the other)
            if (!w->cv.wait_for(lk, kFakeWait, [&] { for (auto& m : w->box) if (!m.taken && m.src == o.peer && m.dst == o.c->rank) { id = m.id; return true; } return false; })) return 4;
            FakeWorld::Msg* m = find(w, id);
            src = m->ptr; bytes = m->bytes;
        }
        if (bytes != o.bytes) return 2;               // NCCL would hang or corrupt on mismatched sizes: here it is an error
        // stream-ordered on the receiver's stream like the real receive, then drained: the sender may reuse its buffer once this returns
        if (o.bytes && hipMemcpyAsync(o.rptr, src, o.bytes, hipMemcpyDeviceToDevice, o.st) != hipSuccess) return 1;
        if (hipStreamSynchronize(o.st) != hipSuccess) return 1;
        std::lock_guard<std::mutex> lk(w->mu);
        if (FakeWorld::Msg* m = find(w, id)) m->taken = true;
        w->cv.notify_all();
    }
    for (FakeOp& o : ops) if (o.kind == 0) {
        FakeWorld* w = o.c->w;
        std::unique_lock<std::mutex> lk(w->mu);
        if (!w->cv.wait_for(lk, kFakeWait, [&] { FakeWorld::Msg* m = find(w, o.id); return !m || m->taken; })) return 4;
        for (size_t i = 0; i < w->box.size(); ++i) if (w->box[i].id == o.id) { w->box.erase(w->box.begin() + i); break; }
    }
    return 0;
}
int fakeGetUniqueId(NcclUid* u) { std::lock_guard<std::mutex> lk(gFakeMu); memset(u, 0, sizeof(*u)); snprintf(u->internal, sizeof(u->internal), "sslam-fake-rccl-%d", ++gFakeIdCounter); return 0; }
int fakeCommInitAll(ncclComm_t* comms, int n, const int*) {
    FakeWorld* w = new FakeWorld(); w->nranks = n; w->refs = n; w->agPtr.assign(n, nullptr);
    for (int r = 0; r < n; ++r) comms[r] = (ncclComm_t) new FakeComm{w, r};
    return 0;
}
int fakeCommInitRank(ncclComm_t* comm, int n, NcclUid id, int rank) {
    std::lock_guard<std::mutex> lk(gFakeMu);
    FakeWorld* w = nullptr;
    for (auto& e : gFakeWorlds) if (memcmp(e.first.internal, id.internal, sizeof(id.internal)) == 0) w = e.second;
    if (!w) { w = new FakeWorld(); w->nranks = n; w->agPtr.assign(n, nullptr); gFakeWorlds.push_back({id, w}); }
    if (w->nranks != n || rank < 0 || rank >= n) return 3;
    ++w->refs;
    *comm = (ncclComm_t) new FakeComm{w, rank};
    return 0;
}
int fakeCommDestroy(ncclComm_t c_) {
    FakeComm* c = (FakeComm*)c_; if (!c) return 0;
    std::lock_guard<std::mutex> lk(gFakeMu);
    if (--c->w->refs == 0) {
        for (size_t i = 0; i < gFakeWorlds.size(); ++i) if (gFakeWorlds[i].second == c->w) { gFakeWorlds.erase(gFakeWorlds.begin() + i); break; }
        delete c->w;
    }
    delete c; return 0;
}
int fakeGroupStart() { ++tFakeDepth; return 0; }
int fakeGroupEnd() { if (--tFakeDepth > 0) return 0; tFakeDepth = 0; return fake_flush(); }
int fakeSend(const void* p, size_t count, int dt, int peer, ncclComm_t c, hipStream_t st) {
    tFakeOps.push_back({0, p, nullptr, count * fake_dtype_bytes(dt), peer, (FakeComm*)c, st, 0});
    return tFakeDepth ? 0 : fake_flush();
}
int fakeRecv(void* p, size_t count, int dt, int peer, ncclComm_t c, hipStream_t st) {
    tFakeOps.push_back({1, nullptr, p, count * fake_dtype_bytes(dt), peer, (FakeComm*)c, st, 0});
    return tFakeDepth ? 0 : fake_flush();
}
int fakeAllGather(const void* sp, void* rp, size_t count, int dt, ncclComm_t c_, hipStream_t st) {
    FakeComm* c = (FakeComm*)c_; FakeWorld* w = c->w; const size_t bytes = count * fake_dtype_bytes(dt);
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    std::vector<const void*> ptrs;
    {
        std::unique_lock<std::mutex> lk(w->mu);
        if (!w->cv.wait_for(lk, kFakeWait, [&] { return w->agLeft == 0; })) return 4;              // the previous round has been left by everybody
        const int gen = w->agGen;
        w->agPtr[c->rank] = sp;
        if (++w->agArrived == w->nranks) { w->agLeft = w->nranks; w->agArrived = 0; ++w->agGen; w->cv.notify_all(); }
        else if (!w->cv.wait_for(lk, kFakeWait, [&] { return w->agGen != gen; })) return 4;
        ptrs = w->agPtr;
    }
    int rc = 0;
    for (int r = 0; r < w->nranks && !rc; ++r) if (bytes && hipMemcpyAsync((char*)rp + (size_t)r * bytes, ptrs[r], bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = 1;
    if (hipStreamSynchronize(st) != hipSuccess) rc = 1;
    std::unique_lock<std::mutex> lk(w->mu);
    if (--w->agLeft == 0) w->cv.notify_all();
    if (!w->cv.wait_for(lk, kFakeWait, [&] { return w->agLeft == 0; })) return 4;                  // nobody's send buffer is reused before everybody has copied it
    return rc;
}
const char* fakeGetErrorString(int e) { return e == 4 ? "fake rccl: a peer did not show up within 60 s" : e == 2 ? "fake rccl: send / receive sizes differ" : e == 3 ? "fake rccl: inconsistent communicator arguments" : "fake rccl: HIP error"; }
Rccl* rccl_fake() {
    static Rccl F;
    static std::once_flag once;
    std::call_once(once, [] {
        F.h = (void*)&F; F.GetUniqueId = fakeGetUniqueId; F.CommInitRank = fakeCommInitRank; F.CommInitAll = fakeCommInitAll; F.CommDestroy = fakeCommDestroy;
        F.GroupStart = fakeGroupStart; F.GroupEnd = fakeGroupEnd; F.Send = fakeSend; F.Recv = fakeRecv; F.AllGather = fakeAllGather; F.GetErrorString = fakeGetErrorString;
    });
    return &F;
}
// selected by a TEST entry point only (sslam_testing_use_rccl_standin, include/sslam_testing.h) -- no environment variable changes which library a product group binds
std::atomic<int> gStandinRequested{0};
bool fake_rccl_requested() { return gStandinRequested.load() != 0; }
#else       // the product library holds no stand-in: a group binds librccl or fails
bool fake_rccl_requested() { return false; }
Rccl* rccl_fake() { return nullptr; }
#endif      // SSLAM_TESTING
// the table a NEW group binds (kept in the group: a process may hold real and stand-in groups side by side)
Rccl* rccl() { return fake_rccl_requested() ? rccl_fake() : rccl_real(); }
#define SSLAM_NCCL(api, expr)                                                                                         \
    do {                                                                                                              \
        int _r = (expr);                                                                                              \
        if (_r != 0) {                                                                                                \
            sslam::set_error("%s failed: %s (rccl status %d)", #expr, (api) && (api)->GetErrorString ? (api)->GetErrorString(_r) : "rccl error", _r);      \
            return SSLAM_ERR_HIP;                                                                                     \
        }                                                                                                             \
    } while (0)

// ------------------------------------------------------------------ record stream
__host__ __device__ inline unsigned record_bytes(int nkp, int nl) {
    return (unsigned)((16 + nkp * (28 + 32) + nl * (68 + 32 + 24) + 15) & ~15);
}

// one workgroup: exclusive scan of the record sizes -> offsets[nframes + 1]; total (or UINT64_MAX when it exceeds the capacity)
__global__ __launch_bounds__(1024) void k_record_offsets(const int* __restrict__ nkp, const int* __restrict__ nl, int nframes, int cap, int lcap,
                                                         unsigned long long* __restrict__ offsets, unsigned long long capacity,
                                                         unsigned long long* __restrict__ total) {
    __shared__ unsigned long long part[1024];
    const int t = threadIdx.x, per = (nframes + 1023) / 1024;
    const int lo = min(t * per, nframes), hi = min(lo + per, nframes);
    unsigned long long s = 0;
    for (int i = lo; i < hi; ++i) s += record_bytes(min(nkp[i], cap), nl ? min(nl[i], lcap) : 0);
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const unsigned long long v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned long long run = part[t] - s;
    for (int i = lo; i < hi; ++i) { offsets[i] = run; run += record_bytes(min(nkp[i], cap), nl ? min(nl[i], lcap) : 0); }
    if (t == 1023) { offsets[nframes] = part[1023]; *total = part[1023] <= capacity ? part[1023] : ~0ull; }
}

// one workgroup per frame: header + the five segments, dword copies (every segment is a multiple of four bytes and starts on one)
__global__ __launch_bounds__(256) void k_record_copy(int frame0, int frameStep, const unsigned* __restrict__ kp, const unsigned* __restrict__ desc,
                                                     const int* __restrict__ nkp, int cap, const unsigned* __restrict__ kl, const unsigned* __restrict__ ldesc,
                                                     const unsigned* __restrict__ linefn, const int* __restrict__ nl, int lcap,
                                                     const unsigned long long* __restrict__ offsets, int nframes, unsigned* __restrict__ out,
                                                     const unsigned long long* __restrict__ total) {
    if (*total == ~0ull) return;
    const int b = blockIdx.x, t = threadIdx.x;
    const int n = min(nkp[b], cap), m = nl ? min(nl[b], lcap) : 0;
    unsigned* o = out + offsets[b] / 4;
    const unsigned bytes = record_bytes(n, m);
    if (t < 4) o[t] = t == 0 ? (unsigned)n : t == 1 ? (unsigned)m : t == 2 ? (unsigned)(frame0 + b * frameStep) : bytes;
    o += 4;
    const unsigned* src[5] = {kp + (size_t)b * cap * 7, desc + (size_t)b * cap * 8, kl ? kl + (size_t)b * lcap * 17 : nullptr,
                              ldesc ? ldesc + (size_t)b * lcap * 8 : nullptr, linefn ? linefn + (size_t)b * lcap * 6 : nullptr};
    const int words[5] = {n * 7, n * 8, m * 17, m * 8, m * 6};
    for (int s = 0; s < 5; ++s) {
        for (int i = t; i < words[s]; i += 256) o[i] = src[s][i];
        o += words[s];
    }
    const int tail = (int)(bytes / 4) - 4 - (words[0] + words[1] + words[2] + words[3] + words[4]);      // alignment padding: defined bytes
    if (t < tail) o[t] = 0;
}

struct Barrier {      // reusable thread barrier (the per-device host threads of a single-process group)
    std::mutex mu; std::condition_variable cv; int count = 0, waiting = 0, gen = 0;
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        const int g = gen;
        if (++waiting == count) { waiting = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return g != gen; });
    }
};

struct Member {       // one GPU of a single-process group
    sslam_ctx* ctx = nullptr; sslam_orb* orb = nullptr; sslam_lines* lines = nullptr;
    ncclComm_t comm = nullptr;
    Rccl* api = nullptr;                    // the table the communicator was made with (real RCCL or the in-process stand-in)
    DevBuf dIn, dKp, dDesc, dN, dKl, dLd, dFn, dNl, dSend, dTotal, dStatus, dRecv;
    HostPinned hTotal, hRecv;
};

}  // namespace

struct sslam_group {
    int nranks = 1, rank = 0, device = 0;
    bool singleProcess = true;
    Rccl* api = nullptr;                   // real RCCL, or the in-process stand-in (sslam_testing_use_rccl_standin(1) when the group was created)
    std::vector<Member> mem;               // single-process: one per device; rank form: one
    sslam_frontend_params params{};
    bool haveParams = false;
    DevBuf dSizes;                         // rank form: nranks x uint64 (all-gathered lengths)
    HostPinned hSizes;
    hipStream_t stream = nullptr;          // rank form: internal stream
    std::mutex mu;
};

extern "C" uint64_t sslam_record_stream_capacity(int nframes, int cap, int lcap) {
    return (uint64_t)std::max(nframes, 0) * record_bytes(std::max(cap, 0), std::max(lcap, 0));
}

extern "C" int sslam_pack_records_dev(sslam_ctx* ctx, int nframes, int frame0, int frame_step,
                                      const sslam_keypoint* d_kp, const uint8_t* d_desc, const int32_t* d_nkp, int cap,
                                      const sslam_keyline* d_kl, const uint8_t* d_ldesc, const double* d_linefn, const int32_t* d_nl, int lcap,
                                      uint8_t* d_out, uint64_t out_capacity, uint64_t* d_total_bytes, void* stream_) {
    const bool lines = d_kl != nullptr;
    if (!ctx || nframes <= 0 || !d_kp || !d_desc || !d_nkp || cap <= 0 || !d_out || !d_total_bytes ||
        (lines && (!d_ldesc || !d_linefn || !d_nl || lcap <= 0))) { set_error("sslam_pack_records_dev: invalid arguments"); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);      // the whole call: launches included (every entry point serialises on the context)
    SSLAM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
    int rc;
    // the per-frame offsets live in a buffer that belongs to the STREAM of the call (a small ring of them per context): packs on different
    // streams never share one, and a buffer is only ever freed (to grow) after its own stream has drained
    DevBuf* offBuf = nullptr;
    for (int i = 0; i < 4 && !offBuf; ++i) if (ctx->recordOffsetsStream[i] == (void*)st && ctx->recordOffsets[i].p) offBuf = &ctx->recordOffsets[i];
    if (!offBuf) {
        int slot = -1;
        for (int i = 0; i < 4 && slot < 0; ++i) if (!ctx->recordOffsets[i].p) slot = i;
        if (slot < 0) {                                      // more than four streams: recycle the least recently used slot once its work is done
            slot = 0;
            for (int i = 1; i < 4; ++i) if (ctx->recordOffsetsUse[i] < ctx->recordOffsetsUse[slot]) slot = i;
            // the remembered handle may belong to a stream the caller has destroyed since (or to a new stream that reuses the handle): a failed
            // synchronise is not an error of this call -- drain the device instead, after which nothing can still read the buffer
            if (hipStreamSynchronize((hipStream_t)ctx->recordOffsetsStream[slot]) != hipSuccess) { (void)hipGetLastError(); SSLAM_HIP(hipDeviceSynchronize()); }
        }
        ctx->recordOffsetsStream[slot] = (void*)st; offBuf = &ctx->recordOffsets[slot];
    }
    ctx->recordOffsetsUse[offBuf - ctx->recordOffsets] = ++ctx->recordOffsetsClock;
    if ((size_t)(nframes + 1) * 8 > offBuf->cap) SSLAM_HIP(hipStreamSynchronize(st));      // a growing buffer is freed first: nothing may still read it
    if ((rc = offBuf->ensure(sizeof(unsigned long long) * ((size_t)nframes + 1)))) return rc;
    unsigned long long* off = offBuf->as<unsigned long long>();
    hipLaunchKernelGGL(k_record_offsets, dim3(1), dim3(1024), 0, st, d_nkp, lines ? d_nl : nullptr, nframes, cap, lcap, off, (unsigned long long)out_capacity,
                       (unsigned long long*)d_total_bytes);
    hipLaunchKernelGGL(k_record_copy, dim3(nframes), dim3(256), 0, st, frame0, frame_step, (const unsigned*)d_kp, (const unsigned*)d_desc, d_nkp, cap,
                       (const unsigned*)d_kl, (const unsigned*)d_ldesc, (const unsigned*)d_linefn, lines ? d_nl : nullptr, lcap, off, nframes, (unsigned*)d_out,
                       (const unsigned long long*)d_total_bytes);
    SSLAM_HIP(hipGetLastError());
    return SSLAM_OK;
}

extern "C" int sslam_unpack_records(const uint8_t* stream, uint64_t bytes, int nframes,
                                    sslam_keypoint* kp_out, uint8_t* desc_out, int32_t* nkp_out, int cap,
                                    sslam_keyline* kl_out, uint8_t* ldesc_out, double* linefn_out, int32_t* nl_out, int lcap, int* nrecords_out) {
    if ((!stream && bytes) || nframes < 0 || !kp_out || !desc_out || !nkp_out || cap <= 0) { set_error("sslam_unpack_records: invalid arguments"); return SSLAM_ERR_INVALID; }
    uint64_t pos = 0; int nrec = 0;
    while (pos + sizeof(sslam_record_header) <= bytes) {
        sslam_record_header hd;
        memcpy(&hd, stream + pos, sizeof(hd));
        if (hd.n_kp < 0 || hd.n_ln < 0 || hd.frame < 0 || hd.frame >= nframes || (unsigned)hd.bytes != record_bytes(hd.n_kp, hd.n_ln) || pos + (uint64_t)hd.bytes > bytes) {
            set_error("sslam_unpack_records: malformed record at byte %llu", (unsigned long long)pos); return SSLAM_ERR_INVALID;
        }
        if (hd.n_kp > cap || (hd.n_ln > 0 && kl_out && hd.n_ln > lcap)) { set_error("sslam_unpack_records: frame %d exceeds the output capacity", hd.frame); return SSLAM_ERR_CAPACITY; }
        const uint8_t* p = stream + pos + sizeof(hd);
        const size_t f = (size_t)hd.frame;
        memcpy(kp_out + f * cap, p, (size_t)hd.n_kp * 28); p += (size_t)hd.n_kp * 28;
        memcpy(desc_out + f * cap * 32, p, (size_t)hd.n_kp * 32); p += (size_t)hd.n_kp * 32;
        nkp_out[f] = hd.n_kp;
        if (kl_out && ldesc_out && linefn_out && nl_out) {
            memcpy(kl_out + f * lcap, p, (size_t)hd.n_ln * 68); p += (size_t)hd.n_ln * 68;
            memcpy(ldesc_out + f * lcap * 32, p, (size_t)hd.n_ln * 32); p += (size_t)hd.n_ln * 32;
            memcpy(linefn_out + f * lcap * 3, p, (size_t)hd.n_ln * 24);
            nl_out[f] = hd.n_ln;
        }
        pos += (uint64_t)hd.bytes; ++nrec;
    }
    if (pos != bytes) { set_error("sslam_unpack_records: %llu trailing bytes", (unsigned long long)(bytes - pos)); return SSLAM_ERR_INVALID; }
    if (nrecords_out) *nrecords_out = nrec;
    return SSLAM_OK;
}

// ------------------------------------------------------------------ group handles
static void member_release(Member& m) {
    if (m.ctx) (void)hipSetDevice(m.ctx->device);
    if (m.comm && m.api) (void)m.api->CommDestroy(m.comm);
    if (m.lines) sslam_lines_destroy(m.lines);
    if (m.orb) sslam_orb_destroy(m.orb);
    DevBuf* bufs[] = {&m.dIn, &m.dKp, &m.dDesc, &m.dN, &m.dKl, &m.dLd, &m.dFn, &m.dNl, &m.dSend, &m.dTotal, &m.dStatus, &m.dRecv};
    for (DevBuf* b : bufs) b->release();
    m.hTotal.release(); m.hRecv.release();
    if (m.ctx) sslam_ctx_destroy(m.ctx);
    m = Member();
}

// TEST entry point (include/sslam_testing.h): groups created while this is on bind the in-process stand-in above instead of librccl, and may hold more members than
// GPUs are visible (dealt round-robin).  Returns the previous setting.
#ifdef SSLAM_TESTING
extern "C" int sslam_testing_use_rccl_standin(int on) { return gStandinRequested.exchange(on ? 1 : 0); }
#endif

extern "C" int sslam_group_create(int ngpu, sslam_group** out) {
    if (!out || ngpu <= 0 || ngpu > 64) { set_error("sslam_group_create: invalid arguments"); return SSLAM_ERR_INVALID; }
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) { set_error("sslam_group_create: no HIP device visible; there is no CPU fallback"); return SSLAM_ERR_NO_DEVICE; }
    const bool fake = fake_rccl_requested();      // the stand-in deals its members over the visible GPUs round-robin (several contexts per GPU)
    if (ngpu > have && !fake) { set_error("sslam_group_create: %d GPUs requested, %d visible", ngpu, have); return SSLAM_ERR_INVALID; }
    Rccl* R = rccl();
    if (!R) { set_error("sslam_group_create: librccl.so.1 could not be loaded"); return SSLAM_ERR_UNSUPPORTED; }
    sslam_group* g = new sslam_group();
    g->nranks = ngpu; g->rank = 0; g->singleProcess = true; g->api = R;
    g->mem.resize(ngpu);
    int rc = SSLAM_OK;
    for (int d = 0; d < ngpu && rc == SSLAM_OK; ++d) rc = sslam_ctx_create(d % have, &g->mem[d].ctx);
    if (rc == SSLAM_OK) {
        std::vector<int> devs(ngpu); std::vector<ncclComm_t> comms(ngpu, nullptr);
        for (int d = 0; d < ngpu; ++d) devs[d] = d % have;
        const int r = R->CommInitAll(comms.data(), ngpu, devs.data());
        if (r != 0) { set_error("ncclCommInitAll failed: %s", R->GetErrorString ? R->GetErrorString(r) : "rccl error"); rc = SSLAM_ERR_HIP; }
        else for (int d = 0; d < ngpu; ++d) { g->mem[d].comm = comms[d]; g->mem[d].api = R; }
    }
    if (rc != SSLAM_OK) { for (auto& m : g->mem) member_release(m); delete g; return rc; }
    *out = g;
    return SSLAM_OK;
}

extern "C" int sslam_group_unique_id(uint8_t id_out[SSLAM_GROUP_ID_BYTES]) {
    if (!id_out) return SSLAM_ERR_INVALID;
    if (!rccl()) { set_error("sslam_group_unique_id: librccl.so.1 could not be loaded"); return SSLAM_ERR_UNSUPPORTED; }
    NcclUid u;
    SSLAM_NCCL(rccl(), rccl()->GetUniqueId(&u));
    memcpy(id_out, u.internal, SSLAM_GROUP_ID_BYTES);
    return SSLAM_OK;
}

extern "C" int sslam_group_create_rank(int device, int rank, int nranks, const uint8_t id[SSLAM_GROUP_ID_BYTES], sslam_group** out) {
    if (!out || !id || nranks <= 0 || rank < 0 || rank >= nranks) { set_error("sslam_group_create_rank: invalid arguments"); return SSLAM_ERR_INVALID; }
    if (!rccl()) { set_error("sslam_group_create_rank: librccl.so.1 could not be loaded"); return SSLAM_ERR_UNSUPPORTED; }
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || device < 0 || device >= have) { set_error("sslam_group_create_rank: device %d not visible", device); return SSLAM_ERR_NO_DEVICE; }
    SSLAM_HIP(hipSetDevice(device));
    sslam_group* g = new sslam_group();
    g->nranks = nranks; g->rank = rank; g->device = device; g->singleProcess = false; g->api = rccl();
    g->mem.resize(1);
    NcclUid u; memcpy(u.internal, id, SSLAM_GROUP_ID_BYTES);
    const int r = g->api->CommInitRank(&g->mem[0].comm, nranks, u, rank);
    if (r != 0) { set_error("ncclCommInitRank failed: %s", g->api->GetErrorString ? g->api->GetErrorString(r) : "rccl error"); delete g; return SSLAM_ERR_HIP; }
    g->mem[0].api = g->api;
    if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess || g->dSizes.ensure(16 * (size_t)nranks + 16) != SSLAM_OK ||
        g->hSizes.ensure(16 * (size_t)nranks + 16) != SSLAM_OK) {
        set_error("sslam_group_create_rank: allocation failed"); (void)g->api->CommDestroy(g->mem[0].comm); g->mem[0].comm = nullptr; delete g; return SSLAM_ERR_HIP;
    }
    *out = g;
    return SSLAM_OK;
}

extern "C" int sslam_group_destroy(sslam_group* g) {
    if (!g) return SSLAM_OK;
    if (!g->singleProcess) (void)hipSetDevice(g->device);
    for (auto& m : g->mem) member_release(m);
    g->dSizes.release(); g->hSizes.release();
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
    return SSLAM_OK;
}

extern "C" int sslam_group_size(const sslam_group* g) { return g ? g->nranks : SSLAM_ERR_INVALID; }
extern "C" int sslam_group_rank(const sslam_group* g) { return g ? g->rank : SSLAM_ERR_INVALID; }

// ------------------------------------------------------------------ the exchange step, one process per GPU
extern "C" int sslam_group_gather_dev(sslam_group* g, const uint8_t* d_send, const uint64_t* d_send_bytes,
                                      uint8_t* d_recv, uint64_t recv_capacity, uint64_t* bytes_per_rank_out, void* stream_) {
    if (!g || g->singleProcess || !d_send || !d_send_bytes || (g->rank == 0 && (!d_recv || !bytes_per_rank_out))) {
        set_error("sslam_group_gather_dev: invalid arguments (a single-process group gathers inside sslam_frontend_batch_sharded)"); return SSLAM_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(g->mu);
    SSLAM_HIP(hipSetDevice(g->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : g->stream;
    Rccl* R = g->api;
    ncclComm_t comm = g->mem[0].comm;
    // 1. lengths: every rank learns every length AND the root's receive capacity (a pair of words per rank), so that the decision to go
    //    on is the same everywhere: a rank that returned before posting its side of the exchange would leave the others blocked in theirs
    uint64_t* dPair = g->dSizes.as<uint64_t>(); uint64_t* dAll = dPair + 2;
    uint64_t* hPair = g->hSizes.as<uint64_t>(); uint64_t* hAll = hPair + 2;
    hPair[1] = g->rank == 0 ? recv_capacity : 0;
    SSLAM_HIP(hipMemcpyAsync(dPair, d_send_bytes, 8, hipMemcpyDeviceToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(dPair + 1, hPair + 1, 8, hipMemcpyHostToDevice, st));
    SSLAM_NCCL(R, R->AllGather(dPair, dAll, 2, kNcclUint64, comm, st));
    SSLAM_HIP(hipMemcpyAsync(hAll, dAll, 16 * (size_t)g->nranks, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    std::vector<uint64_t> hSv((size_t)g->nranks);
    uint64_t* hS = hSv.data();
    uint64_t total = 0;
    for (int r = 0; r < g->nranks; ++r) {
        hS[r] = hAll[2 * r];
        if (hS[r] == ~0ull) { set_error("sslam_group_gather_dev: the record stream of rank %d overflowed its buffer", r); return SSLAM_ERR_CAPACITY; }      // (seen by every rank alike)
        total += hS[r];
    }
    if (total > hAll[1]) {                                       // hAll[1]: rank 0's capacity -- every rank takes the same exit
        set_error("sslam_group_gather_dev: %llu bytes do not fit the root's receive buffer of %llu", (unsigned long long)total, (unsigned long long)hAll[1]);
        return SSLAM_ERR_CAPACITY;
    }
    // 2. payload: one grouped send / receive per peer; the root's own stream is a device-to-device copy (or, for tests on one GPU, a
    //    self send/recv through RCCL with SSLAM_GROUP_SELF_SENDRECV=1)
    const bool selfRccl = getenv("SSLAM_GROUP_SELF_SENDRECV") != nullptr;
    SSLAM_NCCL(R, R->GroupStart());
    int rcN = 0;
    if (g->rank == 0) {
        uint64_t off = 0;
        for (int r = 0; r < g->nranks && rcN == 0; ++r) {
            if (hS[r] && (r != 0 || selfRccl)) rcN = R->Recv(d_recv + off, (size_t)hS[r], kNcclUint8, r, comm, st);
            off += hS[r];
        }
        if (rcN == 0 && selfRccl && hS[0]) rcN = R->Send(d_send, (size_t)hS[0], kNcclUint8, 0, comm, st);
    } else if (hS[g->rank]) rcN = R->Send(d_send, (size_t)hS[g->rank], kNcclUint8, 0, comm, st);
    const int rcE = R->GroupEnd();
    if (rcN != 0 || rcE != 0) { set_error("sslam_group_gather_dev: ncclSend / ncclRecv failed: %s", R->GetErrorString ? R->GetErrorString(rcN ? rcN : rcE) : "rccl error"); return SSLAM_ERR_HIP; }
    if (g->rank == 0 && !selfRccl && hS[0]) SSLAM_HIP(hipMemcpyAsync(d_recv, d_send, (size_t)hS[0], hipMemcpyDeviceToDevice, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    if (g->rank == 0) for (int r = 0; r < g->nranks; ++r) bytes_per_rank_out[r] = hS[r];
    return SSLAM_OK;
}

// ------------------------------------------------------------------ how a batch is dealt over the GPUs (host-only: no HIP, no RCCL)
// Global frame i lives on GPU i mod G; every GPU walks its frames in chunks of C = min(ceil(n / G), 512) slots, so slot j of
// chunk ck on GPU d is global frame (ck * C + j) * G + d.  The tail is uneven: the last chunk of a GPU may hold fewer frames than the same
// chunk elsewhere, or none (that GPU then takes part in the exchange with zero bytes).
extern "C" int sslam_shard_layout(int n, int ngpu, int* chunk_slots_out, int* nchunks_out) {
    if (n < 0 || ngpu <= 0) return SSLAM_ERR_INVALID;
    const int perDev = (n + ngpu - 1) / ngpu, C = std::max(1, std::min(perDev, 512));
    if (chunk_slots_out) *chunk_slots_out = C;
    if (nchunks_out) *nchunks_out = (perDev + C - 1) / C;
    return SSLAM_OK;
}
extern "C" int sslam_shard_frame(int n, int ngpu, int chunk, int gpu, int slot) {
    int C = 0, nChunks = 0;
    if (sslam_shard_layout(n, ngpu, &C, &nChunks) != SSLAM_OK || chunk < 0 || chunk >= nChunks || gpu < 0 || gpu >= ngpu || slot < 0 || slot >= C) return -1;
    const long long f = ((long long)chunk * C + slot) * ngpu + gpu;
    return f < n ? (int)f : -1;
}
extern "C" int sslam_shard_chunk_count(int n, int ngpu, int chunk, int gpu) {      // frames GPU `gpu` holds in chunk `chunk` (its slots 0 .. count-1)
    int C = 0, nChunks = 0;
    if (sslam_shard_layout(n, ngpu, &C, &nChunks) != SSLAM_OK || chunk < 0 || chunk >= nChunks || gpu < 0 || gpu >= ngpu) return 0;
    int c = 0;
    for (int j = 0; j < C; ++j) if (sslam_shard_frame(n, ngpu, chunk, gpu, j) >= 0) c = j + 1;
    return c;
}

// ------------------------------------------------------------------ host-buffer batch over all GPUs of a single-process group
extern "C" int sslam_orb_batch_status_dev(sslam_orb* orb, int cap, int32_t* d_status4, void* stream);
extern "C" int sslam_lines_batch_status_dev(sslam_lines* lines, int cap, int32_t* d_status4, void* stream);

extern "C" int sslam_frontend_batch_sharded(sslam_group* g, const sslam_frontend_params* prm,
                                            const uint8_t* images, int n, int w, int h, size_t stride, size_t image_stride,
                                            sslam_keypoint* kp_out, uint8_t* desc_out, int32_t* nkp_out, int cap,
                                            sslam_keyline* kl_out, uint8_t* ldesc_out, double* linefn_out, int32_t* nl_out, int lcap) {
    if (!g || !g->singleProcess || !prm || n < 0 || w <= 0 || h <= 0 || stride < (size_t)w || cap <= 0 || (n > 0 && (!images || !kp_out || !desc_out || !nkp_out)) ||
        (n > 1 && image_stride < stride * (size_t)(h - 1) + (size_t)w)) {
        set_error("sslam_frontend_batch_sharded: invalid arguments (needs a group from sslam_group_create)"); return SSLAM_ERR_INVALID;
    }
    const bool lines = prm->max_lines > 0;
    if (lines && (lcap <= 0 || (n > 0 && (!kl_out || !ldesc_out || !linefn_out || !nl_out)))) { set_error("sslam_frontend_batch_sharded: line outputs missing"); return SSLAM_ERR_INVALID; }
    if (n == 0) return SSLAM_OK;
    std::lock_guard<std::mutex> lk(g->mu);
    const int G = g->nranks;
    // extractors per device, rebuilt when the parameters change
    if (!g->haveParams || memcmp(&g->params, prm, sizeof(*prm)) != 0) {
        for (auto& m : g->mem) {
            (void)hipSetDevice(m.ctx->device);
            if (m.lines) { sslam_lines_destroy(m.lines); m.lines = nullptr; }
            if (m.orb) { sslam_orb_destroy(m.orb); m.orb = nullptr; }
            int rc = sslam_orb_create(m.ctx, prm->nfeatures, prm->scale_factor, prm->nlevels, prm->ini_th_fast, prm->min_th_fast, &m.orb);
            if (rc == SSLAM_OK && lines) rc = sslam_lines_create(m.ctx, prm->max_lines, &m.lines);
            // sslam_frontend_params has no field for it (its layout is part of the ABI): the members' extractors take the Gaussian variant from the environment,
            // like the drop-in classes do (SSLAM_ORB_BLUR_VARIANT=1: OpenCV 3.4.0's rounded taps, include/sslam_frontend.h)
            if (const char* e = getenv("SSLAM_ORB_BLUR_VARIANT")) {
                if (rc == SSLAM_OK) rc = sslam_orb_set_blur_variant(m.orb, atoi(e));
                if (rc == SSLAM_OK && m.lines) rc = sslam_lines_set_blur_variant(m.lines, atoi(e));
            }
            if (rc != SSLAM_OK) { g->haveParams = false; return rc; }
        }
        g->params = *prm; g->haveParams = true;
    }
    int C = 0, nChunks = 0;
    (void)sslam_shard_layout(n, G, &C, &nChunks);
    const size_t fpx = (size_t)w * h;
    const uint64_t sendCap = sslam_record_stream_capacity(C, cap, lines ? lcap : 0);
    const bool selfRccl = getenv("SSLAM_GROUP_SELF_SENDRECV") != nullptr;
    Barrier bar; bar.count = G;
    std::vector<int> status(G, SSLAM_OK), soft(G, SSLAM_OK), allocOk(G, 1);
    std::vector<std::string> errs(G), softErrs(G);
    // a frame whose records never arrive (a real HIP / RCCL failure on its GPU) keeps -1 here; every other frame is delivered, clamped to
    // the capacities, exactly as sslam_frontend_batch delivers it
    for (int i = 0; i < n; ++i) { nkp_out[i] = -1; if (lines) nl_out[i] = -1; }
    std::vector<uint64_t> sizes(G, 0);
    Rccl* R = g->api;
    auto worker = [&](int d) {
        Member& m = g->mem[d];
        int rc = SSLAM_OK;
        auto fail = [&](int code, const char* what) { if (rc == SSLAM_OK) { rc = code; errs[d] = what ? what : sslam_last_error(); } };
        // truncated rows (SSLAM_ERR_CAPACITY) and frames with too many LSD candidates (SSLAM_ERR_UNSUPPORTED) are a DEFERRED status, like
        // firstStatus in sslam_frontend_batch: the clamped records still travel, the first such status is returned at the end
        auto defer = [&](int code, const char* what) { if (soft[d] == SSLAM_OK) { soft[d] = code; softErrs[d] = what; } };
        if (hipSetDevice(m.ctx->device) != hipSuccess) fail(SSLAM_ERR_HIP, "hipSetDevice failed");
        hipStream_t st = m.ctx->stream;
        if (rc == SSLAM_OK) {
            int e = m.dIn.ensure(fpx * C) | m.dKp.ensure(sizeof(sslam_keypoint) * (size_t)C * cap) | m.dDesc.ensure(32 * (size_t)C * cap) | m.dN.ensure(4 * (size_t)C) |
                    m.dSend.ensure(sendCap + 16) | m.dTotal.ensure(16) | m.dStatus.ensure(32) | m.hTotal.ensure(64);
            if (lines) e |= m.dKl.ensure(sizeof(sslam_keyline) * (size_t)C * lcap) | m.dLd.ensure(32 * (size_t)C * lcap) | m.dFn.ensure(24 * (size_t)C * lcap) | m.dNl.ensure(4 * (size_t)C);
            if (d == 0) e |= m.dRecv.ensure(sendCap * G + 16) | m.hRecv.ensure(sendCap * G + 16);
            if (e) fail(SSLAM_ERR_HIP, nullptr);
        }
        // whether the exchange can run at all is decided together: with the root's receive buffers missing nobody may send
        allocOk[d] = rc == SSLAM_OK ? 1 : 0;
        bar.wait();
        bool everyoneReady = true;
        for (int r = 0; r < G; ++r) everyoneReady = everyoneReady && allocOk[r] != 0;
        if (!everyoneReady) { if (rc == SSLAM_OK) { rc = SSLAM_ERR_HIP; errs[d] = "another GPU of the group could not allocate its buffers"; } status[d] = rc; return; }
        for (int ck = 0; ck < nChunks; ++ck) {
            const int c = sslam_shard_chunk_count(n, G, ck, d);      // local slot j of this chunk = global frame sslam_shard_frame(n, G, ck, d, j)
            uint64_t myBytes = 0;
            if (rc == SSLAM_OK && c > 0) {
                for (int j = 0; j < c && rc == SSLAM_OK; ++j) {
                    const uint8_t* src = images + (size_t)sslam_shard_frame(n, G, ck, d, j) * image_stride;
                    if (hipMemcpy2DAsync(m.dIn.as<uint8_t>() + (size_t)j * fpx, w, src, stride, w, h, hipMemcpyHostToDevice, st) != hipSuccess) fail(SSLAM_ERR_HIP, "H2D failed");
                }
                if (rc == SSLAM_OK && (rc = sslam_orb_extract_batch_dev(m.orb, m.dIn.as<uint8_t>(), w, h, (size_t)w, fpx, c, m.dKp.as<sslam_keypoint>(), m.dDesc.as<uint8_t>(),
                                                                         m.dN.as<int32_t>(), cap, st))) errs[d] = sslam_last_error();
                if (rc == SSLAM_OK && lines && (rc = sslam_lines_extract_batch_dev(m.lines, m.dIn.as<uint8_t>(), w, h, (size_t)w, fpx, c, m.dKl.as<sslam_keyline>(), m.dLd.as<uint8_t>(),
                                                                                    m.dFn.as<double>(), m.dNl.as<int32_t>(), lcap, st))) errs[d] = sslam_last_error();
                if (rc == SSLAM_OK && (rc = sslam_orb_batch_status_dev(m.orb, cap, m.dStatus.as<int32_t>(), st))) errs[d] = sslam_last_error();
                if (rc == SSLAM_OK && lines && (rc = sslam_lines_batch_status_dev(m.lines, lcap, m.dStatus.as<int32_t>() + 4, st))) errs[d] = sslam_last_error();
                if (rc == SSLAM_OK && (rc = sslam_pack_records_dev(m.ctx, c, sslam_shard_frame(n, G, ck, d, 0), G, m.dKp.as<sslam_keypoint>(), m.dDesc.as<uint8_t>(), m.dN.as<int32_t>(), cap,
                                                                   lines ? m.dKl.as<sslam_keyline>() : nullptr, m.dLd.as<uint8_t>(), m.dFn.as<double>(), m.dNl.as<int32_t>(), lcap,
                                                                   m.dSend.as<uint8_t>(), sendCap, m.dTotal.as<uint64_t>(), st))) errs[d] = sslam_last_error();
                if (rc == SSLAM_OK) {
                    uint8_t* hp = m.hTotal.as<uint8_t>();
                    if (hipMemcpyAsync(hp, m.dTotal.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(hp + 16, m.dStatus.p, 32, hipMemcpyDeviceToHost, st) != hipSuccess ||
                        hipStreamSynchronize(st) != hipSuccess) fail(SSLAM_ERR_HIP, "kernels / D2H failed");
                    else {
                        const int* S = (const int*)(hp + 16);
                        if (lines && S[6]) defer(SSLAM_ERR_UNSUPPORTED, "a frame produced more than 8192 LSD candidate rectangles");
                        else if (S[0]) defer(SSLAM_ERR_CAPACITY, "a frame holds more keypoints than cap (rows truncated)");
                        else if (lines && S[4]) defer(SSLAM_ERR_CAPACITY, "a frame holds more lines than lcap (rows truncated)");
                        myBytes = *(const uint64_t*)hp;
                        if (myBytes == ~0ull) { fail(SSLAM_ERR_CAPACITY, "record stream overflow"); myBytes = 0; }
                    }
                }
            }
            if (rc != SSLAM_OK) myBytes = 0;      // a member with a real failure still takes part in the exchange (with nothing), so that nobody hangs
            sizes[d] = myBytes;
            bar.wait();
            // the exchange step: grouped ncclSend / ncclRecv to device 0
            uint64_t total = 0;
            for (int r = 0; r < G; ++r) total += sizes[r];
            int e1 = R->GroupStart(), e2 = 0;
            if (d == 0) {
                uint64_t off = 0;
                for (int r = 0; r < G; ++r) {
                    if (sizes[r] && (r != 0 || selfRccl) && !e2) e2 = R->Recv(m.dRecv.as<uint8_t>() + off, (size_t)sizes[r], kNcclUint8, r, m.comm, st);
                    off += sizes[r];
                }
                if (selfRccl && sizes[0] && !e2) e2 = R->Send(m.dSend.p, (size_t)sizes[0], kNcclUint8, 0, m.comm, st);
            } else if (sizes[d]) e2 = R->Send(m.dSend.p, (size_t)sizes[d], kNcclUint8, 0, m.comm, st);
            const int e3 = R->GroupEnd();
            if (e1 || e2 || e3) fail(SSLAM_ERR_HIP, "ncclSend / ncclRecv failed");
            if (d == 0 && !selfRccl && sizes[0] && hipMemcpyAsync(m.dRecv.p, m.dSend.p, (size_t)sizes[0], hipMemcpyDeviceToDevice, st) != hipSuccess) fail(SSLAM_ERR_HIP, "D2D failed");
            if (d == 0 && total) {
                if (hipMemcpyAsync(m.hRecv.p, m.dRecv.p, (size_t)total, hipMemcpyDeviceToHost, st) != hipSuccess) fail(SSLAM_ERR_HIP, "D2H failed");
            }
            if (hipStreamSynchronize(st) != hipSuccess) fail(SSLAM_ERR_HIP, "exchange failed");
            if (d == 0 && total) {                // whatever arrived is delivered, also after a failure of the root's own extraction
                int nrec = 0;
                const int u = sslam_unpack_records(m.hRecv.as<uint8_t>(), total, n, kp_out, desc_out, nkp_out, cap, lines ? kl_out : nullptr, ldesc_out, linefn_out, nl_out, lcap, &nrec);
                if (u != SSLAM_OK) fail(u, nullptr);
            }
            bar.wait();      // buffers are reused by the next chunk only after the root has taken everything
        }
        status[d] = rc;
    };
    std::vector<std::thread> th;
    for (int d = 1; d < G; ++d) th.emplace_back(worker, d);
    worker(0);
    for (auto& t : th) t.join();
    for (int d = 0; d < G; ++d)
        if (status[d] != SSLAM_OK) { set_error("sslam_frontend_batch_sharded: GPU %d: %s", d, errs[d].c_str()); return status[d]; }
    for (int d = 0; d < G; ++d)
        if (soft[d] != SSLAM_OK) { set_error("sslam_frontend_batch_sharded: GPU %d: %s", d, softErrs[d].c_str()); return soft[d]; }
    return SSLAM_OK;
}
