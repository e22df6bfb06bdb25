// Hamming matchers for MI355X (gfx950): brute-force knn-2 (BFMatcher semantics),
// dense distance matrix, ORBmatcher::SearchForInitialization and the LSDmatcher
// knn2+MAD gates — device side of reference src/ORBmatcher.cc:408-523,1604-1666,
// src/LSDmatcher.cpp:143-183,257-284,364-415, src/Frame.cc:133-148,190-215,368-472.
//
// The path is integer/bitwise: v_bcnt popcounts on 8x u32 XORs, wave-wide min
// reductions with the tie-break order carried in the key; no MFMA.
#include "common.h"
#include <algorithm>

using namespace sslam;

namespace {

#include "match_knn.h"
#include "match_ordered.h"
#include "match_independent.h"

}  // namespace

// =============================================================== host side
static hipStream_t pick(sslam_ctx* c, void* s) { return s ? (hipStream_t)s : c->stream; }

extern "C" int sslam_hamming_knn2_dev(sslam_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_idx, int32_t* d_dist, void* stream) {
    if (!ctx || nq < 0 || nt < 0 || (nq > 0 && (!d_q || !d_idx || !d_dist))) { set_error("sslam_hamming_knn2_dev: invalid arguments"); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (nq == 0) return SSLAM_OK;
    SSLAM_HIP(hipSetDevice(ctx->device));
    { sslam::ProfScope _ps(ctx, "k_knn2", pick(ctx, stream)); hipLaunchKernelGGL(k_knn2, dim3((nq + 3) / 4), dim3(256), 0, pick(ctx, stream), d_q, nq, d_t, nt, d_idx, d_dist); }
    SSLAM_HIP(hipGetLastError());
    return SSLAM_OK;
}

extern "C" int sslam_hamming_knn2_batch_dev(sslam_ctx* ctx, const uint8_t* d_q, const int32_t* d_nq, const uint8_t* d_t, const int32_t* d_nt,
                                            int cap, int nframes, int32_t* d_idx, int32_t* d_dist, void* stream) {
    if (!ctx || !d_q || !d_nq || !d_t || !d_nt || !d_idx || !d_dist || cap <= 0 || nframes <= 0) { set_error("sslam_hamming_knn2_batch_dev: invalid arguments"); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    const char* formEnv = getenv("SSLAM_KNN2_BATCH");      // experiment / test knob, read on every call: popc = the xor + popcount form (the only one beyond 4096 rows), mfma1 = 32 queries per wave
    const int form = !formEnv ? 2 : !strcmp(formEnv, "popc") ? 0 : !strcmp(formEnv, "mfma1") ? 1 : 2;
    const int tilesCap = (cap + 31) / 32;
    if (tilesCap <= KNN_MFMA_MAX_TILES && form) {       // matrix-core form (match_knn.h): train rows expanded to +-64 bytes in operand order, then 32 / 64 queries per wave
        int rc;
        // ONE expand buffer per context (256 B per train row: 3 GB at 12 288 frames of 1 000 rows, never shrunk; sslam_frontend_batch counts it in its chunk estimate).
        // k_knn2_expand writes it and k_knn2_mfma reads it on the caller's stream: a call on another stream first waits for the event behind the previous reader.
        if (ctx->knnDone && ctx->knnLastStream != (void*)st) SSLAM_HIP(hipStreamWaitEvent(st, ctx->knnDone, 0));
        if (!ctx->knnDone) SSLAM_HIP(hipEventCreateWithFlags(&ctx->knnDone, hipEventDisableTiming));
        if ((rc = ctx->knnExpand.ensure((size_t)nframes * tilesCap * 8 * 1024))) return rc;
        { sslam::ProfScope _ps(ctx, "k_knn2_expand", st); hipLaunchKernelGGL(k_knn2_expand, dim3(tilesCap, nframes), dim3(64), 0, st, d_t, d_nt, cap, tilesCap, ctx->knnExpand.as<uint8_t>()); }
        const int qblocks = form == 2 ? (cap + 63) / 64 : tilesCap;
        const dim3 grid(8u * (unsigned)((nframes + 7) / 8) * (unsigned)qblocks);
        sslam::ProfScope _ps(ctx, "k_knn2_batch", st);
        if (form == 2) hipLaunchKernelGGL(k_knn2_mfma<2>, grid, dim3(64), 0, st, d_q, d_nq, ctx->knnExpand.as<uint8_t>(), d_nt, cap, tilesCap, qblocks, nframes, d_idx, d_dist);
        else hipLaunchKernelGGL(k_knn2_mfma<1>, grid, dim3(64), 0, st, d_q, d_nq, ctx->knnExpand.as<uint8_t>(), d_nt, cap, tilesCap, qblocks, nframes, d_idx, d_dist);
        SSLAM_HIP(hipEventRecord(ctx->knnDone, st)); ctx->knnLastStream = (void*)st;
    } else
    { sslam::ProfScope _ps(ctx, "k_knn2_batch", st); hipLaunchKernelGGL(k_knn2_batch, dim3((cap + 15) / 16, nframes), dim3(256), 0, st, d_q, d_nq, d_t, d_nt, cap, d_idx, d_dist); }
    SSLAM_HIP(hipGetLastError());
    return SSLAM_OK;
}

extern "C" int sslam_hamming_knn2(sslam_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist) {
    if (!ctx || nq < 0 || nt < 0 || (nq > 0 && (!q || !idx || !dist)) || (nt > 0 && !t)) { set_error("sslam_hamming_knn2: invalid arguments"); return SSLAM_ERR_INVALID; }
    if (nq == 0) return SSLAM_OK;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = ctx->scratch[0].ensure((size_t)nq * 32))) return rc;
    if ((rc = ctx->scratch[1].ensure((size_t)std::max(nt, 1) * 32))) return rc;
    if ((rc = ctx->scratch[2].ensure((size_t)nq * 16))) return rc;
    hipStream_t st = ctx->stream;
    SSLAM_HIP(hipMemcpyAsync(ctx->scratch[0].p, q, (size_t)nq * 32, hipMemcpyHostToDevice, st));
    if (nt) SSLAM_HIP(hipMemcpyAsync(ctx->scratch[1].p, t, (size_t)nt * 32, hipMemcpyHostToDevice, st));
    int* di = ctx->scratch[2].as<int>();
    if ((rc = sslam_hamming_knn2_dev(ctx, ctx->scratch[0].as<uint8_t>(), nq, ctx->scratch[1].as<uint8_t>(), nt, di, di + (size_t)nq * 2, st))) return rc;
    SSLAM_HIP(hipMemcpyAsync(idx, di, (size_t)nq * 8, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(dist, di + (size_t)nq * 2, (size_t)nq * 8, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    return SSLAM_OK;
}

extern "C" int sslam_hamming_matrix(sslam_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* D) {
    if (!ctx || nq < 0 || nt < 0 || ((nq > 0 && nt > 0) && (!q || !t || !D))) { set_error("sslam_hamming_matrix: invalid arguments"); return SSLAM_ERR_INVALID; }
    if (nq == 0 || nt == 0) return SSLAM_OK;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = ctx->scratch[0].ensure((size_t)nq * 32))) return rc;
    if ((rc = ctx->scratch[1].ensure((size_t)nt * 32))) return rc;
    if ((rc = ctx->scratch[2].ensure((size_t)nq * nt * 2))) return rc;
    hipStream_t st = ctx->stream;
    SSLAM_HIP(hipMemcpyAsync(ctx->scratch[0].p, q, (size_t)nq * 32, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(ctx->scratch[1].p, t, (size_t)nt * 32, hipMemcpyHostToDevice, st));
    { sslam::ProfScope _ps(ctx, "k_hamming_matrix", st); hipLaunchKernelGGL(k_hamming_matrix, dim3((nt + 255) / 256, nq), dim3(256), 0, st, ctx->scratch[0].as<uint8_t>(), nq,
                       ctx->scratch[1].as<uint8_t>(), nt, ctx->scratch[2].as<unsigned short>()); }
    SSLAM_HIP(hipGetLastError());
    SSLAM_HIP(hipMemcpyAsync(D, ctx->scratch[2].p, (size_t)nq * nt * 2, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    return SSLAM_OK;
}

extern "C" int sslam_orb_search_for_initialization_batch_dev(sslam_ctx* ctx,
        const sslam_keypoint* d_kp1, const uint8_t* d_desc1, const int32_t* d_n1,
        const sslam_keypoint* d_kp2, const uint8_t* d_desc2, const int32_t* d_n2,
        int cap, int npairs, float* d_prev, int32_t* d_m12, int32_t* d_nm,
        int window, float nnratio, int checkOri, const float bounds[4], void* stream) {
    if (!ctx || !d_kp1 || !d_desc1 || !d_kp2 || !d_desc2 || !d_n1 || !d_n2 || !d_prev || !d_m12 || !d_nm || cap <= 0 || cap >= (1 << 19) || npairs <= 0 || !bounds) {
        set_error("sslam_orb_search_for_initialization_batch_dev: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);      // scratch buffers and profile records are shared state
    SSLAM_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = ctx->scratch[3].ensure(sizeof(int) * 5 * (size_t)cap * npairs))) return rc;
    SfiArgs A;
    A.kp1 = d_kp1; A.d1 = d_desc1; A.n1 = d_n1; A.kp2 = d_kp2; A.d2 = d_desc2; A.n2 = d_n2;
    A.cap = cap; A.n1s = 0; A.n2s = 0; A.prevMatched = d_prev; A.m12 = d_m12; A.nmatches = d_nm;
    A.scratch = ctx->scratch[3].as<int>(); A.window = window; A.nnratio = nnratio; A.checkOri = checkOri;
    A.minX = bounds[0]; A.maxX = bounds[1]; A.minY = bounds[2]; A.maxY = bounds[3]; A.ccap = 0;
    // a handful of pairs (the single call of Tracking::MonocularInitialization): the LDS-resident kernel, as long as a pair fits the CU's LDS
    const size_t ldsNeed = 64 + (size_t)cap * 15 * 4 + (size_t)cap * 2 * 4;      // per candidate 15 words (14 + the stamp of the speculative form), per F1 keypoint 2
    const char* sfiForm = getenv("SSLAM_SFI_FORM");      // test / experiment knob: "global", "lds" (one wave), default: sixteen speculative waves
    if (npairs <= 8 && ldsNeed <= 150 * 1024 && !getenv("SSLAM_SFI_GLOBAL") && !(sfiForm && sfiForm[0] == 'g')) {
        const bool oneWave = sfiForm && sfiForm[0] == 'l';
        if (ldsNeed > 48 * 1024) SSLAM_HIP(hipFuncSetAttribute(oneWave ? (const void*)k_search_init_lds : (const void*)k_search_init_spec, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsNeed));
        sslam::ProfScope _ps(ctx, "k_search_init", pick(ctx, stream));
        if (oneWave) hipLaunchKernelGGL(k_search_init_lds, dim3(npairs), dim3(64), ldsNeed, pick(ctx, stream), A);
        else hipLaunchKernelGGL(k_search_init_spec, dim3(npairs), dim3(SFI_WAVES * 64), ldsNeed, pick(ctx, stream), A);
    } else {
        // the batch: one wave per pair with the pair's level-0 features in LDS (capacity 3/8 of the rows, at least 256: the level-0 quota of an
        // 8-level pyramid is 21.7 % of nfeatures; a pair beyond it takes the global-memory body inside the same launch).  SSLAM_SFI_BATCH=global: the round 1-3 kernel.
        const int ccap = std::min(cap, std::max(256, cap * 3 / 8));
        const size_t ldsB = 64 + (size_t)ccap * 16 * 4;
        const char* bf = getenv("SSLAM_SFI_BATCH");
        sslam::ProfScope _ps(ctx, "k_search_init", pick(ctx, stream));
        if (ldsB <= 64 * 1024 && !(bf && bf[0] == 'g')) {
            A.ccap = ccap;
            if (ldsB > 48 * 1024) SSLAM_HIP(hipFuncSetAttribute((const void*)k_search_init_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB));
            hipLaunchKernelGGL(k_search_init_lds, dim3(npairs), dim3(64), ldsB, pick(ctx, stream), A);
        } else hipLaunchKernelGGL(k_search_init, dim3(npairs), dim3(64), 0, pick(ctx, stream), A);
    }
    SSLAM_HIP(hipGetLastError());
    return SSLAM_OK;
}

extern "C" int sslam_orb_search_for_initialization(sslam_ctx* ctx,
        const sslam_keypoint* kp1, const uint8_t* desc1, int n1, const sslam_keypoint* kp2, const uint8_t* desc2, int n2,
        float* prev_matched, int32_t* matches12, int window, float nnratio, int checkOri, const float bounds[4], int* nmatches_out) {
    if (!ctx || n1 < 0 || n2 < 0 || !nmatches_out || !bounds || (n1 > 0 && (!kp1 || !desc1 || !prev_matched || !matches12)) || (n2 > 0 && (!kp2 || !desc2))) {
        set_error("sslam_orb_search_for_initialization: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    *nmatches_out = 0;
    if (n1 == 0) return SSLAM_OK;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    const int cap = std::max(std::max(n1, n2), 1);
    if (cap >= (1 << 19)) { set_error("too many keypoints"); return SSLAM_ERR_UNSUPPORTED; }
    hipStream_t st = ctx->stream;
    const size_t kb = sizeof(sslam_keypoint) * (size_t)cap, db = 32 * (size_t)cap;
    int rc;
    // device layout: kp1 | kp2 | d1 | d2 | prev | n1,n2,nm (16 B) | m12 -- everything that goes in is one contiguous range, so is everything
    // that comes back (prev .. m12): ONE staged H2D and ONE D2H through pinned memory instead of six + three pageable copies (a pageable
    // hipMemcpyAsync costs 10-20 us of host time each, more than the kernel of a single call)
    const size_t oPrev = 2 * kb + 2 * db, oN = oPrev + 8 * (size_t)cap, oM = oN + 16, total = oM + 4 * (size_t)cap + 64;
    if ((rc = ctx->scratch[4].ensure(total))) return rc;
    if ((rc = ctx->pinned[2].ensure(total))) return rc;
    uint8_t* base = ctx->scratch[4].as<uint8_t>();
    uint8_t* H = ctx->pinned[2].as<uint8_t>();
    sslam_keypoint* dk1 = (sslam_keypoint*)base; sslam_keypoint* dk2 = (sslam_keypoint*)(base + kb);
    uint8_t* dd1 = base + 2 * kb; uint8_t* dd2 = dd1 + db;
    float* dpm = (float*)(base + oPrev); int* dn = (int*)(base + oN); int* dm12 = (int*)(base + oM);
    memcpy(H, kp1, sizeof(sslam_keypoint) * (size_t)n1);
    if (n2) memcpy(H + kb, kp2, sizeof(sslam_keypoint) * (size_t)n2);
    memcpy(H + 2 * kb, desc1, 32 * (size_t)n1);
    if (n2) memcpy(H + 2 * kb + db, desc2, 32 * (size_t)n2);
    memcpy(H + oPrev, prev_matched, 8 * (size_t)n1);
    int hn[4] = {n1, n2, 0, 0};
    memcpy(H + oN, hn, sizeof(hn));
    SSLAM_HIP(hipMemcpyAsync(base, H, oM, hipMemcpyHostToDevice, st));
    if ((rc = sslam_orb_search_for_initialization_batch_dev(ctx, dk1, dd1, dn, dk2, dd2, dn + 1, cap, 1, dpm, dm12, dn + 2, window, nnratio, checkOri, bounds, st))) return rc;
    SSLAM_HIP(hipMemcpyAsync(H + oPrev, base + oPrev, oM + 4 * (size_t)n1 - oPrev, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    memcpy(prev_matched, H + oPrev, 8 * (size_t)n1);
    memcpy(matches12, H + oM, 4 * (size_t)n1);
    memcpy(hn, H + oN, sizeof(hn));
    *nmatches_out = hn[2];
    return SSLAM_OK;
}

extern "C" int sslam_line_match_batch_dev(sslam_ctx* ctx, const uint8_t* d_l1, const int32_t* d_n1, const uint8_t* d_l2, const int32_t* d_n2,
                                          int cap, int npf, double gate_scale, int ratio_mode, int32_t* d_pairs, int32_t* d_npairs, void* stream) {
    if (!ctx || !d_l1 || !d_l2 || !d_n1 || !d_n2 || !d_pairs || !d_npairs || cap <= 0 || npf <= 0) { set_error("sslam_line_match_batch_dev: invalid arguments"); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    { sslam::ProfScope _ps(ctx, "k_line_match", pick(ctx, stream)); hipLaunchKernelGGL(k_line_match, dim3(npf), dim3(256), 0, pick(ctx, stream), d_l1, d_n1, 0, d_l2, d_n2, 0, cap, gate_scale, ratio_mode, d_pairs, d_npairs, (double*)nullptr); }
    SSLAM_HIP(hipGetLastError());
    return SSLAM_OK;
}

extern "C" int sslam_line_match(sslam_ctx* ctx, const uint8_t* l1, int n1, const uint8_t* l2, int n2, double gate_scale, int ratio_mode,
                                int32_t* pairs_out, int cap, int* npairs_out, double* nn_mad_out, double* nn12_mad_out) {
    if (!ctx || n1 < 0 || n2 < 0 || !npairs_out || (n1 > 0 && (!l1 || !pairs_out)) || (n2 > 0 && !l2)) { set_error("sslam_line_match: invalid arguments"); return SSLAM_ERR_INVALID; }
    *npairs_out = 0;
    if (nn_mad_out) *nn_mad_out = 0;
    if (nn12_mad_out) *nn12_mad_out = 0;
    if (n1 == 0 || n2 < 2) return SSLAM_OK;       // degenerate: defined as no matches
    if (n1 > LM_MAX) { set_error("sslam_line_match: n1=%d exceeds %d", n1, LM_MAX); return SSLAM_ERR_UNSUPPORTED; }
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int c = std::max(n1, n2);
    int rc;
    size_t total = 64 * (size_t)c + 8 * (size_t)c + 64;
    if ((rc = ctx->scratch[5].ensure(total))) return rc;
    uint8_t* base = ctx->scratch[5].as<uint8_t>();
    uint8_t* d1 = base; uint8_t* d2 = base + 32 * (size_t)c;
    int* dp = (int*)(d2 + 32 * (size_t)c); int* dn = dp + 2 * (size_t)c; double* dm = (double*)(dn + 4);
    SSLAM_HIP(hipMemcpyAsync(d1, l1, 32 * (size_t)n1, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(d2, l2, 32 * (size_t)n2, hipMemcpyHostToDevice, st));
    { sslam::ProfScope _ps(ctx, "k_line_match", st); hipLaunchKernelGGL(k_line_match, dim3(1), dim3(256), 0, st, d1, (const int*)nullptr, n1, d2, (const int*)nullptr, n2, c, gate_scale, ratio_mode, dp, dn, dm); }
    SSLAM_HIP(hipGetLastError());
    int np = 0; double mads[2] = {0, 0};
    std::vector<int> hp(2 * (size_t)c);
    SSLAM_HIP(hipMemcpyAsync(&np, dn, sizeof(int), hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(mads, dm, sizeof(mads), hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(hp.data(), dp, 8 * (size_t)c, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    *npairs_out = np;
    if (nn_mad_out) *nn_mad_out = mads[0];
    if (nn12_mad_out) *nn12_mad_out = mads[1];
    if (np > cap) { set_error("sslam_line_match: %d pairs exceed capacity %d", np, cap); return SSLAM_ERR_CAPACITY; }
    memcpy(pairs_out, hp.data(), 8 * (size_t)np);
    return SSLAM_OK;
}

// shared body of the projection matchers: features already on the device (d_feats / d_desc / d_uright), per-call inputs staged here
static int search_proj_core(sslam_ctx* ctx, int kind, int mode, const void* d_feats, const uint8_t* d_desc, int n, const float bounds[4],
                            const float* d_uright, const uint8_t* occupied, const sslam_proj_query* queries, const uint8_t* qdesc, int nq,
                            float nnratio, int th_dist, int check_orientation, int32_t* assigned_out, int* nmatches_out) {
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t oO = 0, oQ = oO + al((size_t)n), oQD = oQ + al(sizeof(sslam_proj_query) * (size_t)nq), oA = oQD + al(32 * (size_t)nq), oN = oA + al(4 * (size_t)n),
           oS = oN + 256, oT = oS + al(4 * (2 * (size_t)n + 2 * (size_t)nq)), oC = oT + al(8 * (size_t)PROJ_K * (size_t)nq), total = oC + al(4 * (size_t)nq);
    int rc;
    if ((rc = ctx->scratch[6].ensure(total))) return rc;
    uint8_t* B = ctx->scratch[6].as<uint8_t>();
    // occupancy, queries and query descriptors sit back to back in B: one staged H2D through pinned memory (each pageable copy costs more
    // host time than it moves data); the results (assigned, count) come back the same way
    if ((rc = ctx->pinned[3].ensure(oN + 256))) return rc;
    uint8_t* H = ctx->pinned[3].as<uint8_t>();
    if (occupied) memcpy(H + oO, occupied, (size_t)n);
    memcpy(H + oQ, queries, sizeof(sslam_proj_query) * (size_t)nq);
    memcpy(H + oQD, qdesc, 32 * (size_t)nq);
    SSLAM_HIP(hipMemcpyAsync(B + oO, H + oO, oA, hipMemcpyHostToDevice, st));
    ProjArgs A;
    A.kind = kind; A.mode = mode; A.feats = (const uint8_t*)d_feats; A.desc = d_desc; A.n = n;
    A.minX = bounds[0]; A.maxX = bounds[1]; A.minY = bounds[2]; A.maxY = bounds[3];
    A.uright = d_uright; A.occIn = occupied ? B + oO : nullptr;
    A.q = (const sslam_proj_query*)(B + oQ); A.qdesc = B + oQD; A.nq = nq; A.nnratio = nnratio; A.thDist = th_dist; A.checkOri = check_orientation;
    A.assigned = (int*)(B + oA); A.nmatches = (int*)(B + oN); A.scratch = (int*)(B + oS);
    A.stats = getenv("SSLAM_PROJ_STATS") ? (long long*)(B + oN + 64) : nullptr;      // development aid: seven counters behind the match count
    const char* form = getenv("SSLAM_PROJ_FORM");      // test / experiment knob: "lds" (sixteen speculative waves on one CU), "wave" (one wave), default: two kernels
    if (nq > 0 && n <= 8192 && !form) {      // one wave per query over the whole chip, then an ordered commit with parallel prefixes (match_ordered.h)
        ProjTopArgs T; T.A = A; T.top = (unsigned long long*)(B + oT); T.cnt = (int*)(B + oC);
        { sslam::ProfScope _ps(ctx, "k_proj_topk", st); hipLaunchKernelGGL(k_proj_topk, dim3((nq + 3) / 4), dim3(256), 0, st, T); }
        const int featsInLds = n <= PROJ_MAXN ? 1 : 0;      // 64 bytes per feature: the re-scans of the commit then never leave the CU
        const size_t lds = (featsInLds ? 64 : 8) * (size_t)n + 64;
        if (lds > 48 * 1024) SSLAM_HIP(hipFuncSetAttribute((const void*)k_proj_commit, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        { sslam::ProfScope _ps(ctx, "k_proj_commit", st); hipLaunchKernelGGL(k_proj_commit, dim3(1), dim3(64), lds, st, T, featsInLds); }
    } else if (n <= PROJ_MAXN && !(form && form[0] == 'w')) {       // the frame fits in LDS: sixteen speculative queries per round
        int n2 = 64; while (n2 < n) n2 <<= 1;
        const size_t lds = (size_t)n * (32 + 7 * 4) + (size_t)n2 * 4 + (GRID_COLS + 2) * 4 + 64;
        if (lds > 48 * 1024) SSLAM_HIP(hipFuncSetAttribute((const void*)k_search_proj_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        sslam::ProfScope _ps(ctx, "k_search_proj_lds", st);
        hipLaunchKernelGGL(k_search_proj_lds, dim3(1), dim3(PROJ_WAVES * 64), lds, st, A);
    } else { sslam::ProfScope _ps(ctx, "k_search_proj", st); hipLaunchKernelGGL(k_search_proj, dim3(1), dim3(64), 0, st, A); }
    SSLAM_HIP(hipGetLastError());
    SSLAM_HIP(hipMemcpyAsync(H + oA, B + oA, oN + 4 - oA, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    memcpy(assigned_out, H + oA, 4 * (size_t)n);
    memcpy(nmatches_out, H + oN, sizeof(int));
    if (A.stats) {
        long long hs[4];
        SSLAM_HIP(hipMemcpy(hs, B + oN + 64, sizeof(hs), hipMemcpyDeviceToHost));
        fprintf(stderr, "proj stats: commit steps %lld, re-scans %lld (%lld cycles), commit kernel %lld cycles\n", hs[0], hs[1], hs[2], hs[3]);
    }
    return SSLAM_OK;
}

extern "C" int sslam_search_by_projection(sslam_ctx* ctx, int kind, int mode, const void* feats, const uint8_t* desc, int n, const float bounds[4],
                                          const float* uright, const uint8_t* occupied, const sslam_proj_query* queries, const uint8_t* qdesc, int nq,
                                          float nnratio, int th_dist, int check_orientation, int32_t* assigned_out, int* nmatches_out) {
    if (!ctx || (kind != 0 && kind != 1) || (mode != 0 && mode != 1) || (kind == 1 && mode == 1 && check_orientation) || n < 0 || nq < 0 || !nmatches_out || !bounds ||
        (n > 0 && (!feats || !desc || !assigned_out)) || (nq > 0 && (!queries || !qdesc)) || n >= (1 << 19)) {
        set_error("sslam_search_by_projection: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    *nmatches_out = 0;
    for (int i = 0; i < n; ++i) assigned_out[i] = -1;
    if (n == 0 || nq == 0) return SSLAM_OK;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t fsz = kind == 0 ? sizeof(sslam_keypoint) : sizeof(sslam_keyline);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t oF = 0, oD = oF + al(fsz * n), oU = oD + al(32 * (size_t)n), total = oU + al(4 * (size_t)n);
    int rc;
    if ((rc = ctx->scratch[7].ensure(total))) return rc;
    uint8_t* B = ctx->scratch[7].as<uint8_t>();
    SSLAM_HIP(hipMemcpyAsync(B + oF, feats, fsz * n, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oD, desc, 32 * (size_t)n, hipMemcpyHostToDevice, st));
    if (uright) SSLAM_HIP(hipMemcpyAsync(B + oU, uright, 4 * (size_t)n, hipMemcpyHostToDevice, st));
    return search_proj_core(ctx, kind, mode, B + oF, B + oD, n, bounds, uright ? (const float*)(B + oU) : nullptr, occupied, queries, qdesc, nq,
                            nnratio, th_dist, check_orientation, assigned_out, nmatches_out);
}

// ---- device-resident frames
extern "C" void sslam_frame_destroy(sslam_frame* f);
extern "C" void sslam_vocab_destroy(sslam_vocab* v);
namespace {
// a handle under construction: destroyed on every early return (the SSLAM_HIP error paths included), handed over with release()
template <class T, void (*Destroy)(T*)> struct HandleGuard {
    T* p; explicit HandleGuard(T* q) : p(q) {}
    ~HandleGuard() { if (p) Destroy(p); }
    T* release() { T* q = p; p = nullptr; return q; }
};
}  // namespace

extern "C" int sslam_frame_upload(sslam_ctx* ctx, int kind, const void* feats, const uint8_t* desc, int n, const float* uright, const float bounds[4],
                                  sslam_frame** out) {
    if (!ctx || !out || (kind != 0 && kind != 1) || n < 0 || n >= (1 << 19) || !bounds || (n > 0 && (!feats || !desc))) {
        set_error("sslam_frame_upload: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    sslam_frame* f = new sslam_frame();
    HandleGuard<sslam_frame, sslam_frame_destroy> guard(f);
    f->ctx = ctx; f->kind = kind; f->n = n; f->hasUright = uright != nullptr;
    for (int i = 0; i < 4; ++i) f->bounds[i] = bounds[i];
    const size_t fsz = kind == 0 ? sizeof(sslam_keypoint) : sizeof(sslam_keyline);
    int rc = SSLAM_OK;
    if ((rc = f->feats.ensure(std::max<size_t>(fsz * n, 256))) || (rc = f->desc.ensure(std::max<size_t>(32 * (size_t)n, 256))) ||
        (uright && (rc = f->uright.ensure(std::max<size_t>(4 * (size_t)n, 256))))) return rc;
    if (n > 0) {
        SSLAM_HIP(hipMemcpyAsync(f->feats.p, feats, fsz * n, hipMemcpyHostToDevice, ctx->stream));
        SSLAM_HIP(hipMemcpyAsync(f->desc.p, desc, 32 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        if (uright) SSLAM_HIP(hipMemcpyAsync(f->uright.p, uright, 4 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        SSLAM_HIP(hipStreamSynchronize(ctx->stream));
    }
    *out = guard.release();
    return SSLAM_OK;
}

// adopt device buffers the extractors already hold (sslam_frame_from_orb / sslam_frame_from_lines): device-to-device snapshot
int sslam_frame_from_device(sslam_ctx* ctx, int kind, const void* d_feats, const uint8_t* d_desc, int n, const float bounds[4], sslam_frame** out) {
    SSLAM_HIP(hipSetDevice(ctx->device));
    sslam_frame* f = new sslam_frame();
    HandleGuard<sslam_frame, sslam_frame_destroy> guard(f);
    f->ctx = ctx; f->kind = kind; f->n = n;
    for (int i = 0; i < 4; ++i) f->bounds[i] = bounds[i];
    const size_t fsz = kind == 0 ? sizeof(sslam_keypoint) : sizeof(sslam_keyline);
    int rc;
    if ((rc = f->feats.ensure(std::max<size_t>(fsz * n, 256))) || (rc = f->desc.ensure(std::max<size_t>(32 * (size_t)n, 256)))) return rc;
    if (n > 0) {
        SSLAM_HIP(hipMemcpyAsync(f->feats.p, d_feats, fsz * n, hipMemcpyDeviceToDevice, ctx->stream));
        SSLAM_HIP(hipMemcpyAsync(f->desc.p, d_desc, 32 * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
        SSLAM_HIP(hipStreamSynchronize(ctx->stream));
    }
    *out = guard.release();
    return SSLAM_OK;
}

extern "C" void sslam_frame_destroy(sslam_frame* f) {
    if (!f) return;
    if (f->ctx) (void)hipSetDevice(f->ctx->device);
    f->feats.release(); f->desc.release(); f->uright.release();
    delete f;
}

extern "C" int sslam_frame_count(const sslam_frame* f) { return f ? f->n : 0; }

extern "C" int sslam_search_by_projection_frame(sslam_ctx* ctx, const sslam_frame* frame, int mode, const uint8_t* occupied,
                                                const sslam_proj_query* queries, const uint8_t* qdesc, int nq,
                                                float nnratio, int th_dist, int check_orientation, int32_t* assigned_out, int* nmatches_out) {
    if (!ctx || !frame || frame->ctx != ctx || (mode != 0 && mode != 1) || (frame->kind == 1 && mode == 1 && check_orientation) || nq < 0 || !nmatches_out ||
        (frame->n > 0 && !assigned_out) || (nq > 0 && (!queries || !qdesc))) {
        set_error("sslam_search_by_projection_frame: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    *nmatches_out = 0;
    for (int i = 0; i < frame->n; ++i) assigned_out[i] = -1;
    if (frame->n == 0 || nq == 0) return SSLAM_OK;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    return search_proj_core(ctx, frame->kind, mode, frame->feats.p, frame->desc.as<uint8_t>(), frame->n, frame->bounds,
                            frame->hasUright ? frame->uright.as<float>() : nullptr, occupied, queries, qdesc, nq, nnratio, th_dist, check_orientation,
                            assigned_out, nmatches_out);
}

extern "C" int sslam_hamming_knn2_frames(sslam_ctx* ctx, const sslam_frame* q, const sslam_frame* t, int32_t* idx, int32_t* dist) {
    if (!ctx || !q || !t || q->ctx != ctx || t->ctx != ctx || (q->n > 0 && (!idx || !dist))) { set_error("sslam_hamming_knn2_frames: invalid arguments"); return SSLAM_ERR_INVALID; }
    if (q->n == 0) return SSLAM_OK;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int rc;
    if ((rc = ctx->scratch[6].ensure(16 * (size_t)q->n))) return rc;
    int32_t* dI = ctx->scratch[6].as<int32_t>();
    int32_t* dD = dI + 2 * (size_t)q->n;
    if ((rc = sslam_hamming_knn2_dev(ctx, q->desc.as<uint8_t>(), q->n, t->desc.as<uint8_t>(), t->n, dI, dD, (void*)st))) return rc;
    SSLAM_HIP(hipMemcpyAsync(idx, dI, 8 * (size_t)q->n, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(dist, dD, 8 * (size_t)q->n, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    return SSLAM_OK;
}

static int search_by_bow_core(sslam_ctx* ctx, const sslam_keypoint* kf_kp, const uint8_t* kf_desc, const uint8_t* kf_valid, int nkf,
                              const sslam_keypoint* f_kp, const uint8_t* f_desc, const uint8_t* f_valid, int nf, const int32_t* node_kf_ptr, const int32_t* node_f_ptr,
                              int nnodes, const int32_t* kf_idx, const int32_t* f_idx, float nnratio, int check_orientation, int strict_th,
                              int32_t* assigned_out, int* nmatches_out) {
    if (!ctx || nkf < 0 || nf < 0 || nnodes < 0 || !nmatches_out || (nf > 0 && !assigned_out) ||
        (nnodes > 0 && (!node_kf_ptr || !node_f_ptr || !kf_idx || !f_idx || !kf_kp || !kf_desc || !kf_valid || !f_kp || !f_desc))) {
        set_error("sslam_orb_search_by_bow: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    *nmatches_out = 0;
    for (int i = 0; i < nf; ++i) assigned_out[i] = -1;
    if (nnodes == 0 || nf == 0 || nkf == 0) return SSLAM_OK;
    const int nk = node_kf_ptr[nnodes], nfi = node_f_ptr[nnodes];
    if (node_kf_ptr[0] != 0 || node_f_ptr[0] != 0 || nk < 0 || nfi < 0) { set_error("sslam_orb_search_by_bow: invalid node offsets"); return SSLAM_ERR_INVALID; }
    for (int i = 0; i < nk; ++i) if (kf_idx[i] < 0 || kf_idx[i] >= nkf) { set_error("sslam_orb_search_by_bow: keyframe feature index out of range"); return SSLAM_ERR_INVALID; }
    for (int i = 0; i < nfi; ++i) if (f_idx[i] < 0 || f_idx[i] >= nf) { set_error("sslam_orb_search_by_bow: frame feature index out of range"); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t ks = sizeof(sslam_keypoint);
    size_t o[16]; size_t off = 0; int k = 0;
    auto take = [&](size_t b) { o[k++] = off; off += al(b); };
    take(ks * nkf); take(32 * (size_t)nkf); take((size_t)nkf); take(ks * nf); take(32 * (size_t)nf);
    take(4 * (size_t)(nnodes + 1)); take(4 * (size_t)(nnodes + 1)); take(4 * (size_t)std::max(nk, 1)); take(4 * (size_t)std::max(nfi, 1));
    take(4 * (size_t)nf); take(256); take(4 * (size_t)nf); take((size_t)nf);
    int rc;
    if ((rc = ctx->scratch[7].ensure(off))) return rc;
    uint8_t* B = ctx->scratch[7].as<uint8_t>();
    // every input (and the -1 fill of the two result arrays) goes through ONE pinned staging buffer and one H2D copy: ten small pageable
    // copies cost more host time than the kernels take (0.11 ms of the call's 0.18)
    const void* src[9] = {kf_kp, kf_desc, kf_valid, f_kp, f_desc, node_kf_ptr, node_f_ptr, kf_idx, f_idx};
    const size_t len[9] = {ks * nkf, 32 * (size_t)nkf, (size_t)nkf, ks * nf, 32 * (size_t)nf, 4 * (size_t)(nnodes + 1), 4 * (size_t)(nnodes + 1), 4 * (size_t)nk, 4 * (size_t)nfi};
    if ((rc = ctx->pinned[1].ensure(off))) return rc;
    uint8_t* H = ctx->pinned[1].as<uint8_t>();
    for (int i = 0; i < 9; ++i) if (len[i]) memcpy(H + o[i], src[i], len[i]);
    memset(H + o[9], 0xFF, 4 * (size_t)nf); memset(H + o[10], 0, 256); memset(H + o[11], 0xFF, 4 * (size_t)nf);
    if (f_valid) memcpy(H + o[12], f_valid, (size_t)nf);
    SSLAM_HIP(hipMemcpyAsync(B, H, off, hipMemcpyHostToDevice, st));
    BowArgs A;
    A.validF = f_valid ? B + o[12] : nullptr; A.strictTh = strict_th;
    A.kpKF = (const sslam_keypoint*)(B + o[0]); A.dKF = B + o[1]; A.validKF = B + o[2]; A.kpF = (const sslam_keypoint*)(B + o[3]); A.dF = B + o[4]; A.nF = nf;
    A.ptrKF = (const int*)(B + o[5]); A.ptrF = (const int*)(B + o[6]); A.nnodes = nnodes; A.idxKF = (const int*)(B + o[7]); A.idxF = (const int*)(B + o[8]);
    A.nnratio = nnratio; A.checkOri = check_orientation; A.assigned = (int*)(B + o[9]); A.nmatches = (int*)(B + o[10]); A.qbin = (int*)(B + o[11]);
    bool disjoint = true;                     // DBoW2 puts a feature under exactly one node; if a caller's lists do not, replay in order
    {
        std::vector<uint8_t> seen((size_t)nf, 0);
        for (int i = 0; i < nfi && disjoint; ++i) { if (seen[f_idx[i]]) disjoint = false; seen[f_idx[i]] = 1; }
    }
    { sslam::ProfScope _ps(ctx, "k_search_bow", st); hipLaunchKernelGGL(k_search_bow, dim3(disjoint ? std::min(nnodes, 4096) : 1), dim3(64), 0, st, A); }
    // (a fused form -- the last workgroup to finish runs this pass, saving the launch -- was measured: the two agent-scope fences per
    // workgroup cost more than the launch, 0.111 against 0.103 ms per call; both passes in ONE workgroup of sixteen waves, no fences at
    // all: 0.235 ms -- two nodes per wave in series instead of one workgroup per node)
    { sslam::ProfScope _ps(ctx, "k_bow_finish", st); hipLaunchKernelGGL(k_bow_finish, dim3(1), dim3(256), 0, st, A.assigned, A.qbin, nf, check_orientation, A.nmatches); }
    SSLAM_HIP(hipGetLastError());
    SSLAM_HIP(hipMemcpyAsync(H + o[9], B + o[9], o[10] + 4 - o[9], hipMemcpyDeviceToHost, st));      // assigned + the count, back through the same staging
    SSLAM_HIP(hipStreamSynchronize(st));
    memcpy(assigned_out, H + o[9], 4 * (size_t)nf);
    memcpy(nmatches_out, H + o[10], sizeof(int));
    return SSLAM_OK;
}

extern "C" int sslam_orb_search_by_bow(sslam_ctx* ctx, const sslam_keypoint* kf_kp, const uint8_t* kf_desc, const uint8_t* kf_valid, int nkf,
                                       const sslam_keypoint* f_kp, const uint8_t* f_desc, int nf, const int32_t* node_kf_ptr, const int32_t* node_f_ptr,
                                       int nnodes, const int32_t* kf_idx, const int32_t* f_idx, float nnratio, int check_orientation,
                                       int32_t* assigned_out, int* nmatches_out) {
    return search_by_bow_core(ctx, kf_kp, kf_desc, kf_valid, nkf, f_kp, f_desc, nullptr, nf, node_kf_ptr, node_f_ptr, nnodes, kf_idx, f_idx, nnratio, check_orientation, 0,
                              assigned_out, nmatches_out);
}

// ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12), src/ORBmatcher.cc:525-658: the same walk with the
// second keyframe in the frame's role, candidates limited to features that own a good map point, `dist < TH_LOW`, and the result indexed by
// the first keyframe's features (every feature sits in one node and a taken candidate is skipped, so the map is one-to-one).
extern "C" int sslam_orb_search_by_bow_keyframes(sslam_ctx* ctx, const sslam_keypoint* kf1_kp, const uint8_t* kf1_desc, const uint8_t* kf1_valid, int n1,
                                                 const sslam_keypoint* kf2_kp, const uint8_t* kf2_desc, const uint8_t* kf2_valid, int n2,
                                                 const int32_t* node_kf1_ptr, const int32_t* node_kf2_ptr, int nnodes, const int32_t* kf1_idx, const int32_t* kf2_idx,
                                                 float nnratio, int check_orientation, int32_t* matches12_out, int* nmatches_out) {
    if (n1 < 0 || (n1 > 0 && !matches12_out) || (n2 > 0 && nnodes > 0 && !kf2_valid)) { set_error("sslam_orb_search_by_bow_keyframes: invalid arguments"); return SSLAM_ERR_INVALID; }
    for (int i = 0; i < n1; ++i) matches12_out[i] = -1;
    std::vector<int32_t> assigned((size_t)std::max(n2, 1), -1);
    const int rc = search_by_bow_core(ctx, kf1_kp, kf1_desc, kf1_valid, n1, kf2_kp, kf2_desc, kf2_valid, n2, node_kf1_ptr, node_kf2_ptr, nnodes, kf1_idx, kf2_idx, nnratio,
                                      check_orientation, 1, assigned.data(), nmatches_out);
    if (rc) return rc;
    for (int j = 0; j < n2; ++j) if (assigned[j] >= 0) matches12_out[assigned[j]] = j;
    return SSLAM_OK;
}


// MapPoint / MapLine ::ComputeDistinctiveDescriptors for `nsets` observation sets at once: set s owns descriptor rows
// ptr[s] .. ptr[s+1] of `desc`; best_out[s] = index (inside the set) of the descriptor with the least median distance to
// the rest, first such row on ties, -1 for an empty set.
extern "C" int sslam_distinctive_descriptors(sslam_ctx* ctx, const uint8_t* desc, const int32_t* ptr, int nsets, int32_t* best_out) {
    if (!ctx || nsets < 0 || (nsets > 0 && (!ptr || !best_out))) { set_error("sslam_distinctive_descriptors: invalid arguments"); return SSLAM_ERR_INVALID; }
    if (nsets == 0) return SSLAM_OK;
    const int total = ptr[nsets];
    if (ptr[0] != 0 || total < 0 || (total > 0 && !desc)) { set_error("sslam_distinctive_descriptors: invalid set offsets"); return SSLAM_ERR_INVALID; }
    for (int s2 = 0; s2 < nsets; ++s2) {
        const int n = ptr[s2 + 1] - ptr[s2];
        if (n < 0) { set_error("sslam_distinctive_descriptors: offsets must be non-decreasing"); return SSLAM_ERR_INVALID; }
        if (n > DISTINCT_MAXN) { set_error("sslam_distinctive_descriptors: a set of %d descriptors exceeds the supported %d", n, DISTINCT_MAXN); return SSLAM_ERR_UNSUPPORTED; }
    }
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t oD = 0, oP = oD + al(32 * (size_t)std::max(total, 1)), oB = oP + al(4 * (size_t)(nsets + 1)), tot = oB + al(4 * (size_t)nsets);
    int rc;
    if ((rc = ctx->scratch[6].ensure(tot))) return rc;
    uint8_t* B = ctx->scratch[6].as<uint8_t>();
    if (total > 0) SSLAM_HIP(hipMemcpyAsync(B + oD, desc, 32 * (size_t)total, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oP, ptr, 4 * (size_t)(nsets + 1), hipMemcpyHostToDevice, st));
    { sslam::ProfScope _ps(ctx, "k_distinctive", st); hipLaunchKernelGGL(k_distinctive, dim3(std::min(nsets, 4096)), dim3(64), 0, st, B + oD, (const int32_t*)(B + oP), nsets, (int32_t*)(B + oB)); }
    SSLAM_HIP(hipGetLastError());
    SSLAM_HIP(hipMemcpyAsync(best_out, B + oB, 4 * (size_t)nsets, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    return SSLAM_OK;
}

// The candidate search of ORBmatcher::Fuse (both overloads) and LSDmatcher::Fuse on a device-resident keyframe: per query the
// feature with the smallest descriptor distance inside the window (first in GetFeaturesInArea / GetLinesInArea order on ties),
// -1 / INT_MAX when the window holds no admissible feature.  The caller applies `bestDist <= TH_LOW` and the Replace /
// AddObservation bookkeeping in query order, exactly as the reference does after its inner loop.
extern "C" int sslam_fuse_search(sslam_ctx* ctx, const sslam_frame* kf, int chi2_mode, const float* inv_level_sigma2, int nlevels,
                                 const sslam_proj_query* queries, const uint8_t* qdesc, int nq, int32_t* best_idx_out, int32_t* best_dist_out) {
    if (!ctx || !kf || kf->ctx != ctx || (chi2_mode != 0 && chi2_mode != 1) || nq < 0 || (nq > 0 && (!queries || !qdesc || !best_idx_out || !best_dist_out)) ||
        (chi2_mode == 1 && (kf->kind != 0 || !inv_level_sigma2 || nlevels <= 0 || nlevels > 64))) {
        set_error("sslam_fuse_search: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    for (int i = 0; i < nq; ++i) { best_idx_out[i] = -1; best_dist_out[i] = 0x7fffffff; }
    if (nq == 0 || kf->n == 0) return SSLAM_OK;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t oQ = 0, oQD = oQ + al(sizeof(sslam_proj_query) * (size_t)nq), oS = oQD + al(32 * (size_t)nq), oI = oS + 256, oD = oI + al(4 * (size_t)nq),
                 total = oD + al(4 * (size_t)nq);
    int rc;
    if ((rc = ctx->scratch[6].ensure(total))) return rc;
    uint8_t* B = ctx->scratch[6].as<uint8_t>();
    SSLAM_HIP(hipMemcpyAsync(B + oQ, queries, sizeof(sslam_proj_query) * (size_t)nq, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oQD, qdesc, 32 * (size_t)nq, hipMemcpyHostToDevice, st));
    if (chi2_mode) SSLAM_HIP(hipMemcpyAsync(B + oS, inv_level_sigma2, sizeof(float) * (size_t)nlevels, hipMemcpyHostToDevice, st));
    FuseArgs A;
    A.kind = kf->kind; A.chi2 = chi2_mode; A.feats = kf->feats.p; A.desc = kf->desc.as<uint8_t>(); A.n = kf->n;
    A.minX = kf->bounds[0]; A.maxX = kf->bounds[1]; A.minY = kf->bounds[2]; A.maxY = kf->bounds[3];
    A.uright = kf->hasUright ? kf->uright.as<float>() : nullptr; A.invSigma2 = (const float*)(B + oS); A.nlevels = nlevels;
    A.q = (const sslam_proj_query*)(B + oQ); A.qdesc = B + oQD; A.nq = nq; A.bestIdx = (int*)(B + oI); A.bestDist = (int*)(B + oD);
    { sslam::ProfScope _ps(ctx, "k_fuse_search", st); hipLaunchKernelGGL(k_fuse_search, dim3(std::min(nq, 8192)), dim3(64), 0, st, A); }
    SSLAM_HIP(hipGetLastError());
    SSLAM_HIP(hipMemcpyAsync(best_idx_out, B + oI, 4 * (size_t)nq, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(best_dist_out, B + oD, 4 * (size_t)nq, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    return SSLAM_OK;
}

// ORBmatcher::SearchForTriangulation on two device-resident keyframes (their mvuRight travel with the handles).
extern "C" int sslam_orb_search_for_triangulation(sslam_ctx* ctx, const sslam_frame* kf1, const sslam_frame* kf2, const uint8_t* free1, const uint8_t* free2,
                                                  const int32_t* node_kf1_ptr, const int32_t* node_kf2_ptr, int nnodes, const int32_t* kf1_idx, const int32_t* kf2_idx,
                                                  const float F12[9], float ex, float ey, const float* scale_factors2, const float* level_sigma2_2, int nlevels,
                                                  int only_stereo, int check_orientation, int32_t* matches12_out, int* nmatches_out) {
    if (!ctx || !kf1 || !kf2 || kf1->ctx != ctx || kf2->ctx != ctx || kf1->kind != 0 || kf2->kind != 0 || nnodes < 0 || !nmatches_out || !F12 ||
        !scale_factors2 || !level_sigma2_2 || nlevels <= 0 || nlevels > 64 || (kf1->n > 0 && (!free1 || !matches12_out)) || (kf2->n > 0 && !free2) ||
        (nnodes > 0 && (!node_kf1_ptr || !node_kf2_ptr || !kf1_idx || !kf2_idx))) {
        set_error("sslam_orb_search_for_triangulation: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    *nmatches_out = 0;
    const int n1 = kf1->n, n2 = kf2->n;
    for (int i = 0; i < n1; ++i) matches12_out[i] = -1;
    if (n1 == 0 || n2 == 0 || nnodes == 0) return SSLAM_OK;
    const int total1 = node_kf1_ptr[nnodes], total2 = node_kf2_ptr[nnodes];
    if (node_kf1_ptr[0] != 0 || node_kf2_ptr[0] != 0 || total1 < 0 || total2 < 0) { set_error("sslam_orb_search_for_triangulation: invalid node offsets"); return SSLAM_ERR_INVALID; }
    for (int i = 0; i < total1; ++i) if (kf1_idx[i] < 0 || kf1_idx[i] >= n1) { set_error("sslam_orb_search_for_triangulation: keyframe-1 index out of range"); return SSLAM_ERR_INVALID; }
    for (int i = 0; i < total2; ++i) if (kf2_idx[i] < 0 || kf2_idx[i] >= n2) { set_error("sslam_orb_search_for_triangulation: keyframe-2 index out of range"); return SSLAM_ERR_INVALID; }
    if (total1 == 0 || total2 == 0) return SSLAM_OK;
    std::vector<int32_t> nodeOf((size_t)total1);
    for (int nd = 0; nd < nnodes; ++nd) for (int a = node_kf1_ptr[nd]; a < node_kf1_ptr[nd + 1]; ++a) nodeOf[a] = nd;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al(bytes); return o; };
    const size_t oF1 = take(n1), oF2 = take(n2), oP1 = take(4 * (size_t)(nnodes + 1)), oP2 = take(4 * (size_t)(nnodes + 1)), oI1 = take(4 * (size_t)total1),
                 oI2 = take(4 * (size_t)total2), oNO = take(4 * (size_t)total1), oSF = take(4 * (size_t)nlevels), oSG = take(4 * (size_t)nlevels),
                 oM = take(4 * (size_t)n1), oQB = take(4 * (size_t)n1), oN = take(4);
    int rc;
    if ((rc = ctx->scratch[6].ensure(off))) return rc;
    uint8_t* B = ctx->scratch[6].as<uint8_t>();
    SSLAM_HIP(hipMemcpyAsync(B + oF1, free1, n1, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oF2, free2, n2, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oP1, node_kf1_ptr, 4 * (size_t)(nnodes + 1), hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oP2, node_kf2_ptr, 4 * (size_t)(nnodes + 1), hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oI1, kf1_idx, 4 * (size_t)total1, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oI2, kf2_idx, 4 * (size_t)total2, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oNO, nodeOf.data(), 4 * (size_t)total1, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oSF, scale_factors2, 4 * (size_t)nlevels, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(B + oSG, level_sigma2_2, 4 * (size_t)nlevels, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemsetAsync(B + oM, 0xFF, 4 * (size_t)n1, st));
    TriArgs A;
    A.kp1 = kf1->feats.as<sslam_keypoint>(); A.d1 = kf1->desc.as<uint8_t>(); A.ur1 = kf1->hasUright ? kf1->uright.as<float>() : nullptr; A.free1 = B + oF1; A.n1 = n1;
    A.kp2 = kf2->feats.as<sslam_keypoint>(); A.d2 = kf2->desc.as<uint8_t>(); A.ur2 = kf2->hasUright ? kf2->uright.as<float>() : nullptr; A.free2 = B + oF2;
    A.ptr1 = (const int*)(B + oP1); A.ptr2 = (const int*)(B + oP2); A.nnodes = nnodes; A.idx1 = (const int*)(B + oI1); A.idx2 = (const int*)(B + oI2);
    A.nodeOf = (const int*)(B + oNO); A.total1 = total1;
    for (int i = 0; i < 9; ++i) A.F[i] = F12[i];
    A.ex = ex; A.ey = ey; A.scale2 = (const float*)(B + oSF); A.sigma2_2 = (const float*)(B + oSG); A.nlevels = nlevels;
    A.onlyStereo = only_stereo; A.checkOri = check_orientation;
    A.m12 = (int*)(B + oM); A.qbin = (int*)(B + oQB); A.nmatches = (int*)(B + oN);
    { sslam::ProfScope _ps(ctx, "k_tri_search", st); hipLaunchKernelGGL(k_tri_search, dim3(std::min(total1, 8192)), dim3(64), 0, st, A); }
    { sslam::ProfScope _ps(ctx, "k_tri_finish", st); hipLaunchKernelGGL(k_tri_finish, dim3(1), dim3(256), 0, st, A.m12, A.qbin, n1, check_orientation, A.nmatches); }
    SSLAM_HIP(hipGetLastError());
    SSLAM_HIP(hipMemcpyAsync(matches12_out, B + oM, 4 * (size_t)n1, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(nmatches_out, B + oN, sizeof(int), hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    return SSLAM_OK;
}

// ---- DBoW2 vocabulary
extern "C" int sslam_vocab_create(sslam_ctx* ctx, int nnodes, int levels, const int32_t* child_ptr, const int32_t* children, const uint8_t* node_desc,
                                  const int32_t* word_id, const double* weight, sslam_vocab** out) {
    if (!ctx || !out || nnodes < 1 || levels < 1 || !child_ptr || !node_desc || !word_id || !weight) { set_error("sslam_vocab_create: invalid arguments"); return SSLAM_ERR_INVALID; }
    const int nch = child_ptr[nnodes];
    if (child_ptr[0] != 0 || nch < 0 || (nch > 0 && !children)) { set_error("sslam_vocab_create: invalid child offsets"); return SSLAM_ERR_INVALID; }
    for (int i = 0; i < nnodes; ++i) if (child_ptr[i + 1] < child_ptr[i]) { set_error("sslam_vocab_create: child offsets must be non-decreasing"); return SSLAM_ERR_INVALID; }
    for (int i = 0; i < nch; ++i) if (children[i] <= 0 || children[i] >= nnodes) { set_error("sslam_vocab_create: child id out of range"); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    sslam_vocab* v = new sslam_vocab();
    HandleGuard<sslam_vocab, sslam_vocab_destroy> guard(v);
    v->ctx = ctx; v->nnodes = nnodes; v->levels = levels;
    for (int i = 1; i < nnodes; ++i) v->nwords += child_ptr[i + 1] == child_ptr[i];
    int rc;
    if ((rc = v->childPtr.ensure(4 * (size_t)(nnodes + 1))) || (rc = v->children.ensure(std::max<size_t>(4 * (size_t)nch, 256))) || (rc = v->desc.ensure(32 * (size_t)nnodes)) ||
        (rc = v->wordId.ensure(4 * (size_t)nnodes)) || (rc = v->weight.ensure(8 * (size_t)nnodes))) return rc;
    hipStream_t st = ctx->stream;
    SSLAM_HIP(hipMemcpyAsync(v->childPtr.p, child_ptr, 4 * (size_t)(nnodes + 1), hipMemcpyHostToDevice, st));
    if (nch > 0) SSLAM_HIP(hipMemcpyAsync(v->children.p, children, 4 * (size_t)nch, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(v->desc.p, node_desc, 32 * (size_t)nnodes, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(v->wordId.p, word_id, 4 * (size_t)nnodes, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipMemcpyAsync(v->weight.p, weight, 8 * (size_t)nnodes, hipMemcpyHostToDevice, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    *out = guard.release();
    return SSLAM_OK;
}

extern "C" void sslam_vocab_destroy(sslam_vocab* v) {
    if (!v) return;
    if (v->ctx) (void)hipSetDevice(v->ctx->device);
    v->childPtr.release(); v->children.release(); v->desc.release(); v->wordId.release(); v->weight.release();
    delete v;
}

static int bow_core(sslam_ctx* ctx, const sslam_vocab* v, const uint8_t* d_desc, int n, int levelsup, int32_t* word_out, double* weight_out, int32_t* node_out) {
    hipStream_t st = ctx->stream;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t oW = 0, oV = oW + al(4 * (size_t)n), oN = oV + al(8 * (size_t)n), total = oN + al(4 * (size_t)n);
    int rc;
    if ((rc = ctx->scratch[6].ensure(total))) return rc;
    uint8_t* B = ctx->scratch[6].as<uint8_t>();
    { sslam::ProfScope _ps(ctx, "k_bow_transform", st);
      hipLaunchKernelGGL(k_bow_transform, dim3((n + 255) / 256), dim3(256), 0, st, d_desc, n, v->childPtr.as<int>(), v->children.as<int>(), v->desc.as<uint8_t>(),
                         v->wordId.as<int>(), v->weight.as<double>(), v->levels - levelsup, (int*)(B + oW), (double*)(B + oV), (int*)(B + oN)); }
    SSLAM_HIP(hipGetLastError());
    SSLAM_HIP(hipMemcpyAsync(word_out, B + oW, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(weight_out, B + oV, 8 * (size_t)n, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(node_out, B + oN, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    return SSLAM_OK;
}

extern "C" int sslam_bow_transform_frame(sslam_ctx* ctx, const sslam_vocab* vocab, const sslam_frame* frame, int levelsup,
                                         int32_t* word_out, double* weight_out, int32_t* node_out) {
    if (!ctx || !vocab || !frame || vocab->ctx != ctx || frame->ctx != ctx || levelsup < 0 || (frame->n > 0 && (!word_out || !weight_out || !node_out))) {
        set_error("sslam_bow_transform_frame: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    if (frame->n == 0) return SSLAM_OK;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    return bow_core(ctx, vocab, frame->desc.as<uint8_t>(), frame->n, levelsup, word_out, weight_out, node_out);
}

extern "C" int sslam_bow_transform(sslam_ctx* ctx, const sslam_vocab* vocab, const uint8_t* desc, int n, int levelsup,
                                   int32_t* word_out, double* weight_out, int32_t* node_out) {
    if (!ctx || !vocab || vocab->ctx != ctx || n < 0 || levelsup < 0 || (n > 0 && (!desc || !word_out || !weight_out || !node_out))) {
        set_error("sslam_bow_transform: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    if (n == 0) return SSLAM_OK;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SSLAM_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = ctx->scratch[7].ensure(32 * (size_t)n))) return rc;
    SSLAM_HIP(hipMemcpyAsync(ctx->scratch[7].p, desc, 32 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    return bow_core(ctx, vocab, ctx->scratch[7].as<uint8_t>(), n, levelsup, word_out, weight_out, node_out);
}
