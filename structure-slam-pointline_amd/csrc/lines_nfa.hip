// The NFA stage of the line front-end (rect_nfa / rect_improve of LSD_REFINE_ADV: lsd_nfa.h) as a translation unit of its own.
// Why: build.py compiles THIS unit with -mllvm -disable-machine-licm.  The stage is fp64 polynomial code (log-gamma terms, log10, the binomial
// tail) inside loops; with machine LICM the compiler hoists the polynomials' constants out of the loops into registers the fused kernel does
// not have at four waves per SIMD, spills them, and reloads them from scratch one at a time with a wait each -- 208 bytes of scratch and 213
// scratch loads in k_nfa_all.  Without it the constants are rematerialised where they are used: 16 bytes of scratch, 28.4 -> 26.8 ms per
// 12 288 frames.  The sequential core in lines.hip is the other way round (78.9 -> 79.4 ms without the hoisting), hence two units
// (docs/history/DESIGN_rounds_1-4.md 5g).  The kernels are launched from here; lines.hip calls sslam::launch_nfa_stage.
#include "common.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cfloat>
#include <algorithm>

using namespace sslam;

namespace {

#define SSLAM_NFA_STAGE_ONLY 1
#include "lsd_plan.h"
#include "lsd_nfa.h"

}  // namespace

namespace sslam {

int launch_nfa_stage(sslam_ctx* ctx, hipStream_t st, uint8_t* ws, const void* plan, size_t planBytes, const double* lgam, int nframes) {
    if (planBytes != sizeof(LsdPlan)) { set_error("launch_nfa_stage: plan layout mismatch between translation units"); return SSLAM_ERR_INVALID; }
    LsdPlan P; memcpy(&P, plan, sizeof(P));
    int evalWaves = nframes >= 1024 ? 1 : nframes >= 64 ? 4 : nframes >= 16 ? 16 : 32;      // waves per frame walking the NFA evaluations
    int countWaves = nframes >= 2048 ? 1 : nframes >= 128 ? 8 : nframes >= 16 ? 64 : 128;   // waves per frame walking the rectangle counts (round 4: 128 / 32 for a handful of frames, 6.25 -> 6.09 ms per frame)
    if (const char* e = getenv("SSLAM_COUNT_WAVES")) countWaves = std::max(1, atoi(e));
    if (const char* e = getenv("SSLAM_EVAL_WAVES")) evalWaves = std::max(1, atoi(e));
    // The whole NFA stage as ONE launch (lsd_nfa.h): one wave per frame when the frames themselves fill the chip (k_nfa_all, calls of >= 2048 frames).
    // Below that the 18 launches stay: a single frame's stage is bound by the work of each wave, not by launch boundaries (kernel durations add up
    // to the stage's 0.6 ms; a one-workgroup form with sixteen waves was measured at 6.42 / 7.70 ms p50 / p90 per frame against 6.25 / 7.40 and removed in round 5).
    // What helped instead: 128 counting and 32 evaluating waves per frame (6.09 / 7.16).
    // SSLAM_NFA_FUSED=0 forces the launches, =2 the one-wave form (tests).
    bool nfaFused = countWaves == 1 && evalWaves == 1;
    if (const char* e = getenv("SSLAM_NFA_FUSED")) nfaFused = atoi(e) == 2 || (nfaFused && atoi(e) != 0);
    if (nfaFused) {
        sslam::ProfScope _ps(ctx, "k_nfa_all", st);
        hipLaunchKernelGGL(k_nfa_all<768>, dim3(nframes), dim3(64), 0, st, ws, P, lgam);
    } else {
        for (int stage = 0; stage <= 4; ++stage) {
            { static const char* kCountNames[5] = {"k_nfa_count", "k_nfa_count/s1", "k_nfa_count/s2", "k_nfa_count/s3", "k_nfa_count/s4"};
              sslam::ProfScope _ps(ctx, getenv("SSLAM_PROF_STAGES") ? kCountNames[stage] : "k_nfa_count", st); hipLaunchKernelGGL(k_nfa_count, dim3(countWaves, nframes), dim3(64), 0, st, ws, P, stage); }
            if (stage == 0) {
                { sslam::ProfScope _ps(ctx, "k_nfa_eval", st); hipLaunchKernelGGL(k_nfa_eval, dim3(evalWaves, nframes), dim3(64), 0, st, ws, P, -1, lgam); }
                { sslam::ProfScope _ps(ctx, "k_nfa_accept", st); hipLaunchKernelGGL(k_nfa_accept, dim3(4, nframes), dim3(256), 0, st, ws, P, -1); }
            }
            { sslam::ProfScope _ps(ctx, "k_nfa_eval", st); hipLaunchKernelGGL(k_nfa_eval, dim3(evalWaves, nframes), dim3(64), 0, st, ws, P, stage, lgam); }
            { sslam::ProfScope _ps(ctx, "k_nfa_accept", st); hipLaunchKernelGGL(k_nfa_accept, dim3(4, nframes), dim3(256), 0, st, ws, P, stage); }
        }
        { sslam::ProfScope _ps(ctx, "k_nfa_finish", st); hipLaunchKernelGGL(k_nfa_finish, dim3(4, nframes), dim3(256), 0, st, ws, P); }
    }
    SSLAM_HIP(hipGetLastError());
    return SSLAM_OK;
}

// The streaming form (lsd_nfa.h, k_nfa_stream; SSLAM_NFA_STREAM=1): `waves` single-wave workgroups per frame.  Called twice per extraction: on the second stream next
// to the core (spinTicks > 0: the waves wait for the main wave's rectangles), and on the main stream behind both (spinTicks = 0: takes what is unclaimed and returns).
// ldsPad: dynamic LDS the kernel does not use -- next to the core it keeps the consumers off the compute units of the main wave and of the helpers (their workgroups hold
// > 120 KB of the 160; a consumer that asks for 40 KB does not fit beside them: a wave sharing the main wave's SIMD would slow the one chain the frame waits for).
int launch_nfa_stream(sslam_ctx* ctx, hipStream_t st, uint8_t* ws, const void* plan, size_t planBytes, const double* lgam, uint8_t* clArea, size_t clFrameBytes,
                      size_t stageOff, int nframes, int waves, long long spinTicks, size_t ldsPad, int takeMax, int sleepReps, const char* scope) {
    if (planBytes != sizeof(LsdPlan)) { set_error("launch_nfa_stream: plan layout mismatch between translation units"); return SSLAM_ERR_INVALID; }
    takeMax = std::max(1, std::min(NFA_STREAM_BLOCK, takeMax));
    if (waves < 1 || nframes < 1 || (stageOff & 7)) { set_error("launch_nfa_stream: invalid arguments"); return SSLAM_ERR_INVALID; }
    LsdPlan P; memcpy(&P, plan, sizeof(P));
    if (scope) { sslam::ProfScope _ps(ctx, scope, st); hipLaunchKernelGGL(k_nfa_stream, dim3(waves, nframes), dim3(64), ldsPad, st, ws, P, lgam, clArea, clFrameBytes, stageOff, spinTicks, takeMax, sleepReps); }
    else hipLaunchKernelGGL(k_nfa_stream, dim3(waves, nframes), dim3(64), ldsPad, st, ws, P, lgam, clArea, clFrameBytes, stageOff, spinTicks, takeMax, sleepReps);
    SSLAM_HIP(hipGetLastError());
    return SSLAM_OK;
}

}  // namespace sslam
