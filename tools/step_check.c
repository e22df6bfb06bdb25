/* The bench's step (BASELINE configs[2]: ORB + LSD/LBD extract of a resident batch + matching against the previous frames' features, point and line branch on two HIP
 * streams, the point branch gated on the core event) through the C ABI + the HIP runtime alone -- what structure-slam-pointline_amd/pipeline.py FrontendBatch.step does with
 * torch as plumbing, without Python: a fresh GPU box runs it in seconds instead of the minutes of `import torch`, so a kernel experiment costs ~20 s of GPU time.
 *     python tools/step_check_prepare.py        (CPU: frames + the oracle's answers)
 *     gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/step_check.c -Lstructure-slam-pointline_amd/lib -lsslam_frontend -L/opt/rocm/lib -lamdhip64 \
 *         -Wl,-rpath,'$ORIGIN/../structure-slam-pointline_amd/lib' -Wl,-rpath,/opt/rocm/lib -o tools/step_check
 *     tools/step_check [batch=12288] [steps=5] [warmup=2] [one_stream=0]        (STEP_PROFILE=1: per launch scope, ms per step)
 * Prints frames/s (wall clock around the timed steps, device synchronised on both sides), mean keypoints / lines / matches per frame (bench.py's config block has the
 * same figures), and compares the first 16 frames' keypoints + descriptors and the first 64 frames' lines with the CPU oracle.  NOT the bench: bench.py is what the driver runs. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <hip/hip_runtime_api.h>
#include "sslam_frontend.h"

enum { W = 640, H = 480, U = 64, LCAP = 200, NREF = 16 };
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define SCHK(x) do { int r_ = (x); if (r_) { fprintf(stderr, "%s: %d %s\n", #x, r_, sslam_last_error()); exit(2); } } while (0)
static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
typedef struct { sslam_keypoint* kp; uint8_t* desc; int32_t* n; sslam_keyline* kl; uint8_t* ldesc; double* fn; int32_t* nl; } Feat;

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 12288, steps = argc > 2 ? atoi(argv[2]) : 5, warm = argc > 3 ? atoi(argv[3]) : 2, one = argc > 4 ? atoi(argv[4]) : 0;
    const size_t fsz = (size_t)W * H;
    unsigned char* host = malloc(2 * U * fsz);
    FILE* f = fopen("tools/step_frames.raw", "rb");
    if (!f || fread(host, 1, 2 * U * fsz, f) != 2 * U * fsz) { fprintf(stderr, "tools/step_frames.raw missing (tools/step_check_prepare.py)\n"); return 2; }
    fclose(f);
    HIPCHK(hipSetDevice(0));
    sslam_ctx* ctx = NULL; sslam_orb* orb = NULL; sslam_lines* ln = NULL;
    SCHK(sslam_ctx_create(0, &ctx)); SCHK(sslam_orb_create(ctx, 1000, 1.2f, 8, 20, 7, &orb)); SCHK(sslam_lines_create(ctx, LCAP, &ln));
    if (getenv("STEP_NFA_VARIANT")) SCHK(sslam_lines_set_nfa_variant(ln, atoi(getenv("STEP_NFA_VARIANT"))));      /* decision D11's other form: timing only (the expected lines are the default's) */
    const int cap = sslam_orb_max_keypoints(orb);
    unsigned char *dCur, *dPrev;
    HIPCHK(hipMalloc((void**)&dCur, (size_t)B * fsz)); HIPCHK(hipMalloc((void**)&dPrev, (size_t)B * fsz));
    HIPCHK(hipMemcpy(dCur, host, (size_t)(B < U ? B : U) * fsz, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dPrev, host + U * fsz, (size_t)(B < U ? B : U) * fsz, hipMemcpyHostToDevice));
    for (size_t have = U; have < (size_t)B; have *= 2) {      /* tile the 64 scenes to the batch */
        const size_t n = have * 2 <= (size_t)B ? have : (size_t)B - have;
        HIPCHK(hipMemcpy(dCur + have * fsz, dCur, n * fsz, hipMemcpyDeviceToDevice)); HIPCHK(hipMemcpy(dPrev + have * fsz, dPrev, n * fsz, hipMemcpyDeviceToDevice));
    }
    Feat ft[2];
    for (int k = 0; k < 2; ++k) {
        HIPCHK(hipMalloc((void**)&ft[k].kp, (size_t)B * cap * sizeof(sslam_keypoint))); HIPCHK(hipMalloc((void**)&ft[k].desc, (size_t)B * cap * 32)); HIPCHK(hipMalloc((void**)&ft[k].n, B * 4));
        HIPCHK(hipMalloc((void**)&ft[k].kl, (size_t)B * LCAP * sizeof(sslam_keyline))); HIPCHK(hipMalloc((void**)&ft[k].ldesc, (size_t)B * LCAP * 32)); HIPCHK(hipMalloc((void**)&ft[k].fn, (size_t)B * LCAP * 24));
        HIPCHK(hipMalloc((void**)&ft[k].nl, B * 4));
        HIPCHK(hipMemset(ft[k].n, 0, B * 4)); HIPCHK(hipMemset(ft[k].nl, 0, B * 4));
    }
    float *pm, *pm0; int32_t *m12, *nmatch, *knnIdx, *knnDist, *lpairs, *nlpairs;
    HIPCHK(hipMalloc((void**)&pm, (size_t)B * cap * 8)); HIPCHK(hipMalloc((void**)&pm0, (size_t)B * cap * 8)); HIPCHK(hipMalloc((void**)&m12, (size_t)B * cap * 4)); HIPCHK(hipMalloc((void**)&nmatch, B * 4));
    HIPCHK(hipMalloc((void**)&knnIdx, (size_t)B * cap * 8)); HIPCHK(hipMalloc((void**)&knnDist, (size_t)B * cap * 8));
    HIPCHK(hipMalloc((void**)&lpairs, (size_t)B * LCAP * 8)); HIPCHK(hipMalloc((void**)&nlpairs, B * 4));
    hipStream_t s1, s2; hipEvent_t core, e1, e2;
    /* STEP_POINT_PRIO / STEP_LINE_PRIO: HIP stream priorities of the point / line branch (0 = default, -1 = high, 1 = low) */
    HIPCHK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, getenv("STEP_POINT_PRIO") ? atoi(getenv("STEP_POINT_PRIO")) : 0));
    HIPCHK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, getenv("STEP_LINE_PRIO") ? atoi(getenv("STEP_LINE_PRIO")) : 0));
    HIPCHK(hipEventCreateWithFlags(&core, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    const float bounds[4] = {0.f, (float)W, 0.f, (float)H};
    Feat *P = &ft[1], *C = &ft[0];
    /* the previous frames' features, once (bench.py: pipe.extract(prev, "prev")) */
    SCHK(sslam_orb_extract_batch_dev(orb, dPrev, W, H, W, fsz, B, P->kp, P->desc, P->n, cap, s1));
    SCHK(sslam_lines_extract_batch_dev(ln, dPrev, W, H, W, fsz, B, P->kl, P->ldesc, P->fn, P->nl, LCAP, s1));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy2D(pm0, 8, P->kp, sizeof(sslam_keypoint), 8, (size_t)B * cap, hipMemcpyDeviceToDevice));      /* vbPrevMatched starts at F1's keypoint positions (Tracking.cc:340-342); a pristine copy, restored every step */
    /* STEP_GATE: when the point branch starts -- "pyr" (default, as pipeline.py): the pyramid at once (beside the line prologue), FAST and what follows with the sequential LSD
     * core (sslam_orb_set_gate_event); "core": all of it with the core (rounds 3-5); "none": everything at once */
    const char* gate = getenv("STEP_GATE") ? getenv("STEP_GATE") : "auto";
    if (!one && strcmp(gate, "none")) {
        HIPCHK(hipEventRecord(core, s2)); SCHK(sslam_lines_set_core_event(ln, core));
        if (!strcmp(gate, "auto")) gate = sslam_lines_core_guest_form(ln, B) ? "pyr" : "core";      /* as pipeline.py */
        if (!strcmp(gate, "pyr")) SCHK(sslam_orb_set_gate_event(orb, core));
    }
    double t0 = 0;
    for (int it = 0; it < warm + steps; ++it) {
        if (it == warm) { HIPCHK(hipDeviceSynchronize()); if (getenv("STEP_PROFILE")) sslam_profile_enable(ctx, 1); t0 = now_ms(); }
        hipStream_t sl = one ? s1 : s2;
        SCHK(sslam_lines_extract_batch_dev(ln, dCur, W, H, W, fsz, B, C->kl, C->ldesc, C->fn, C->nl, LCAP, sl));
        SCHK(sslam_line_match_batch_dev(ctx, P->ldesc, P->nl, C->ldesc, C->nl, LCAP, B, 0.5, 0, lpairs, nlpairs, sl));
        if (!one && !strcmp(gate, "core")) HIPCHK(hipStreamWaitEvent(s1, core, 0));      /* the point branch starts when the sequential LSD core does */
        SCHK(sslam_orb_extract_batch_dev(orb, dCur, W, H, W, fsz, B, C->kp, C->desc, C->n, cap, s1));
        HIPCHK(hipMemcpyAsync(pm, pm0, (size_t)B * cap * 8, hipMemcpyDeviceToDevice, s1));      /* (pipeline.py copies the positions out of the keypoint records here: the same bytes) */
        SCHK(sslam_orb_search_for_initialization_batch_dev(ctx, P->kp, P->desc, P->n, C->kp, C->desc, C->n, cap, B, pm, m12, nmatch, 100, 0.9f, 1, bounds, s1));
        SCHK(sslam_hamming_knn2_batch_dev(ctx, P->desc, P->n, C->desc, C->n, cap, B, knnIdx, knnDist, s1));
        if (!one) {      /* join: the next step's branches start behind both */
            HIPCHK(hipEventRecord(e1, s1)); HIPCHK(hipEventRecord(e2, s2));
            HIPCHK(hipStreamWaitEvent(s1, e2, 0)); HIPCHK(hipStreamWaitEvent(s2, e1, 0));
        }
    }
    HIPCHK(hipDeviceSynchronize());
    const double dt = now_ms() - t0;
    printf("batch %d, %d steps (+%d warm-up), %s: %.1f ms per step, %.0f frames/s\n", B, steps, warm, one ? "one stream" : "two streams", dt / steps, (double)B * steps / dt * 1e3);
    if (getenv("STEP_PROFILE")) {
        const char* nm[64]; double ms[64]; int lc[64];
        const int nk = sslam_profile_drain(ctx, nm, ms, lc, 64); sslam_profile_enable(ctx, 0);
        printf("launch scopes, ms per step:");
        for (int k = 0; k < nk && k < 64; ++k) if (ms[k] / steps >= 0.3) printf(" %s %.1f", nm[k], ms[k] / steps);
        printf("\n");
    }
    /* sanity + parity */
    const int R = B < U ? B : U;
    int32_t *hn = malloc(B * 4), *hnl = malloc(B * 4), *hnm = malloc(B * 4), *hnlp = malloc(B * 4);
    HIPCHK(hipMemcpy(hn, C->n, B * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(hnl, C->nl, B * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hnm, nmatch, B * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(hnlp, nlpairs, B * 4, hipMemcpyDeviceToHost));
    double a = 0, b = 0, c = 0, d = 0; int rep_bad = 0;
    for (int i = 0; i < B; ++i) { a += hn[i]; b += hnl[i]; c += hnm[i]; d += hnlp[i]; if (hn[i] != hn[i % U] || hnl[i] != hnl[i % U] || hnm[i] != hnm[i % U]) ++rep_bad; }
    printf("per frame: %.3f keypoints, %.3f lines, %.3f ORB matches, %.3f line matches; copies of a scene with other counts: %d\n", a / B, b / B, c / B, d / B, rep_bad);
    int bad_orb = 0, bad_ln = 0;
    f = fopen("tools/step_expected_orb.bin", "rb");
    if (f) {
        sslam_keypoint* wk = malloc(cap * sizeof(sslam_keypoint)); uint8_t* wd = malloc((size_t)cap * 32);
        sslam_keypoint* gk = malloc(cap * sizeof(sslam_keypoint)); uint8_t* gd = malloc((size_t)cap * 32);
        for (int i = 0; i < NREF && i < R; ++i) {
            int n = 0; if (fread(&n, 4, 1, f) != 1 || n > cap) break;
            if (fread(wk, sizeof(sslam_keypoint), n, f) != (size_t)n || fread(wd, 32, n, f) != (size_t)n) break;
            HIPCHK(hipMemcpy(gk, C->kp + (size_t)i * cap, sizeof(sslam_keypoint) * n, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(gd, C->desc + (size_t)i * cap * 32, (size_t)n * 32, hipMemcpyDeviceToHost));
            bad_orb += !(hn[i] == n && !memcmp(gk, wk, sizeof(sslam_keypoint) * n) && !memcmp(gd, wd, (size_t)n * 32));
        }
        fclose(f);
    } else bad_orb = -1;
    f = fopen("tools/lat_expected.bin", "rb");
    if (f) {
        static sslam_keyline wkl[256], gkl[256]; static uint8_t wld[256 * 32], gld[256 * 32]; static double wfn[256 * 3], gfn[256 * 3];
        for (int i = 0; i < R; ++i) {
            int n = 0; if (fread(&n, 4, 1, f) != 1 || n > 256) break;
            if (fread(wkl, sizeof(sslam_keyline), n, f) != (size_t)n || fread(wld, 32, n, f) != (size_t)n || fread(wfn, 24, n, f) != (size_t)n) break;
            int ok = hnl[i] == n;
            if (ok) {
                HIPCHK(hipMemcpy(gkl, C->kl + (size_t)i * LCAP, sizeof(sslam_keyline) * n, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(gld, C->ldesc + (size_t)i * LCAP * 32, (size_t)n * 32, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(gfn, C->fn + (size_t)i * LCAP * 3, (size_t)n * 24, hipMemcpyDeviceToHost));
                ok = !memcmp(gld, wld, (size_t)n * 32) && !memcmp(gfn, wfn, (size_t)n * 24);
                for (int k = 0; ok && k < n; ++k) { sslam_keyline x = gkl[k], y = wkl[k]; x.angle = y.angle = 0; ok = !memcmp(&x, &y, sizeof(x)); }
            }
            bad_ln += !ok;
        }
        fclose(f);
    } else bad_ln = -1;
    printf("against the CPU oracle: %d of %d frames differ in keypoints / descriptors, %d of %d in lines (-1: expected file missing)\n", bad_orb, NREF < R ? NREF : R, bad_ln, R);
    return (bad_orb > 0 || bad_ln > 0 || rep_bad) ? 1 : 0;
}
