/* Small resident batches (2 .. 64 frames per call: the cluster form of the LSD core) and single frames under a concurrent ORB load, through the C ABI + the HIP
 * runtime alone (no Python on the GPU box), per knob setting, against the CPU oracle's precomputed lines of the bench's 64 frames (tools/lat_check_prepare.py):
 *     gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/batch_check.c -Lstructure-slam-pointline_amd/lib -lsslam_frontend -L/opt/rocm/lib -lamdhip64 -lpthread \
 *         -Wl,-rpath,'$ORIGIN/../structure-slam-pointline_amd/lib' -Wl,-rpath,/opt/rocm/lib -o tools/batch_check
 *     tools/batch_check "" "SSLAM_NFA_STREAM=1" ...
 * Part 1: sslam_lines_extract_batch_dev on the first n = 2, 8, 19, 41, 64 frames: every frame's keylines / LBD bytes / line functions against the oracle, ms per call.
 * Part 2: sslam_lines_extract frame by frame while a second thread keeps a second context busy with sslam_orb_extract_batch_dev of 64 frames (the consumers and the core's
 *         workgroups start late and share compute units): parity again. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>
#include <hip/hip_runtime_api.h>
#include "sslam_frontend.h"

enum { W = 640, H = 480, CAP = 256, NF = 64 };
typedef struct { int n; sslam_keyline kl[CAP]; unsigned char d[CAP * 32]; double fn[CAP * 3]; } Out;
static unsigned char img[NF][W * H];
static Out want[NF];
static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static int same(const Out* w, int n, const sslam_keyline* kl, const unsigned char* d, const double* fn) {
    if (n != w->n || memcmp(d, w->d, 32 * (size_t)n) || memcmp(fn, w->fn, 24 * (size_t)n)) return 0;
    for (int k = 0; k < n; ++k) { sslam_keyline x = kl[k], y = w->kl[k]; x.angle = y.angle = 0; if (memcmp(&x, &y, sizeof(x))) return 0; }      /* (KeyLine.angle: the suite allows 1 ulp) */
    return 1;
}

static volatile int orbStop = 0; static long orbCalls = 0;
static unsigned char* dImgs;
static void* orb_load(void* arg) {
    (void)arg;
    sslam_ctx* c2 = NULL; sslam_orb* ox = NULL;
    if (sslam_ctx_create(0, &c2) || sslam_orb_create(c2, 1000, 1.2f, 8, 20, 7, &ox)) { fprintf(stderr, "orb load: %s\n", sslam_last_error()); return NULL; }
    void *dkp, *ddesc, *dcnt;
    HIPCHK(hipMalloc(&dkp, (size_t)NF * 1100 * sizeof(sslam_keypoint))); HIPCHK(hipMalloc(&ddesc, (size_t)NF * 1100 * 32)); HIPCHK(hipMalloc(&dcnt, NF * 4));
    while (!orbStop) {
        if (sslam_orb_extract_batch_dev(ox, dImgs, W, H, W, (size_t)W * H, NF, (sslam_keypoint*)dkp, (uint8_t*)ddesc, (int32_t*)dcnt, 1100, NULL)) { fprintf(stderr, "orb batch: %s\n", sslam_last_error()); break; }
        sslam_ctx_synchronize(c2); ++orbCalls;
    }
    sslam_orb_destroy(ox); sslam_ctx_destroy(c2);
    return NULL;
}

int main(int argc, char** argv) {
    FILE* f = fopen("tools/lat_frames.raw", "rb");
    if (!f || fread(img, 1, sizeof(img), f) != sizeof(img)) { fprintf(stderr, "tools/lat_frames.raw missing (tools/lat_check_prepare.py)\n"); return 2; }
    fclose(f);
    f = fopen("tools/lat_expected.bin", "rb");
    if (!f) return 2;
    for (int i = 0; i < NF; ++i) {
        int n = 0; if (fread(&n, 4, 1, f) != 1 || n < 0 || n > CAP) return 2;
        want[i].n = n;
        if (fread(want[i].kl, sizeof(sslam_keyline), n, f) != (size_t)n || fread(want[i].d, 32, n, f) != (size_t)n || fread(want[i].fn, 24, n, f) != (size_t)n) return 2;
    }
    fclose(f);
    sslam_ctx* ctx = NULL; sslam_lines* L = NULL;
    if (sslam_ctx_create(0, &ctx) || sslam_lines_create(ctx, 200, &L)) { fprintf(stderr, "create: %s\n", sslam_last_error()); return 2; }
    void *dKl, *dD, *dFn, *dCnt;
    HIPCHK(hipMalloc((void**)&dImgs, sizeof(img))); HIPCHK(hipMemcpy(dImgs, img, sizeof(img), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&dKl, (size_t)NF * CAP * sizeof(sslam_keyline))); HIPCHK(hipMalloc(&dD, (size_t)NF * CAP * 32)); HIPCHK(hipMalloc(&dFn, (size_t)NF * CAP * 24)); HIPCHK(hipMalloc(&dCnt, NF * 4));
    static sslam_keyline hKl[NF * CAP]; static unsigned char hD[NF * CAP * 32]; static double hFn[NF * CAP * 3]; static int hCnt[NF];
    int bad_total = 0;
    const int sizes[5] = {2, 8, 19, 41, 64};
    for (int a = 1; a < argc || a == 1; ++a) {
        char buf[512]; const char* names[16]; int nn = 0;
        snprintf(buf, sizeof(buf), "%s", a < argc ? argv[a] : "");
        for (char* tok = strtok(buf, ","); tok && nn < 16; tok = strtok(NULL, ",")) { char* eq = strchr(tok, '='); if (!eq) continue; *eq = 0; setenv(tok, eq + 1, 1); names[nn++] = tok; }
        printf("%s\n", a < argc && argv[a][0] ? argv[a] : "(default)");
        for (int s = 0; s < 5; ++s) {
            const int nf = sizes[s]; int bad = 0; double best = 1e9;
            for (int r = 0; r < 4; ++r) {
                HIPCHK(hipMemset(dCnt, 0xFF, NF * 4));
                const double t0 = now_ms();
                if (sslam_lines_extract_batch_dev(L, dImgs, W, H, W, (size_t)W * H, nf, (sslam_keyline*)dKl, (uint8_t*)dD, (double*)dFn, (int32_t*)dCnt, CAP, NULL)) { fprintf(stderr, "batch: %s\n", sslam_last_error()); return 2; }
                sslam_ctx_synchronize(ctx);
                const double t = now_ms() - t0; if (r && t < best) best = t;
                HIPCHK(hipMemcpy(hCnt, dCnt, nf * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(hKl, dKl, (size_t)nf * CAP * sizeof(sslam_keyline), hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(hD, dD, (size_t)nf * CAP * 32, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(hFn, dFn, (size_t)nf * CAP * 24, hipMemcpyDeviceToHost));
                for (int i = 0; i < nf; ++i) bad += !(hCnt[i] >= 0 && hCnt[i] <= CAP && same(&want[i], hCnt[i], hKl + (size_t)i * CAP, hD + (size_t)i * CAP * 32, hFn + (size_t)i * CAP * 3));
            }
            printf("    %2d frames per call: %7.3f ms per call (fastest of 3), %d of %d frames differ from the oracle\n", nf, best, bad, 4 * nf); fflush(stdout);
            bad_total += bad;
        }
        {   /* part 2 */
            pthread_t th; orbStop = 0; orbCalls = 0;
            pthread_create(&th, NULL, orb_load, NULL);
            struct timespec ts = {0, 200 * 1000 * 1000}; nanosleep(&ts, NULL);
            static Out got; int bad = 0; double sum = 0;
            for (int r = 0; r < 2; ++r) for (int i = 0; i < NF; ++i) {
                const double t0 = now_ms();
                if (sslam_lines_extract(L, img[i], W, H, W, got.kl, got.d, got.fn, CAP, &got.n)) { fprintf(stderr, "extract: %s\n", sslam_last_error()); return 2; }
                sum += now_ms() - t0;
                bad += !same(&want[i], got.n, got.kl, got.d, got.fn);
            }
            orbStop = 1; pthread_join(th, NULL);
            printf("    single frames under a concurrent ORB load (%ld batches of 64 meanwhile): mean %.3f ms, %d of %d differ from the oracle\n", orbCalls, sum / (2 * NF), bad, 2 * NF); fflush(stdout);
            bad_total += bad;
        }
        for (int k = 0; k < nn; ++k) unsetenv(names[k]);
        if (a >= argc) break;
    }
    sslam_lines_destroy(L); sslam_ctx_destroy(ctx);
    printf("%s\n", bad_total ? "DIFFERENT" : "all equal");
    return bad_total ? 1 : 0;
}
