/* Single-frame line extraction through the C ABI alone (no Python: a fresh GPU box spends a second on it, not the minutes of `import torch`): latency per knob setting and
 * parity with the CPU oracle's precomputed answers.
 *     python tools/lat_check_prepare.py            (CPU: the bench's 64 frames -> tools/lat_frames.raw, the oracle's lines for them -> tools/lat_expected.bin)
 *     gcc -O2 -Iinclude tools/lat_check.c -Lstructure-slam-pointline_amd/lib -lsslam_frontend -Wl,-rpath,'$ORIGIN/../structure-slam-pointline_amd/lib' -o tools/lat_check
 *     tools/lat_check <reps> "" "SSLAM_NFA_STREAM=1" "SSLAM_NFA_STREAM=32,SSLAM_NFA_STREAM_LDS=0" ...
 * Each argument after <reps> is one setting: comma-separated NAME=VALUE pairs set for that run and unset afterwards ("" = the default path). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "sslam_frontend.h"

enum { CAP = 512, MAXNF = 64 };
static int W = 640, H = 480, NF = 64, MAXL = 200;      /* LAT_W / LAT_H / LAT_NF / LAT_LINES + LAT_FRAMES / LAT_EXPECTED: another frame set (tools/lat_check_prepare.py <w> <h> <n> <lines>) */
typedef struct { int n; sslam_keyline kl[CAP]; unsigned char d[CAP * 32]; double fn[CAP * 3]; } Out;
static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
static int cmp_d(const void* a, const void* b) { const double x = *(const double*)a, y = *(const double*)b; return x < y ? -1 : x > y; }

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 2;
    if (getenv("LAT_W")) W = atoi(getenv("LAT_W")); if (getenv("LAT_H")) H = atoi(getenv("LAT_H")); if (getenv("LAT_NF")) NF = atoi(getenv("LAT_NF")); if (getenv("LAT_LINES")) MAXL = atoi(getenv("LAT_LINES"));
    if (NF < 8 || NF > MAXNF || MAXL > CAP) return 2;
    const size_t fsz = (size_t)W * H;
    unsigned char* imgs = malloc(fsz * NF);
    static Out want[MAXNF], got;
    FILE* f = fopen(getenv("LAT_FRAMES") ? getenv("LAT_FRAMES") : "tools/lat_frames.raw", "rb");
    if (!f || fread(imgs, 1, fsz * NF, f) != fsz * NF) { fprintf(stderr, "tools/lat_frames.raw missing (tools/lat_check_prepare.py)\n"); return 2; }
    fclose(f);
    f = fopen(getenv("LAT_EXPECTED") ? getenv("LAT_EXPECTED") : "tools/lat_expected.bin", "rb");
    if (!f) { fprintf(stderr, "tools/lat_expected.bin missing\n"); return 2; }
    for (int i = 0; i < NF; ++i) {
        int n = 0;
        if (fread(&n, 4, 1, f) != 1 || n < 0 || n > CAP) return 2;
        want[i].n = n;
        if (fread(want[i].kl, sizeof(sslam_keyline), n, f) != (size_t)n || fread(want[i].d, 32, n, f) != (size_t)n || fread(want[i].fn, 24, n, f) != (size_t)n) return 2;
    }
    fclose(f);
    sslam_ctx* ctx = NULL; sslam_lines* L = NULL;
    if (sslam_ctx_create(0, &ctx) || sslam_lines_create(ctx, MAXL, &L)) { fprintf(stderr, "create: %s\n", sslam_last_error()); return 2; }
    int bad_total = 0;
    for (int a = 2; a < argc || a == 2; ++a) {
        char buf[512]; const char* names[16]; int nn = 0;
        snprintf(buf, sizeof(buf), "%s", a < argc ? argv[a] : "");
        for (char* tok = strtok(buf, ","); tok && nn < 16; tok = strtok(NULL, ",")) { char* eq = strchr(tok, '='); if (!eq) continue; *eq = 0; setenv(tok, eq + 1, 1); names[nn++] = tok; }
        int bad = 0, angle_only = 0;
        for (int i = 0; i < 8; ++i) sslam_lines_extract(L, imgs + fsz * i, W, H, W, got.kl, got.d, got.fn, CAP, &got.n);      /* warm-up (first call plans the workspace) */
        static double t[MAXNF * 16];
        int nt = 0;
        for (int r = 0; r < reps && r < 16; ++r) for (int i = 0; i < NF; ++i) {
            const double t0 = now_ms();
            const int rc = sslam_lines_extract(L, imgs + fsz * i, W, H, W, got.kl, got.d, got.fn, CAP, &got.n);
            t[nt++] = now_ms() - t0;
            if (rc) { fprintf(stderr, "extract: %d %s\n", rc, sslam_last_error()); return 2; }
            int ok = got.n == want[i].n && !memcmp(got.d, want[i].d, 32 * got.n) && !memcmp(got.fn, want[i].fn, 24 * got.n);
            if (ok && memcmp(got.kl, want[i].kl, sizeof(sslam_keyline) * got.n)) {      /* which field? (the suite allows 1 ulp on KeyLine.angle) */
                int other = 0;
                for (int k = 0; k < got.n; ++k) { sslam_keyline x = got.kl[k], y = want[i].kl[k]; x.angle = y.angle = 0; other |= memcmp(&x, &y, sizeof(x)) != 0; }
                if (other) ok = 0; else ++angle_only;
            }
            bad += !ok;
        }
        qsort(t, nt, sizeof(double), cmp_d);
        double sum = 0; for (int i = 0; i < nt; ++i) sum += t[i];
        printf("%-58s p50 %.3f  p90 %.3f  mean %.3f ms   vs oracle: %d of %d differ (%d in KeyLine.angle bits only)\n", a < argc && argv[a][0] ? argv[a] : "(default)", t[nt / 2], t[(nt * 9) / 10], sum / nt, bad, nt, angle_only);
        fflush(stdout);
        bad_total += bad;
        if (getenv("LAT_PROFILE")) {      /* per launch scope: HIP events around each launch on the context's stream (sslam_profile_*), one more pass over the frames */
            sslam_profile_enable(ctx, 1);
            for (int i = 0; i < NF; ++i) sslam_lines_extract(L, imgs + fsz * i, W, H, W, got.kl, got.d, got.fn, CAP, &got.n);
            const char* nm[64]; double ms[64]; int ln[64];
            const int nk = sslam_profile_drain(ctx, nm, ms, ln, 64);
            sslam_profile_enable(ctx, 0);
            double tot = 0; for (int k = 0; k < nk && k < 64; ++k) tot += ms[k];
            printf("    launch scopes, us per frame (sum %.0f):", tot * 1e3 / NF);
            for (int k = 0; k < nk && k < 64; ++k) printf(" %s %.0f (%d)", nm[k], ms[k] * 1e3 / NF, ln[k] / NF);
            printf("\n"); fflush(stdout);
        }
        for (int k = 0; k < nn; ++k) unsetenv(names[k]);
        if (a >= argc) break;
    }
    sslam_lines_destroy(L); sslam_ctx_destroy(ctx);
    return bad_total ? 1 : 0;
}
