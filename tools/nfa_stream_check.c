/* SSLAM_NFA_STREAM against the default path through the C ABI alone (no Python: starts in a second on a fresh box):
 *     gcc -O2 -Iinclude tools/nfa_stream_check.c -Lstructure-slam-pointline_amd/lib -lsslam_frontend -Wl,-rpath,'$ORIGIN/../structure-slam-pointline_amd/lib' -o tools/nfa_stream_check
 *     timeout 60 tools/nfa_stream_check
 * Eight synthetic 640x480 frames; each extracted by the default path (twice: determinism), then with SSLAM_NFA_STREAM=1 under three patience settings (consumers wait /
 * give up at once / give up after 0.2 ms); keylines, LBD bytes and line functions compared byte for byte; then 24 extractions per mode timed.  Exit status 0 = all equal. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "sslam_frontend.h"

enum { W = 640, H = 480, CAP = 256, NIMG = 8 };
static unsigned lcg(unsigned* s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }
static void frame(unsigned seed, unsigned char* img) {
    unsigned s = seed * 2654435761u + 12345u;
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) img[y * W + x] = (unsigned char)(90 + (x * 40) / W + (y * 30) / H);
    for (int k = 0; k < 60; ++k) {      /* filled quadrilaterals with slanted sides and bars: long and short segments at every angle */
        int x0 = (int)(lcg(&s) % W), y0 = (int)(lcg(&s) % H), w = 10 + (int)(lcg(&s) % 180), h = 6 + (int)(lcg(&s) % 140), sh = (int)(lcg(&s) % 120) - 60;
        unsigned char v = (unsigned char)(lcg(&s) % 256);
        for (int y = y0; y < y0 + h && y < H; ++y) { const int off = sh * (y - y0) / h; for (int x = x0 + off; x < x0 + off + w; ++x) if (x >= 0 && x < W) img[y * W + x] = v; }
    }
    for (int i = 0; i < W * H; ++i) { int v = img[i] + (int)(lcg(&s) % 7) - 3; img[i] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v); }
}
static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

typedef struct { int n; sslam_keyline kl[CAP]; unsigned char d[CAP * 32]; double fn[CAP * 3]; } Out;
static int extract(sslam_lines* L, const unsigned char* img, Out* o) {
    memset(o, 0, sizeof(*o));
    const int rc = sslam_lines_extract(L, img, W, H, W, o->kl, o->d, o->fn, CAP, &o->n);
    if (rc) fprintf(stderr, "sslam_lines_extract: %d %s\n", rc, sslam_last_error());
    return rc;
}
static int same(const Out* a, const Out* b) {
    return a->n == b->n && !memcmp(a->kl, b->kl, sizeof(sslam_keyline) * a->n) && !memcmp(a->d, b->d, 32 * a->n) && !memcmp(a->fn, b->fn, 24 * a->n);
}

int main(void) {
    sslam_ctx* ctx = NULL; sslam_lines* L = NULL;
    if (sslam_ctx_create(0, &ctx) || sslam_lines_create(ctx, 200, &L)) { fprintf(stderr, "create: %s\n", sslam_last_error()); return 2; }
    static unsigned char img[NIMG][W * H];
    static Out ref[NIMG], o;
    int bad = 0;
    for (int i = 0; i < NIMG; ++i) frame(1000 + i, img[i]);
    unsetenv("SSLAM_NFA_STREAM");
    for (int i = 0; i < NIMG; ++i) { if (extract(L, img[i], &ref[i])) return 2; if (extract(L, img[i], &o)) return 2; if (!same(&ref[i], &o)) { printf("frame %d: default path not deterministic\n", i); ++bad; } }
    printf("default path: lines per frame"); for (int i = 0; i < NIMG; ++i) printf(" %d", ref[i].n); printf("\n"); fflush(stdout);
    const char* ticks[3] = {NULL, "0", "20000"};
    for (int m = 0; m < 3; ++m) {
        setenv("SSLAM_NFA_STREAM", "1", 1);
        if (ticks[m]) setenv("SSLAM_NFA_STREAM_TICKS", ticks[m], 1); else unsetenv("SSLAM_NFA_STREAM_TICKS");
        int diff = 0;
        for (int rep = 0; rep < 3; ++rep) for (int i = 0; i < NIMG; ++i) { if (extract(L, img[i], &o)) return 2; if (!same(&ref[i], &o)) { ++diff; printf("  frame %d rep %d: %d lines vs %d\n", i, rep, o.n, ref[i].n); } }
        printf("SSLAM_NFA_STREAM=1 ticks=%s: %d of %d extractions differ from the default path\n", ticks[m] ? ticks[m] : "default", diff, 3 * NIMG); fflush(stdout);
        bad += diff;
    }
    for (int m = 0; m < 2; ++m) {
        if (m) { setenv("SSLAM_NFA_STREAM", "1", 1); unsetenv("SSLAM_NFA_STREAM_TICKS"); } else unsetenv("SSLAM_NFA_STREAM");
        for (int i = 0; i < NIMG; ++i) extract(L, img[i], &o);
        double best = 1e9, sum = 0;
        for (int rep = 0; rep < 3; ++rep) for (int i = 0; i < NIMG; ++i) { const double t0 = now_ms(); extract(L, img[i], &o); const double t = now_ms() - t0; sum += t; if (t < best) best = t; }
        printf("%s: %.3f ms per frame (mean of %d), fastest %.3f\n", m ? "SSLAM_NFA_STREAM=1" : "default          ", sum / (3 * NIMG), 3 * NIMG, best); fflush(stdout);
    }
    sslam_lines_destroy(L); sslam_ctx_destroy(ctx);
    printf("%s\n", bad ? "DIFFERENT" : "all equal");
    return bad ? 1 : 0;
}
