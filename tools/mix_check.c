/* Line extraction of frames of many shapes through the C ABI alone (no Python on the GPU box), per knob setting, against the CPU oracle's precomputed lines:
 *     python tools/mix_check_prepare.py     (CPU -> tools/mix_frames.bin)
 *     gcc -O2 -Iinclude tools/mix_check.c -Lstructure-slam-pointline_amd/lib -lsslam_frontend -Wl,-rpath,'$ORIGIN/../structure-slam-pointline_amd/lib' -o tools/mix_check
 *     tools/mix_check <reps> "" "SSLAM_NFA_STREAM=1" "SSLAM_NFA_FUSED=2" ...          (settings as in tools/lat_check.c) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sslam_frontend.h"

enum { MAXF = 64, MAXCAP = 512 };
typedef struct { int w, h, cap, n; unsigned char* img; sslam_keyline* kl; unsigned char* d; double* fn; } Case;

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 2;
    static Case cs[MAXF]; int nc = 0;
    FILE* f = fopen("tools/mix_frames.bin", "rb");
    if (!f || fread(&nc, 4, 1, f) != 1 || nc < 1 || nc > MAXF) { fprintf(stderr, "tools/mix_frames.bin missing (tools/mix_check_prepare.py)\n"); return 2; }
    for (int i = 0; i < nc; ++i) {
        int hd[4]; if (fread(hd, 4, 4, f) != 4) return 2;
        Case* c = &cs[i]; c->w = hd[0]; c->h = hd[1]; c->cap = hd[2]; c->n = hd[3];
        c->img = malloc((size_t)c->w * c->h); c->kl = malloc(sizeof(sslam_keyline) * (c->n + 1)); c->d = malloc(32 * (c->n + 1)); c->fn = malloc(24 * (c->n + 1));
        if (fread(c->img, 1, (size_t)c->w * c->h, f) != (size_t)c->w * c->h || fread(c->kl, sizeof(sslam_keyline), c->n, f) != (size_t)c->n ||
            fread(c->d, 32, c->n, f) != (size_t)c->n || fread(c->fn, 24, c->n, f) != (size_t)c->n) return 2;
    }
    fclose(f);
    sslam_ctx* ctx = NULL;
    if (sslam_ctx_create(0, &ctx)) { fprintf(stderr, "create: %s\n", sslam_last_error()); return 2; }
    static sslam_keyline kl[MAXCAP]; static unsigned char d[MAXCAP * 32]; static double fn[MAXCAP * 3];
    int bad_total = 0;
    for (int a = 2; a < argc || a == 2; ++a) {
        char buf[512]; const char* names[16]; int nn = 0;
        snprintf(buf, sizeof(buf), "%s", a < argc ? argv[a] : "");
        for (char* tok = strtok(buf, ","); tok && nn < 16; tok = strtok(NULL, ",")) { char* eq = strchr(tok, '='); if (!eq) continue; *eq = 0; setenv(tok, eq + 1, 1); names[nn++] = tok; }
        int bad = 0, runs = 0, lines = 0; char which[256] = "";
        sslam_lines* L[3] = {NULL, NULL, NULL}; const int caps[3] = {40, 200, 400};
        for (int k = 0; k < 3; ++k) if (sslam_lines_create(ctx, caps[k], &L[k])) { fprintf(stderr, "lines_create: %s\n", sslam_last_error()); return 2; }
        for (int r = 0; r < reps; ++r) for (int q = 0; q < nc; ++q) {
            const int i = r & 1 ? nc - 1 - q : q;      /* the handles see the sizes in two orders: every call re-plans its workspace for another shape */
            const Case* c = &cs[i]; int n = -1;
            sslam_lines* h = L[c->cap == 40 ? 0 : c->cap == 200 ? 1 : 2];
            const int rc = sslam_lines_extract(h, c->img, c->w, c->h, (size_t)c->w, kl, d, fn, MAXCAP, &n);
            int ok = rc == 0 && n == c->n && !memcmp(d, c->d, 32 * (size_t)n) && !memcmp(fn, c->fn, 24 * (size_t)n);
            if (ok) for (int k = 0; k < n; ++k) { sslam_keyline x = kl[k], y = c->kl[k]; x.angle = y.angle = 0; if (memcmp(&x, &y, sizeof(x))) ok = 0; }      /* (KeyLine.angle: the suite allows 1 ulp) */
            if (!ok) { ++bad; if (strlen(which) < 200) sprintf(which + strlen(which), " %d(%dx%d rc=%d n=%d/%d)", i, c->w, c->h, rc, n, c->n); }
            ++runs; lines += c->n;
        }
        for (int k = 0; k < 3; ++k) sslam_lines_destroy(L[k]);
        printf("%-50s %d of %d extractions differ from the oracle (%d lines compared)%s\n", a < argc && argv[a][0] ? argv[a] : "(default)", bad, runs, lines, which); fflush(stdout);
        bad_total += bad;
        for (int k = 0; k < nn; ++k) unsetenv(names[k]);
        if (a >= argc) break;
    }
    sslam_ctx_destroy(ctx);
    return bad_total ? 1 : 0;
}
