// Order-dependent matchers (an accepted match changes what later queries may take): SearchForInitialization, the
// projection-window family (one-wave form and the LDS-resident speculative form), SearchByBoW.
// Part of match.hip (included there, inside its anonymous namespace: one translation unit).  Not a standalone header.
#pragma once

// ---------------------------------------------------------------- SearchForInitialization
// One wave per frame pair.  The i1 loop is sequential (vMatchedDistance / un-match
// semantics); the candidate scan of each step is wave-parallel.  The
// GetFeaturesInArea order (grid cell x-major, then y, then index — D.4) is carried in
// the reduction key so "first strictly smaller distance wins" is reproduced.
struct SfiArgs {
    const sslam_keypoint* kp1; const uint8_t* d1; const int* n1;
    const sslam_keypoint* kp2; const uint8_t* d2; const int* n2;
    int cap, n1s, n2s;             // n1s/n2s used when n1/n2 pointers are null
    float* prevMatched; int* m12; int* nmatches;
    int* scratch;                  // per pair: matchedDist[cap], m21[cap], cand[cap], key[cap], bin[cap]
    int window; float nnratio; int checkOri;
    float minX, maxX, minY, maxY;
    int ccap;                      // k_search_init_lds: LDS capacity in level-0 candidates / level-0 keypoints of F1 (0: the pair's own counts)
};

// the pair's state in global memory (what a pair falls back to when its level-0 features do not fit the LDS capacity of the batch launch)
__device__ __forceinline__ void sfi_global_body(const SfiArgs& A, int p, int lane, int* __restrict__ hist) {
    const int n1 = A.n1 ? A.n1[p] : A.n1s, n2 = A.n2 ? A.n2[p] : A.n2s;
    const sslam_keypoint* kp1 = A.kp1 + (size_t)p * A.cap;
    const sslam_keypoint* kp2 = A.kp2 + (size_t)p * A.cap;
    const uint8_t* d1 = A.d1 + (size_t)p * A.cap * 32;
    const uint8_t* d2 = A.d2 + (size_t)p * A.cap * 32;
    float* pm = A.prevMatched + (size_t)p * A.cap * 2;
    int* m12 = A.m12 + (size_t)p * A.cap;
    int* matchedDist = A.scratch + (size_t)p * A.cap * 5;
    int* m21 = matchedDist + A.cap;
    int* cand = m21 + A.cap;       // compact list of F2 level-0, in-grid keypoints
    int* ckey = cand + A.cap;      // their GetFeaturesInArea order key
    int* binOf = ckey + A.cap;     // rotation bin of i1 (or -1)

    const float invW = __fdiv_rn((float)GRID_COLS, __fsub_rn(A.maxX, A.minX));
    const float invH = __fdiv_rn((float)GRID_ROWS, __fsub_rn(A.maxY, A.minY));
    for (int i = lane; i < n1; i += 64) { m12[i] = -1; binOf[i] = -1; }
    for (int i = lane; i < n2; i += 64) { matchedDist[i] = 0x7FFFFFFF; m21[i] = -1; }
    if (lane < HISTO_LENGTH) hist[lane] = 0;
    // candidates of level 0 that sit in the 64x48 grid (PosInGrid, src/Frame.cc:462-472)
    int nc = 0;
    for (int j0 = 0; j0 < n2; j0 += 64) {
        int j = j0 + lane;
        bool ok = false; int key = 0;
        if (j < n2) {
            const sslam_keypoint k = kp2[j];
            int px = (int)roundf(__fmul_rn(__fsub_rn(k.x, A.minX), invW));
            int py = (int)roundf(__fmul_rn(__fsub_rn(k.y, A.minY), invH));
            ok = k.octave == 0 && px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS;     // level1 == 0 -> minLevel=maxLevel=0
            key = ((px * GRID_ROWS + py) << 19) | j;
        }
        unsigned long long m = __ballot(ok);
        if (ok) { int o = nc + mbcnt(m); cand[o] = j; ckey[o] = key; }
        nc += __popcll(m);
    }
    __syncthreads();
    int nmatches = 0;
    const float r = (float)A.window;
    for (int i1 = 0; i1 < n1; ++i1) {
        const int level1 = kp1[i1].octave;
        if (level1 > 0) continue;
        const float cx = pm[i1 * 2], cy = pm[i1 * 2 + 1];
        const uint4 q0 = ((const uint4*)(d1 + (size_t)i1 * 32))[0], q1 = ((const uint4*)(d1 + (size_t)i1 * 32))[1];
        unsigned long long b = ~0ull; unsigned s2 = 0x7FFFFFFFu;    // best key (dist<<32|orderkey), second-best distance
        for (int c = lane; c < nc; c += 64) {
            const int j = cand[c];
            const float dx = __fsub_rn(kp2[j].x, cx), dy = __fsub_rn(kp2[j].y, cy);
            if (!(fabsf(dx) < r && fabsf(dy) < r)) continue;
            const uint4* tp = (const uint4*)(d2 + (size_t)j * 32);
            const int dist = hamming256(q0, q1, tp[0], tp[1]);
            if (matchedDist[j] <= dist) continue;
            unsigned long long k = ((unsigned long long)dist << 32) | (unsigned)ckey[c];
            if (k < b) { if (b != ~0ull) s2 = min(s2, (unsigned)(b >> 32)); b = k; }
            else s2 = min(s2, (unsigned)dist);
        }
        const unsigned long long best = wave_min_u64(b);
        if (best == ~0ull) continue;                   // vIndices2 empty, or every candidate suppressed: bestDist stays INT_MAX
        unsigned other = (b == best) ? s2 : min(s2, (unsigned)(b >> 32));
        if (b == ~0ull) other = 0x7FFFFFFFu;
        const unsigned second = wave_min_u32(other);
        const int bestDist = (int)(best >> 32);
        const int bestIdx2 = (int)(best & 0x7FFFF);
        if (bestDist <= TH_LOW && (float)bestDist < __fmul_rn((float)(int)second, A.nnratio)) {
            const int prev = m21[bestIdx2];
            if (prev >= 0) { if (lane == 0) m12[prev] = -1; nmatches--; }
            if (lane == 0) { m12[i1] = bestIdx2; m21[bestIdx2] = i1; matchedDist[bestIdx2] = bestDist; }
            nmatches++;
            if (A.checkOri) {
                float rot = __fsub_rn(kp1[i1].angle, kp2[bestIdx2].angle);
                if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                int bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
                if (bin == HISTO_LENGTH) bin = 0;
                if (lane == 0) { binOf[i1] = bin; hist[bin]++; }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (A.checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;      // ComputeThreeMaxima
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            const int s = hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
        int removed = 0;
        for (int i0 = 0; i0 < n1; i0 += 64) {
            int i = i0 + lane;
            bool rm = false;
            if (i < n1) {
                int bn = binOf[i];
                rm = bn >= 0 && bn != ind1 && bn != ind2 && bn != ind3 && m12[i] >= 0;
                if (rm) m12[i] = -1;
            }
            removed += __popcll(__ballot(rm));
        }
        nmatches -= removed;
    }
    __syncthreads();
    for (int i = lane; i < n1; i += 64) {
        int m = m12[i];
        if (m >= 0) { pm[i * 2] = kp2[m].x; pm[i * 2 + 1] = kp2[m].y; }
    }
    if (lane == 0) A.nmatches[p] = nmatches;
}

__global__ __launch_bounds__(64) void k_search_init(SfiArgs A) {
    __shared__ int hist[HISTO_LENGTH];
    sfi_global_body(A, xcd_mix_frame(blockIdx.x, gridDim.x), threadIdx.x, hist);      // one wave per frame pair: see xcd_mix_frame
}


// Low-latency form for the single call (Tracking::MonocularInitialization hands over ONE frame pair and waits): the same sequential i1
// loop, but everything an iteration touches lives in LDS -- the level-0 keypoints of F1 as a compact list, and per candidate of F2 its
// position, GetFeaturesInArea order key, descriptor, matched distance and owner -- so a step is LDS reads plus one prefetched query
// descriptor instead of a chain of dependent global loads (~1 us each with one wave on the chip).  Semantics identical to k_search_init:
// candidates are compacted in ascending index, so (cell, slot) orders like (cell, index).
// dynamic LDS: per candidate 14 words (x, y, key, index, matchedDist, owner, 8 descriptor words) + per F1 keypoint 2 words (level-0 list, bin)
__global__ __launch_bounds__(64) void k_search_init_lds(SfiArgs A) {
    extern __shared__ __align__(16) unsigned sfi[];
    const int p = A.ccap > 0 ? xcd_mix_frame(blockIdx.x, gridDim.x) : blockIdx.x, lane = threadIdx.x;
    const int n1 = A.n1 ? A.n1[p] : A.n1s, n2 = A.n2 ? A.n2[p] : A.n2s;
    // Round 4: also the batch form.  Only LEVEL-0 features enter the search (a fifth of a frame's keypoints), so the arrays are sized by a
    // capacity in level-0 features chosen by the host (ccap; 0 = the pair's own counts, the single call) -- ~25 KB for 1000-keypoint frames, six
    // pairs per compute unit -- and a pair that exceeds it takes the global-memory body.  The batch kernel before gathered every candidate's
    // position, key, descriptor and matched distance from global memory for every query: 2.45 MB of HBM traffic per pair for 72 KB of inputs.
    const int cc = A.ccap > 0 ? A.ccap : n2, lc = A.ccap > 0 ? A.ccap : n1;
    const sslam_keypoint* kp1 = A.kp1 + (size_t)p * A.cap;
    const sslam_keypoint* kp2 = A.kp2 + (size_t)p * A.cap;
    const uint8_t* d1 = A.d1 + (size_t)p * A.cap * 32;
    const uint8_t* d2 = A.d2 + (size_t)p * A.cap * 32;
    float* pm = A.prevMatched + (size_t)p * A.cap * 2;
    int* m12 = A.m12 + (size_t)p * A.cap;
    uint4* cdesc = (uint4*)sfi;                              // [2 * cc]
    float* cxs = (float*)(cdesc + 2 * (size_t)cc);            // [cc]
    float* cys = cxs + cc;
    int* ckey = (int*)(cys + cc);
    int* cj = ckey + cc;
    int* md = cj + cc;                                       // matched distance per candidate slot
    int* owner = md + cc;                                    // m21 per candidate slot
    int* list1 = owner + cc;                                 // level-0 keypoints of F1, ascending  [lc]
    int* binOf = list1 + lc;                                 // rotation bin per entry of list1, -1  [lc]
    __shared__ int hist[HISTO_LENGTH];
    const float invW = __fdiv_rn((float)GRID_COLS, __fsub_rn(A.maxX, A.minX));
    const float invH = __fdiv_rn((float)GRID_ROWS, __fsub_rn(A.maxY, A.minY));
    if (lane < HISTO_LENGTH) hist[lane] = 0;
    int nl = 0;
    for (int i0 = 0; i0 < n1; i0 += 64) {
        const int i = i0 + lane;
        const bool l0 = i < n1 && kp1[i].octave == 0;
        if (i < n1) m12[i] = -1;
        const unsigned long long m = __ballot(l0);
        if (l0) { const int o = nl + mbcnt(m); if (o < lc) { list1[o] = i; binOf[o] = -1; } }
        nl += __popcll(m);
    }
    int nc = 0;
    for (int j0 = 0; j0 < n2; j0 += 64) {
        const int j = j0 + lane;
        bool ok = false; int cell = 0; float x = 0, y = 0;
        if (j < n2) {
            const sslam_keypoint k = kp2[j];
            const int px = (int)roundf(__fmul_rn(__fsub_rn(k.x, A.minX), invW));
            const int py = (int)roundf(__fmul_rn(__fsub_rn(k.y, A.minY), invH));
            ok = k.octave == 0 && px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS;
            cell = px * GRID_ROWS + py; x = k.x; y = k.y;
        }
        const unsigned long long m = __ballot(ok);
        if (ok && nc + mbcnt(m) < cc) {
            const int o = nc + mbcnt(m);
            cxs[o] = x; cys[o] = y; ckey[o] = (cell << 19) | o; cj[o] = j; md[o] = 0x7FFFFFFF; owner[o] = -1;
            const uint4* tp = (const uint4*)(d2 + (size_t)j * 32);
            cdesc[2 * o] = tp[0]; cdesc[2 * o + 1] = tp[1];
        }
        nc += __popcll(m);
    }
    __syncthreads();
    if (nl > lc || nc > cc) { sfi_global_body(A, p, lane, hist); return; }      // (wave-uniform) more level-0 features than the launch planned for
    int nmatches = 0;
    const float r = (float)A.window;
    uint4 q0n = make_uint4(0, 0, 0, 0), q1n = q0n; float cxn = 0, cyn = 0;
    if (nl > 0) { const int i = list1[0]; q0n = ((const uint4*)(d1 + (size_t)i * 32))[0]; q1n = ((const uint4*)(d1 + (size_t)i * 32))[1]; cxn = pm[i * 2]; cyn = pm[i * 2 + 1]; }
    for (int t = 0; t < nl; ++t) {
        const int i1 = list1[t];
        const uint4 q0 = q0n, q1 = q1n; const float cx = cxn, cy = cyn;
        if (t + 1 < nl) { const int i = list1[t + 1]; q0n = ((const uint4*)(d1 + (size_t)i * 32))[0]; q1n = ((const uint4*)(d1 + (size_t)i * 32))[1]; cxn = pm[i * 2]; cyn = pm[i * 2 + 1]; }
        unsigned long long b = ~0ull; unsigned s2 = 0x7FFFFFFFu;
        for (int c = lane; c < nc; c += 64) {
            const float dx = __fsub_rn(cxs[c], cx), dy = __fsub_rn(cys[c], cy);
            if (!(fabsf(dx) < r && fabsf(dy) < r)) continue;
            const int dist = hamming256(q0, q1, cdesc[2 * c], cdesc[2 * c + 1]);
            if (md[c] <= dist) continue;
            const unsigned long long k = ((unsigned long long)dist << 32) | (unsigned)ckey[c];
            if (k < b) { if (b != ~0ull) s2 = min(s2, (unsigned)(b >> 32)); b = k; }
            else s2 = min(s2, (unsigned)dist);
        }
        const unsigned long long best = wave_min_u64(b);
        if (best == ~0ull) continue;
        unsigned other = (b == best) ? s2 : min(s2, (unsigned)(b >> 32));
        if (b == ~0ull) other = 0x7FFFFFFFu;
        const unsigned second = wave_min_u32(other);
        const int bestDist = (int)(best >> 32);
        const int slot = (int)(best & 0x7FFFF);
        if (bestDist <= TH_LOW && (float)bestDist < __fmul_rn((float)(int)second, A.nnratio)) {
            const int bestIdx2 = cj[slot];
            const int prev = owner[slot];
            if (prev >= 0) { if (lane == 0) m12[prev] = -1; nmatches--; }
            if (lane == 0) { m12[i1] = bestIdx2; owner[slot] = i1; md[slot] = bestDist; }
            nmatches++;
            if (A.checkOri) {
                float rot = __fsub_rn(kp1[i1].angle, kp2[bestIdx2].angle);
                if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                int bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
                if (bin == HISTO_LENGTH) bin = 0;
                if (lane == 0) { binOf[t] = bin; hist[bin]++; }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (A.checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;      // ComputeThreeMaxima
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            const int sh = hist[i];
            if (sh > max1) { max3 = max2; max2 = max1; max1 = sh; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sh > max2) { max3 = max2; max2 = sh; ind3 = ind2; ind2 = i; }
            else if (sh > max3) { max3 = sh; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
        int removed = 0;
        for (int t0 = 0; t0 < nl; t0 += 64) {
            const int t = t0 + lane;
            bool rm = false;
            if (t < nl) {
                const int bn = binOf[t], i = list1[t];
                rm = bn >= 0 && bn != ind1 && bn != ind2 && bn != ind3 && m12[i] >= 0;
                if (rm) m12[i] = -1;
            }
            removed += __popcll(__ballot(rm));
        }
        nmatches -= removed;
    }
    __syncthreads();
    for (int i = lane; i < n1; i += 64) {
        const int m = m12[i];
        if (m >= 0) { pm[i * 2] = kp2[m].x; pm[i * 2 + 1] = kp2[m].y; }
    }
    if (lane == 0) A.nmatches[p] = nmatches;
}

// Speculative form of the single call (round 3): sixteen waves evaluate sixteen consecutive level-0 keypoints of F1 against the state at the
// start of a round; wave 0 then commits them.  A match only ever LOWERS the matched distance of its candidate slot (a candidate is eligible
// while matchedDist > dist), so an earlier keypoint of the round can change a later one's result only through that slot being the later
// one's best or second-best candidate -- the rule of k_search_proj_lds.  Every accepting wave stamps its slot with (round, wave); if no
// wave finds an earlier stamp of the round on its best / second, all sixteen commit at once (un-matching the slots' previous owners, which
// are then keypoints of earlier rounds: distinct slots, distinct owners); otherwise the round is replayed serially with re-evaluation.
// Same LDS layout as k_search_init_lds plus the stamps; same results (tests/test_match_gpu.py compares all three forms with the oracle).
constexpr int SFI_WAVES = 16;
struct SfiLds { uint4* cdesc; float* cxs; float* cys; int* ckey; int* cj; int* md; int* owner; int* stamp; };
__device__ __forceinline__ void sfi_scan(const SfiLds& S, int nc, float cx, float cy, float r, const uint4& q0, const uint4& q1, int lane,
                                         unsigned long long& best, unsigned long long& second) {
    unsigned long long b = ~0ull, s2 = ~0ull;
    for (int c = lane; c < nc; c += 64) {
        const float dx = __fsub_rn(S.cxs[c], cx), dy = __fsub_rn(S.cys[c], cy);
        if (!(fabsf(dx) < r && fabsf(dy) < r)) continue;
        const int dist = hamming256(q0, q1, S.cdesc[2 * c], S.cdesc[2 * c + 1]);
        if (S.md[c] <= dist) continue;
        const unsigned long long k = ((unsigned long long)dist << 32) | (unsigned)S.ckey[c];
        if (k < b) { s2 = b; b = k; } else if (k < s2) s2 = k;
    }
    best = wave_min_u64(b);
    second = wave_min_u64(b == best ? s2 : b);
}
__global__ __launch_bounds__(SFI_WAVES * 64) void k_search_init_spec(SfiArgs A) {
    extern __shared__ __align__(16) unsigned sfi[];
    constexpr int NT = SFI_WAVES * 64;
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n1 = A.n1 ? A.n1[p] : A.n1s, n2 = A.n2 ? A.n2[p] : A.n2s;
    const sslam_keypoint* kp1 = A.kp1 + (size_t)p * A.cap;
    const sslam_keypoint* kp2 = A.kp2 + (size_t)p * A.cap;
    const uint8_t* d1 = A.d1 + (size_t)p * A.cap * 32;
    const uint8_t* d2 = A.d2 + (size_t)p * A.cap * 32;
    float* pm = A.prevMatched + (size_t)p * A.cap * 2;
    int* m12 = A.m12 + (size_t)p * A.cap;
    SfiLds S;
    S.cdesc = (uint4*)sfi;
    S.cxs = (float*)(S.cdesc + 2 * (size_t)n2); S.cys = S.cxs + n2;
    S.ckey = (int*)(S.cys + n2); S.cj = S.ckey + n2; S.md = S.cj + n2; S.owner = S.md + n2; S.stamp = S.owner + n2;
    int* list1 = S.stamp + n2;                               // level-0 keypoints of F1, ascending  [n1]
    int* binOf = list1 + n1;                                 // rotation bin per F1 keypoint, -1   [n1]
    __shared__ int hist[HISTO_LENGTH];
    __shared__ int sh_nl, sh_nc, sh_nm;
    __shared__ int rAcc[SFI_WAVES], rBest[SFI_WAVES], rSecond[SFI_WAVES], rDist[SFI_WAVES], rBin[SFI_WAVES];
    const float invW = __fdiv_rn((float)GRID_COLS, __fsub_rn(A.maxX, A.minX));
    const float invH = __fdiv_rn((float)GRID_ROWS, __fsub_rn(A.maxY, A.minY));
    if (tid < HISTO_LENGTH) hist[tid] = 0;
    if (tid == 0) sh_nm = 0;
    for (int i = tid; i < n1; i += NT) { m12[i] = -1; binOf[i] = -1; }
    if (wave == 0) {                                         // the two compact lists, in ascending index (one wave: ordered compaction by ballots)
        int nl = 0;
        for (int i0 = 0; i0 < n1; i0 += 64) {
            const int i = i0 + lane;
            const bool l0 = i < n1 && kp1[i].octave == 0;
            const unsigned long long m = __ballot(l0);
            if (l0) list1[nl + mbcnt(m)] = i;
            nl += __popcll(m);
        }
        int nc = 0;
        for (int j0 = 0; j0 < n2; j0 += 64) {
            const int j = j0 + lane;
            bool ok = false; int cell = 0; float x = 0, y = 0;
            if (j < n2) {
                const sslam_keypoint k = kp2[j];
                const int px = (int)roundf(__fmul_rn(__fsub_rn(k.x, A.minX), invW));
                const int py = (int)roundf(__fmul_rn(__fsub_rn(k.y, A.minY), invH));
                ok = k.octave == 0 && px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS;
                cell = px * GRID_ROWS + py; x = k.x; y = k.y;
            }
            const unsigned long long m = __ballot(ok);
            if (ok) {
                const int o = nc + mbcnt(m);
                S.cxs[o] = x; S.cys[o] = y; S.ckey[o] = (cell << 19) | o; S.cj[o] = j; S.md[o] = 0x7FFFFFFF; S.owner[o] = -1; S.stamp[o] = 0;
            }
            nc += __popcll(m);
        }
        if (lane == 0) { sh_nl = nl; sh_nc = nc; }
    }
    __syncthreads();
    const int nl = sh_nl, nc = sh_nc;
    for (int c = tid; c < nc; c += NT) {                    // descriptors of the candidates, by all waves
        const uint4* tp = (const uint4*)(d2 + (size_t)S.cj[c] * 32);
        S.cdesc[2 * c] = tp[0]; S.cdesc[2 * c + 1] = tp[1];
    }
    __syncthreads();
    const float r = (float)A.window;
    auto decide = [&](int i1, unsigned long long best, unsigned long long second, int& acc, int& slot, int& slot2, int& dist, int& bin) {
        acc = 0; slot = -1; slot2 = -2; dist = 0; bin = -1;
        if (best == ~0ull) return;                           // vIndices2 empty, or every candidate suppressed
        slot = (int)(best & 0x7FFFF);
        if (second != ~0ull) slot2 = (int)(second & 0x7FFFF);
        const int bestDist = (int)(best >> 32);
        const unsigned sec = second == ~0ull ? 0x7FFFFFFFu : (unsigned)(second >> 32);
        if (bestDist <= TH_LOW && (float)bestDist < __fmul_rn((float)(int)sec, A.nnratio)) {
            acc = 1; dist = bestDist;
            if (A.checkOri) {
                float rot = __fsub_rn(kp1[i1].angle, kp2[S.cj[slot]].angle);
                if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
                if (bin == HISTO_LENGTH) bin = 0;
            }
        }
    };
    for (int base = 0; base < nl; base += SFI_WAVES) {
        const int t = base + wave;
        int acc = 0, slot = -1, slot2 = -2, dist = 0, bin = -1;
        if (t < nl) {
            const int i1 = list1[t];
            const uint4 q0 = ((const uint4*)(d1 + (size_t)i1 * 32))[0], q1 = ((const uint4*)(d1 + (size_t)i1 * 32))[1];
            unsigned long long best, second;
            sfi_scan(S, nc, pm[i1 * 2], pm[i1 * 2 + 1], r, q0, q1, lane, best, second);
            decide(i1, best, second, acc, slot, slot2, dist, bin);
        }
        if (lane == 0) { rAcc[wave] = acc; rBest[wave] = slot; rSecond[wave] = slot2; rDist[wave] = dist; rBin[wave] = bin; }
        __syncthreads();
        if (wave == 0) {
            const int cnt = min(SFI_WAVES, nl - base);
            const int myAcc = lane < cnt ? rAcc[lane] : 0, myB = lane < cnt ? rBest[lane] : -1, myS = lane < cnt ? rSecond[lane] : -2;
            const int myDist = lane < cnt ? rDist[lane] : 0, myBin = lane < cnt ? rBin[lane] : -1;
            const int roundTag = base / SFI_WAVES + 1;
            if (lane < cnt && myAcc) atomicMax(&S.stamp[myB], (roundTag << 5) | (31 - lane));
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            bool clash = false;
            if (lane < cnt) {
                if (myB >= 0) { const int st = S.stamp[myB]; clash |= (st >> 5) == roundTag && 31 - (st & 31) < lane; }
                if (myS >= 0) { const int st = S.stamp[myS]; clash |= (st >> 5) == roundTag && 31 - (st & 31) < lane; }
            }
            int delta = 0;
            if (!__ballot(clash)) {
                bool unmatch = false;
                if (lane < cnt && myAcc) {
                    const int i1 = list1[base + lane];
                    const int prev = S.owner[myB];
                    if (prev >= 0) { m12[prev] = -1; unmatch = true; }
                    m12[i1] = S.cj[myB]; S.owner[myB] = i1; S.md[myB] = myDist;
                    if (A.checkOri) { binOf[i1] = myBin; atomicAdd(&hist[myBin], 1); }
                }
                delta = __popcll(__ballot(lane < cnt && myAcc)) - __popcll(__ballot(unmatch));
            } else {
                // serial replay of the round: a keypoint is re-evaluated when an earlier one of the round took its best or second-best slot
                int myTaken = -3, nTaken = 0;
                for (int w = 0; w < cnt; ++w) {
                    int a = __builtin_amdgcn_readlane(myAcc, w), b = __builtin_amdgcn_readlane(myB, w), ds = __builtin_amdgcn_readlane(myDist, w), bn = __builtin_amdgcn_readlane(myBin, w);
                    const int s2 = __builtin_amdgcn_readlane(myS, w);
                    const int i1 = list1[base + w];
                    if (nTaken > 0 && __ballot(lane < nTaken && (myTaken == b || myTaken == s2))) {
                        const uint4 q0 = ((const uint4*)(d1 + (size_t)i1 * 32))[0], q1 = ((const uint4*)(d1 + (size_t)i1 * 32))[1];
                        unsigned long long best, second; int sl2;
                        sfi_scan(S, nc, pm[i1 * 2], pm[i1 * 2 + 1], r, q0, q1, lane, best, second);
                        decide(i1, best, second, a, b, sl2, ds, bn);
                    }
                    if (!a) continue;
                    const int prev = S.owner[b];
                    if (prev >= 0) { if (lane == 0) m12[prev] = -1; --delta; }
                    if (lane == 0) { m12[i1] = S.cj[b]; S.owner[b] = i1; S.md[b] = ds; if (A.checkOri) { binOf[i1] = bn; hist[bn]++; } }
                    ++delta;
                    if (lane == nTaken) myTaken = b;
                    ++nTaken;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                }
            }
            if (lane == 0) sh_nm += delta;
        }
        __syncthreads();
    }
    int nmatches = sh_nm;
    __syncthreads();
    if (A.checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;      // ComputeThreeMaxima
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            const int sh = hist[i];
            if (sh > max1) { max3 = max2; max2 = max1; max1 = sh; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sh > max2) { max3 = max2; max2 = sh; ind3 = ind2; ind2 = i; }
            else if (sh > max3) { max3 = sh; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
        if (wave == 0) {
            int removed = 0;
            for (int i0 = 0; i0 < n1; i0 += 64) {
                const int i = i0 + lane;
                bool rm = false;
                if (i < n1) {
                    const int bn = binOf[i];
                    rm = bn >= 0 && bn != ind1 && bn != ind2 && bn != ind3 && m12[i] >= 0;
                    if (rm) m12[i] = -1;
                }
                removed += __popcll(__ballot(rm));
            }
            nmatches -= removed;
        }
    }
    __syncthreads();
    for (int i = tid; i < n1; i += NT) {
        const int m = m12[i];
        if (m >= 0) { pm[i * 2] = kp2[m].x; pm[i * 2 + 1] = kp2[m].y; }
    }
    if (tid == 0) A.nmatches[p] = nmatches;
}

// ---------------------------------------------------------------- projection-window matchers
// ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th)  src/ORBmatcher.cc:45-129   (kind 0, mode 0)
// ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)  src/ORBmatcher.cc:1331-1473 (kind 0, mode 1)
// LSDmatcher::SearchByProjection(Frame&, vector<MapLine*>&, th) / (Frame&, const Frame&, ...) src/LSDmatcher.cpp:185-255,22-141 (kind 1, mode 0)
// One wave per problem.  Queries are consumed sequentially (an accepted match occupies its keypoint for
// the later queries); the candidate scan of each query is wave-parallel, with the reference's candidate
// order (GetFeaturesInArea cell order / GetLinesInArea index order) carried in the reduction key.
struct ProjArgs {
    int kind, mode;
    const void* feats; const uint8_t* desc; int n;
    float minX, maxX, minY, maxY;
    const float* uright; const uint8_t* occIn;
    const sslam_proj_query* q; const uint8_t* qdesc; int nq;
    float nnratio; int thDist, checkOri;
    int* assigned; int* nmatches;
    int* scratch;          // occ[n], key[n], qbin[nq], qidx[nq]
    long long* stats;      // SSLAM_PROJ_STATS (development aid, two-kernel form): commit steps, re-scans, cycles in re-scans, total cycles
};

__global__ __launch_bounds__(64) void k_search_proj(ProjArgs A) {
    const int lane = threadIdx.x;
    const int n = A.n, nq = A.nq;
    int* occ = A.scratch; int* key = occ + n; int* qbin = key + n; int* qidx = qbin + nq;
    const sslam_keypoint* kps = (const sslam_keypoint*)A.feats;
    const sslam_keyline* kls = (const sslam_keyline*)A.feats;
    __shared__ int hist[HISTO_LENGTH];
    if (lane < HISTO_LENGTH) hist[lane] = 0;
    const float invW = __fdiv_rn((float)GRID_COLS, __fsub_rn(A.maxX, A.minX));
    const float invH = __fdiv_rn((float)GRID_ROWS, __fsub_rn(A.maxY, A.minY));
    for (int i = lane; i < n; i += 64) {
        occ[i] = A.occIn ? (int)A.occIn[i] : 0;
        A.assigned[i] = -1;
        int k = i;                                  // lines: GetLinesInArea scans in index order
        if (A.kind == 0) {
            const int px = (int)roundf(__fmul_rn(__fsub_rn(kps[i].x, A.minX), invW));
            const int py = (int)roundf(__fmul_rn(__fsub_rn(kps[i].y, A.minY), invH));
            k = (px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS) ? (((px * GRID_ROWS + py) << 19) | i) : -1;
        }
        key[i] = k;
    }
    for (int i = lane; i < nq; i += 64) qbin[i] = -1;
    __syncthreads();
    int nmatches = 0;
    for (int iq = 0; iq < nq; ++iq) {
        const sslam_proj_query Q = A.q[iq];
        if (!Q.valid) continue;
        const uint4 q0 = ((const uint4*)(A.qdesc + (size_t)iq * 32))[0], q1 = ((const uint4*)(A.qdesc + (size_t)iq * 32))[1];
        unsigned long long b = ~0ull, s = ~0ull;
        bool any = false;
        for (int i = lane; i < n; i += 64) {
            const int k = key[i];
            if (k < 0) continue;
            int oct;
            if (A.kind == 0) {
                const sslam_keypoint kp = kps[i];
                oct = kp.octave;
                if (Q.min_level > 0 || Q.max_level >= 0) {
                    if (oct < Q.min_level) continue;
                    if (Q.max_level >= 0 && oct > Q.max_level) continue;
                }
                const float dx = __fsub_rn(kp.x, Q.u), dy = __fsub_rn(kp.y, Q.v);
                if (!(fabsf(dx) < Q.radius && fabsf(dy) < Q.radius)) continue;
            } else {
                const sslam_keyline kl = kls[i];
                oct = kl.octave;
                const double mxp = 0.5 * (double)__fadd_rn(Q.u, Q.u2) - (double)kl.pt_x, myp = 0.5 * (double)__fadd_rn(Q.v, Q.v2) - (double)kl.pt_y;
                const float distance = (float)(mxp * mxp + myp * myp);
                if (distance > __fmul_rn(Q.radius, Q.radius)) continue;
                const float slope = __fsub_rn(__fdiv_rn(__fsub_rn(Q.v, Q.v2), __fsub_rn(Q.u, Q.u2)), kl.angle);
                if ((double)slope > (double)Q.radius * 0.01) continue;
                if (Q.min_level > 0 || Q.max_level > 0) {
                    if (oct < Q.min_level) continue;
                    if (Q.max_level >= 0 && oct > Q.max_level) continue;
                }
            }
            any = true;                              // vIndices non-empty
            if (occ[i]) continue;
            if (A.kind == 0 && A.uright) {
                const float ur = A.uright[i];
                if (ur > 0 && fabsf(__fsub_rn(Q.ur, ur)) > Q.radius) continue;
            }
            const uint4* tp = (const uint4*)(A.desc + (size_t)i * 32);
            const unsigned long long kk = ((unsigned long long)hamming256(q0, q1, tp[0], tp[1]) << 32) | (unsigned)k;
            if (kk < b) { s = b; b = kk; } else if (kk < s) s = kk;
        }
        if (!__ballot(any)) continue;
        const unsigned long long best = wave_min_u64(b);
        const unsigned long long second = wave_min_u64(b == best ? s : b);
        int bestDist = 256, bestLevel = -1, bestIdx = -1, bestDist2 = 256, bestLevel2 = -1;
        if (best != ~0ull && (int)(best >> 32) < 256) {
            bestDist = (int)(best >> 32); bestIdx = (int)(best & 0x7FFFF);
            bestLevel = A.kind == 0 ? kps[bestIdx].octave : kls[bestIdx].octave;
        }
        if (A.mode == 0 && second != ~0ull && (int)(second >> 32) < 256) {
            bestDist2 = (int)(second >> 32);
            const int i2 = (int)(second & 0x7FFFF);
            bestLevel2 = A.kind == 0 ? kps[i2].octave : kls[i2].octave;
        }
        if (bestDist <= A.thDist && bestIdx >= 0) {
            if (A.mode == 0 && bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(A.nnratio, (float)bestDist2)) continue;
            ++nmatches;
            if (lane == 0) { A.assigned[bestIdx] = iq; occ[bestIdx] = Q.obs_positive ? 1 : 0; }
            if (A.mode == 1 && A.checkOri) {
                float rot = __fsub_rn(Q.angle, kps[bestIdx].angle);
                if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                int bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
                if (bin == HISTO_LENGTH) bin = 0;
                if (lane == 0) { qbin[iq] = bin; qidx[iq] = bestIdx; hist[bin]++; }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (A.mode == 1 && A.checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            const int c = hist[i];
            if (c > max1) { max3 = max2; max2 = max1; max1 = c; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (c > max2) { max3 = max2; max2 = c; ind3 = ind2; ind2 = i; }
            else if (c > max3) { max3 = c; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
        int removed = 0;
        for (int i0 = 0; i0 < nq; i0 += 64) {
            const int i = i0 + lane;
            bool rm = false;
            if (i < nq) { const int bn = qbin[i]; rm = bn >= 0 && bn != ind1 && bn != ind2 && bn != ind3; if (rm) A.assigned[qidx[i]] = -2; }      // matched, then removed by the rotation check: the reference NULLs the pointer (src/ORBmatcher.cc:1465)
            removed += __popcll(__ballot(rm));
        }
        nmatches -= removed;
    }
    if (lane == 0) *A.nmatches = nmatches;
}


// ------------------------------------------------------------------ projection matchers, two-kernel form (round 3: the single call)
// The LDS-resident kernel below keeps one CU busy: sixteen waves share four SIMDs, so a round of sixteen candidate scans is bound by that
// CU's issue rate (5.5 k cycles per round measured, 63 rounds per 1000 queries), and its commit falls back to a serial replay whenever two
// queries of a round meet on a feature (62 % of the rounds on neighbouring keypoints): 0.43-0.51 ms per call against 0.26-0.44 ms on one
// CPU core.  The candidate SET of a query is static -- window, level range, stereo gate and the initial occupancy do not depend on earlier
// queries; only "taken by an earlier accepted query" does.  So:
//   k_proj_topk    one wave per query over the whole chip: scan every feature once, keep the PROJ_K smallest keys (distance, then the
//                  reference's candidate order -- GetFeaturesInArea cell order / GetLinesInArea index order --, level in the low bits) and
//                  the number of eligible candidates;
//   k_proj_commit  one wave walks the queries in order, 64 at a time: every pending query picks the first (and second) entry of its list
//                  that is still free, accepting queries stamp their feature, and the longest prefix of queries on whose picks no EARLIER
//                  pending query has a stamp commits at once (their decisions cannot influence each other); the first query with a stamp
//                  in front of it simply picks again in the next step, now as the first pending one.  A query whose list runs out of
//                  free entries although it had more candidates than PROJ_K is re-scanned against the live occupancy (rare).
// The result equals the sequential loop: a query's pick is "smallest key among its candidates that are free when its turn comes", and a
// prefix commits only when that set of free candidates is already final for every query in it.
// Measured (1000 queries against 1000 keypoints): 0.14-0.22 ms per call, CPU oracle 0.26-0.46 ms (profiles/r03_matchers.txt).
#ifndef SSLAM_PROJ_K
#define SSLAM_PROJ_K 8
#endif
constexpr int PROJ_K = SSLAM_PROJ_K;      // list length: with 4, 3 % of the queries of a dense frame ran out of free entries and paid a re-scan (11-13 k cycles each, two thirds of the commit)
struct ProjTopArgs { ProjArgs A; unsigned long long* top; int* cnt; };

// the candidate test of one (query, feature) pair, features in global memory; returns false when the feature is no candidate.
// key = dist << 35 | order << 4 | level, order = (cell << 19 | index) for keypoints, index for lines
template <class OccFn>
__device__ __forceinline__ bool proj_candidate(const ProjArgs& A, const sslam_proj_query& Q, const uint4& q0, const uint4& q1, int i, float invW, float invH, OccFn occupied,
                                               unsigned long long& key) {
    const sslam_keypoint* kps = (const sslam_keypoint*)A.feats;
    const sslam_keyline* kls = (const sslam_keyline*)A.feats;
    int oct; unsigned order = (unsigned)i;
    if (A.kind == 0) {
        const sslam_keypoint kp = kps[i];
        oct = kp.octave;
        if (Q.min_level > 0 || Q.max_level >= 0) {
            if (oct < Q.min_level) return false;
            if (Q.max_level >= 0 && oct > Q.max_level) return false;
        }
        const float dx = __fsub_rn(kp.x, Q.u), dy = __fsub_rn(kp.y, Q.v);
        if (!(fabsf(dx) < Q.radius && fabsf(dy) < Q.radius)) return false;
        const int px = (int)roundf(__fmul_rn(__fsub_rn(kp.x, A.minX), invW));
        const int py = (int)roundf(__fmul_rn(__fsub_rn(kp.y, A.minY), invH));
        if (!(px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS)) return false;      // not in the grid: GetFeaturesInArea never returns it
        order = ((unsigned)(px * GRID_ROWS + py) << 19) | (unsigned)i;
    } else {
        const sslam_keyline kl = kls[i];
        oct = kl.octave;
        const double mxp = 0.5 * (double)__fadd_rn(Q.u, Q.u2) - (double)kl.pt_x, myp = 0.5 * (double)__fadd_rn(Q.v, Q.v2) - (double)kl.pt_y;
        const float distance = (float)(mxp * mxp + myp * myp);
        if (distance > __fmul_rn(Q.radius, Q.radius)) return false;
        const float slope = __fsub_rn(__fdiv_rn(__fsub_rn(Q.v, Q.v2), __fsub_rn(Q.u, Q.u2)), kl.angle);
        if ((double)slope > (double)Q.radius * 0.01) return false;
        if (Q.min_level > 0 || Q.max_level > 0) {
            if (oct < Q.min_level) return false;
            if (Q.max_level >= 0 && oct > Q.max_level) return false;
        }
    }
    if (occupied(i)) return false;
    if (A.kind == 0 && A.uright) {
        const float ur = A.uright[i];
        if (ur > 0 && fabsf(__fsub_rn(Q.ur, ur)) > Q.radius) return false;
    }
    const uint4* tp = (const uint4*)(A.desc + (size_t)i * 32);
    key = ((unsigned long long)hamming256(q0, q1, tp[0], tp[1]) << 35) | ((unsigned long long)order << 4) | (unsigned)(oct & 15);
    return true;
}
__device__ __forceinline__ int proj_key_feature(unsigned long long key) { return (int)((key >> 4) & 0x7FFFFu); }

__global__ __launch_bounds__(256) void k_proj_topk(ProjTopArgs T) {
    const ProjArgs& A = T.A;
    const int lane = threadIdx.x & 63, iq = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (iq >= A.nq) return;
    const sslam_proj_query Q = A.q[iq];
    unsigned long long t[PROJ_K];
#pragma unroll
    for (int k = 0; k < PROJ_K; ++k) t[k] = ~0ull;
    int cnt = 0;
    if (Q.valid) {
        const float invW = __fdiv_rn((float)GRID_COLS, __fsub_rn(A.maxX, A.minX));
        const float invH = __fdiv_rn((float)GRID_ROWS, __fsub_rn(A.maxY, A.minY));
        const uint4 q0 = ((const uint4*)(A.qdesc + (size_t)iq * 32))[0], q1 = ((const uint4*)(A.qdesc + (size_t)iq * 32))[1];
        const uint8_t* occIn = A.occIn;
        for (int i = lane; i < A.n; i += 64) {
            unsigned long long kk;
            if (!proj_candidate(A, Q, q0, q1, i, invW, invH, [&](int f) { return occIn && occIn[f]; }, kk)) continue;
            ++cnt;
            if (kk < t[PROJ_K - 1]) {                        // the lane's own PROJ_K smallest, ascending (one insertion pass)
                t[PROJ_K - 1] = kk;
#pragma unroll
                for (int k = PROJ_K - 1; k > 0; --k) if (t[k] < t[k - 1]) { const unsigned long long x = t[k - 1]; t[k - 1] = t[k]; t[k] = x; }
            }
        }
    }
    cnt = wave_sum(cnt);
#pragma unroll
    for (int k = 0; k < PROJ_K; ++k) {                       // merge: the wave's smallest, K times (keys are unique: they carry the feature index)
        const unsigned long long m = wave_min_u64(t[0]);
        if (lane == 0) T.top[(size_t)iq * PROJ_K + k] = m;
        if (t[0] == m && m != ~0ull) {
#pragma unroll
            for (int j = 0; j + 1 < PROJ_K; ++j) t[j] = t[j + 1];
            t[PROJ_K - 1] = ~0ull;
        }
    }
    if (lane == 0) T.cnt[iq] = cnt;
}

// accept / reject on (best, second) keys of the layout above: thresholds, same-level ratio test (mode 0), rotation bin (mode 1)
__device__ __forceinline__ void proj_decide_keys(const ProjArgs& A, float qAngle, unsigned long long b, unsigned long long s2, int& acc, int& bin) {
    acc = 0; bin = -1;
    if (b == ~0ull) return;
    const int bestDist = (int)(b >> 35), bestLevel = (int)(b & 15);
    int bestDist2 = 256, bestLevel2 = -1;
    if (A.mode == 0 && s2 != ~0ull) { bestDist2 = (int)(s2 >> 35); bestLevel2 = (int)(s2 & 15); }
    if (bestDist > A.thDist || bestDist >= 256) return;
    if (A.mode == 0 && bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(A.nnratio, (float)bestDist2)) return;
    acc = 1;
    if (A.mode == 1 && A.checkOri) {
        float rot = __fsub_rn(qAngle, ((const sslam_keypoint*)A.feats)[proj_key_feature(b)].angle);
        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
        bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
        if (bin == HISTO_LENGTH) bin = 0;
    }
}

// the same test on a copy of the frame's features in LDS (k_proj_commit's re-scan); every field first, the tests afterwards: one LDS round
// trip per feature instead of one per early exit
struct ProjFeatLds { const uint4* desc; const float* x; const float* y; const float* ang; const float* ur; const int* oct; const int* order; };
__device__ __forceinline__ bool proj_candidate_lds(const ProjArgs& A, const ProjFeatLds& F, const sslam_proj_query& Q, const uint4& q0, const uint4& q1, int i, const int* __restrict__ occ,
                                                   unsigned long long& key) {
    const int order = F.order[i], oct = F.oct[i], occupied = occ[i];
    const float fx = F.x[i], fy = F.y[i], fang = F.ang[i], fur = F.ur[i];
    bool ok = order >= 0 && !occupied;                        // (order < 0: keypoint outside the grid)
    if (A.kind == 0) {
        if (Q.min_level > 0 || Q.max_level >= 0) ok = ok && oct >= Q.min_level && !(Q.max_level >= 0 && oct > Q.max_level);
        const float dx = __fsub_rn(fx, Q.u), dy = __fsub_rn(fy, Q.v);
        ok = ok && fabsf(dx) < Q.radius && fabsf(dy) < Q.radius;
        if (A.uright) ok = ok && !(fur > 0 && fabsf(__fsub_rn(Q.ur, fur)) > Q.radius);
    } else {
        const double mxp = 0.5 * (double)__fadd_rn(Q.u, Q.u2) - (double)fx, myp = 0.5 * (double)__fadd_rn(Q.v, Q.v2) - (double)fy;
        const float distance = (float)(mxp * mxp + myp * myp);
        ok = ok && !(distance > __fmul_rn(Q.radius, Q.radius));
        const float slope = __fsub_rn(__fdiv_rn(__fsub_rn(Q.v, Q.v2), __fsub_rn(Q.u, Q.u2)), fang);
        ok = ok && !((double)slope > (double)Q.radius * 0.01);
        if (Q.min_level > 0 || Q.max_level > 0) ok = ok && oct >= Q.min_level && !(Q.max_level >= 0 && oct > Q.max_level);
    }
    if (!ok) return false;
    key = ((unsigned long long)hamming256(q0, q1, F.desc[2 * i], F.desc[2 * i + 1]) << 35) | ((unsigned long long)(unsigned)order << 4) | (unsigned)(oct & 15);
    return true;
}

// dynamic LDS: occ[n], stamp[n] (+ when the frame fits, PROJ_MAXN features: descriptors, positions, angle, right coordinate, level, order key)
__global__ __launch_bounds__(64) void k_proj_commit(ProjTopArgs T, int featsInLds) {
    extern __shared__ __align__(16) int pc[];
    const ProjArgs& A = T.A;
    const int lane = threadIdx.x, n = A.n, nq = A.nq;
    ProjFeatLds F;
    uint4* fdesc = (uint4*)pc;                                  // [2n] when featsInLds
    int* rest = featsInLds ? (int*)(fdesc + 2 * (size_t)n) : pc;
    int* occ = rest; int* stamp = rest + n;
    float* fx = (float*)(stamp + n); float* fy = fx + n; float* fang = fy + n; float* fur = fang + n; int* foct = (int*)(fur + n); int* ford = foct + n;
    F.desc = fdesc; F.x = fx; F.y = fy; F.ang = fang; F.ur = fur; F.oct = foct; F.order = ford;
    int* qbin = A.scratch + 2 * n; int* qidx = qbin + nq;          // the scratch layout of the other forms
    __shared__ int hist[HISTO_LENGTH];
    if (lane < HISTO_LENGTH) hist[lane] = 0;
    const float invW = __fdiv_rn((float)GRID_COLS, __fsub_rn(A.maxX, A.minX));
    const float invH = __fdiv_rn((float)GRID_ROWS, __fsub_rn(A.maxY, A.minY));
    for (int i = lane; i < n; i += 64) { occ[i] = A.occIn ? (int)A.occIn[i] : 0; stamp[i] = 0; A.assigned[i] = -1; }
    if (featsInLds) for (int i = lane; i < n; i += 64) {
        int order = i;
        if (A.kind == 0) {
            const sslam_keypoint kp = ((const sslam_keypoint*)A.feats)[i];
            fx[i] = kp.x; fy[i] = kp.y; fang[i] = kp.angle; foct[i] = kp.octave;
            const int px = (int)roundf(__fmul_rn(__fsub_rn(kp.x, A.minX), invW)), py = (int)roundf(__fmul_rn(__fsub_rn(kp.y, A.minY), invH));
            order = (px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS) ? (((px * GRID_ROWS + py) << 19) | i) : -1;
        } else {
            const sslam_keyline kl = ((const sslam_keyline*)A.feats)[i];
            fx[i] = kl.pt_x; fy[i] = kl.pt_y; fang[i] = kl.angle; foct[i] = kl.octave;
        }
        ford[i] = order; fur[i] = A.uright ? A.uright[i] : -1.f;
        fdesc[2 * i] = ((const uint4*)A.desc)[2 * i]; fdesc[2 * i + 1] = ((const uint4*)A.desc)[2 * i + 1];
    }
    for (int i = lane; i < nq; i += 64) qbin[i] = -1;
    __syncthreads();
    int nmatches = 0, it = 0;
    long long stRescan = 0, cyRescan = 0; const long long tK0 = __builtin_readcyclecounter();
    for (int base = 0; base < nq; base += 64) {
        const int qi = base + lane, end = min(64, nq - base);
        unsigned long long ks[PROJ_K];
#pragma unroll
        for (int k = 0; k < PROJ_K; ++k) ks[k] = ~0ull;
        int cnt = 0, valid = 0, obs = 0; float qAngle = 0.f;
        if (lane < end) {
            const sslam_proj_query* Qp = A.q + qi;
            valid = Qp->valid; obs = Qp->obs_positive; qAngle = Qp->angle; cnt = T.cnt[qi];
            const unsigned long long* tp = T.top + (size_t)qi * PROJ_K;
#pragma unroll
            for (int k = 0; k < PROJ_K; ++k) ks[k] = tp[k];
        }
        int start = 0;
        while (start < end) {
            ++it;
            const bool active = lane >= start && lane < end && valid != 0;
            // first and second list entry whose feature is free now
            unsigned long long b = ~0ull, s2 = ~0ull; int nfree = 0;
            if (active) {
#pragma unroll
                for (int k = 0; k < PROJ_K; ++k) {
                    if (ks[k] == ~0ull) continue;
                    if (occ[proj_key_feature(ks[k])]) continue;
                    if (nfree == 0) b = ks[k]; else if (nfree == 1) s2 = ks[k];
                    ++nfree;
                }
            }
            const int need = A.mode == 0 ? 2 : 1;
            const bool exhausted = active && nfree < need && cnt > PROJ_K;      // the list ran dry although more candidates exist
            int acc = 0, bin = -1;
            if (active && !exhausted) proj_decide_keys(A, qAngle, b, s2, acc, bin);
            const int fb = b != ~0ull ? proj_key_feature(b) : -1, fs = (A.mode == 0 && s2 != ~0ull) ? proj_key_feature(s2) : -1;
            const int tag = (it << 6) | (63 - lane);
            if (active && acc) atomicMax(&stamp[fb], tag);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            bool clash = exhausted;
            if (active && !exhausted) {
                if (fb >= 0) { const int st = stamp[fb]; clash |= (st >> 6) == it && 63 - (st & 63) < lane; }
                if (fs >= 0) { const int st = stamp[fs]; clash |= (st >> 6) == it && 63 - (st & 63) < lane; }
            }
            const unsigned long long cm = __ballot(clash);
            const int c = cm ? __ffsll((long long)cm) - 1 : end;          // lanes below `start` are inactive: never set
            const bool commit = active && lane < c && acc;
            if (commit) {
                A.assigned[fb] = qi;
                if (obs) occ[fb] = 1;
                if (A.mode == 1 && A.checkOri) { qbin[qi] = bin; qidx[qi] = fb; atomicAdd(&hist[bin], 1); }
            }
            nmatches += __popcll(__ballot(commit));
            start = c;
            if (c < end && ((__ballot(exhausted) >> c) & 1ull)) {
                // query base + c is the first pending one and its list is spent: scan its candidates against the live occupancy
                ++stRescan; const long long tr0 = A.stats ? __builtin_readcyclecounter() : 0;
                const int jq = base + c;
                const sslam_proj_query Q = A.q[jq];
                const uint4 q0 = ((const uint4*)(A.qdesc + (size_t)jq * 32))[0], q1 = ((const uint4*)(A.qdesc + (size_t)jq * 32))[1];
                unsigned long long lb = ~0ull, ls = ~0ull;
                for (int i = lane; i < n; i += 64) {
                    unsigned long long kk;
                    if (!(featsInLds ? proj_candidate_lds(A, F, Q, q0, q1, i, occ, kk) : proj_candidate(A, Q, q0, q1, i, invW, invH, [&](int f) { return occ[f] != 0; }, kk))) continue;
                    if (kk < lb) { ls = lb; lb = kk; } else if (kk < ls) ls = kk;
                }
                const unsigned long long best = wave_min_u64(lb);
                const unsigned long long second = wave_min_u64(lb == best ? ls : lb);
                int a2 = 0, bn2 = -1;
                proj_decide_keys(A, Q.angle, best, A.mode == 0 ? second : ~0ull, a2, bn2);
                if (a2) {
                    const int f = proj_key_feature(best);
                    if (lane == 0) {
                        A.assigned[f] = jq;
                        if (Q.obs_positive) occ[f] = 1;
                        if (A.mode == 1 && A.checkOri) { qbin[jq] = bn2; qidx[jq] = f; atomicAdd(&hist[bn2], 1); }
                    }
                    ++nmatches;
                }
                start = c + 1;
                if (A.stats) cyRescan += __builtin_readcyclecounter() - tr0;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
    }
    __syncthreads();
    if (A.mode == 1 && A.checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            const int c = hist[i];
            if (c > max1) { max3 = max2; max2 = max1; max1 = c; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (c > max2) { max3 = max2; max2 = c; ind3 = ind2; ind2 = i; }
            else if (c > max3) { max3 = c; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
        int removed = 0;
        for (int i0 = 0; i0 < nq; i0 += 64) {
            const int i = i0 + lane;
            bool rm = false;
            if (i < nq) { const int bn = qbin[i]; rm = bn >= 0 && bn != ind1 && bn != ind2 && bn != ind3; if (rm) A.assigned[qidx[i]] = -2; }      // matched, then removed by the rotation check (src/ORBmatcher.cc:1465)
            removed += __popcll(__ballot(rm));
        }
        nmatches -= removed;
    }
    if (lane == 0) *A.nmatches = nmatches;
    if (lane == 0 && A.stats) { A.stats[0] = it; A.stats[1] = stRescan; A.stats[2] = cyRescan; A.stats[3] = __builtin_readcyclecounter() - tK0; }
}

// ---------------------------------------------------------------- SearchByBoW(KeyFrame*, Frame&)
// src/ORBmatcher.cc:159-291.  One wave: the shared vocabulary nodes are walked in ascending id, the keyframe features
// of a node sequentially (a matched frame feature is skipped by the later ones), the node's frame features lane-parallel
// with the list position in the reduction key (first strictly smaller distance wins).
struct BowArgs {
    const sslam_keypoint* kpKF; const uint8_t* dKF; const uint8_t* validKF;
    const sslam_keypoint* kpF; const uint8_t* dF; int nF;
    const int* ptrKF; const int* ptrF; int nnodes; const int* idxKF; const int* idxF;
    float nnratio; int checkOri; int* assigned; int* nmatches; int* qbin;   // qbin[nF]: rotation bin recorded for a frame feature
    const uint8_t* validF; int strictTh;     // KeyFrame-KeyFrame form (:525-658): candidates need a good map point of their own, and the gate is dist < TH_LOW
};

// A frame feature belongs to one vocabulary node, so the only order dependence of SearchByBoW -- a frame feature that is
// already matched is skipped (:216-217) -- stays inside a node: one wave per node walks that node's keyframe features in order
// (grid = 1 replays all nodes in order, used when a caller's lists share a feature between nodes).  The rotation histogram
// only needs counts; k_bow_finish builds it, prunes and counts.  assigned / qbin arrive as -1, *nmatches as 0.
__global__ __launch_bounds__(64) void k_search_bow(BowArgs A) {
    const int lane = threadIdx.x;
    for (int nd = blockIdx.x; nd < A.nnodes; nd += gridDim.x) {
        const int f0 = A.ptrF[nd], f1 = A.ptrF[nd + 1];
        for (int a = A.ptrKF[nd]; a < A.ptrKF[nd + 1]; ++a) {
            const int ik = A.idxKF[a];
            if (!A.validKF[ik]) continue;
            const uint4 q0 = ((const uint4*)(A.dKF + (size_t)ik * 32))[0], q1 = ((const uint4*)(A.dKF + (size_t)ik * 32))[1];
            unsigned long long b = ~0ull, s = ~0ull;
            for (int p = f0 + lane; p < f1; p += 64) {
                const int jf = A.idxF[p];
                if (A.assigned[jf] >= 0 || (A.validF && !A.validF[jf])) continue;
                const uint4* tp = (const uint4*)(A.dF + (size_t)jf * 32);
                const unsigned long long kk = ((unsigned long long)hamming256(q0, q1, tp[0], tp[1]) << 32) | (unsigned)(p - f0);
                if (kk < b) { s = b; b = kk; } else if (kk < s) s = kk;
            }
            const unsigned long long best = wave_min_u64(b);
            const unsigned long long second = wave_min_u64(b == best ? s : b);
            int bestDist1 = 256, bestDist2 = 256, bestIdxF = -1;
            if (best != ~0ull && (int)(best >> 32) < 256) { bestDist1 = (int)(best >> 32); bestIdxF = A.idxF[f0 + (int)(unsigned)best]; }
            if (second != ~0ull && (int)(second >> 32) < 256) bestDist2 = (int)(second >> 32);
            if ((A.strictTh ? bestDist1 < TH_LOW : bestDist1 <= TH_LOW) && (float)bestDist1 < __fmul_rn(A.nnratio, (float)bestDist2)) {
                if (lane == 0) {
                    if (A.checkOri) {
                        float rot = __fsub_rn(A.kpKF[ik].angle, A.kpF[bestIdxF].angle);
                        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                        int bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
                        if (bin == HISTO_LENGTH) bin = 0;
                        A.qbin[bestIdxF] = bin;
                    }
                    A.assigned[bestIdxF] = ik;
                }
                __syncthreads();          // the next keyframe feature of this node must see the assignment
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_bow_finish(int* __restrict__ assigned, const int* __restrict__ qbin, int nF, int checkOri, int* __restrict__ nmatches) {
    __shared__ int hist[HISTO_LENGTH];
    __shared__ int keep[3];
    __shared__ int total;
    const int t = threadIdx.x;
    if (t < HISTO_LENGTH) hist[t] = 0;
    if (t == 0) total = 0;
    __syncthreads();
    if (checkOri) {
        for (int i = t; i < nF; i += 256) if (assigned[i] >= 0) atomicAdd(&hist[qbin[i]], 1);
        __syncthreads();
        if (t == 0) {
            int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
            for (int i = 0; i < HISTO_LENGTH; ++i) {
                const int c = hist[i];
                if (c > max1) { max3 = max2; max2 = max1; max1 = c; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (c > max2) { max3 = max2; max2 = c; ind3 = ind2; ind2 = i; }
                else if (c > max3) { max3 = c; ind3 = i; }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
            keep[0] = ind1; keep[1] = ind2; keep[2] = ind3;
        }
        __syncthreads();
    }
    int cnt = 0;
    for (int i = t; i < nF; i += 256) {
        if (assigned[i] < 0) continue;
        if (checkOri) { const int bn = qbin[i]; if (bn != keep[0] && bn != keep[1] && bn != keep[2]) { assigned[i] = -1; continue; } }
        ++cnt;
    }
    atomicAdd(&total, cnt);
    __syncthreads();
    if (t == 0) *nmatches = total;
}

// ------------------------------------------------------------------ projection matchers, low-latency form
// Same semantics as k_search_proj, built for the single-frame call.  The frame's features are sorted by their
// GetFeaturesInArea order (grid column, row, index: one bitonic sort in LDS) and staged in LDS in that order (position,
// level, angle, occupancy, descriptors), so a query only scans the sorted range of the grid columns its window touches.
// Sixteen waves evaluate sixteen consecutive queries speculatively against the occupancy at the start of the round, each
// wave also taking its own accept / reject decision; wave 0 then commits them in query order and re-evaluates a query only
// if an earlier query of the same round occupied its best or second-best feature (removing any other candidate cannot
// change best / second, so the check is exact).  Global memory is touched once per round (the next round's queries are
// prefetched); the one-wave kernel above spends ~7 us per query on dependent global loads.
constexpr int PROJ_WAVES = 16;
constexpr int PROJ_MAXN = 2048;            // features that fit: 64 B each in sorted order + the sort keys
struct ProjLds { float* px; float* py; float* ang; int* oct; float* ur; int* occ; unsigned* ord; uint4* desc; int* colStart; int* stamp; };
struct ProjDecision { int acc, bestP, secondP, bin; };

// best / second-best candidate of one query (keys: dist | sorted position | level; the position is unique, so the level bits
// below it never take part in a comparison)
__device__ __forceinline__ void proj_scan(const ProjArgs& A, const ProjLds& S, const sslam_proj_query& Q, const uint4& q0, const uint4& q1, int lane,
                                          float invW, unsigned long long& best, unsigned long long& second, bool& anyOut) {
    unsigned long long b = ~0ull, s = ~0ull;
    bool any = false;
    int p0 = 0, p1 = A.n;
    if (A.kind == 0) {        // grid columns the window can touch (KeyFrame/Frame::GetFeaturesInArea's own cell range)
        const int c0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.u, A.minX), Q.radius), invW)));
        const int c1 = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.u, A.minX), Q.radius), invW)));
        if (c0 >= GRID_COLS || c1 < 0) { p0 = p1 = 0; }
        else { p0 = S.colStart[c0]; p1 = S.colStart[c1 + 1]; }
    }
    for (int p = p0 + lane; p < p1; p += 64) {
        const int oct = S.oct[p];
        if (A.kind == 0) {
            if (Q.min_level > 0 || Q.max_level >= 0) {
                if (oct < Q.min_level) continue;
                if (Q.max_level >= 0 && oct > Q.max_level) continue;
            }
            const float dx = __fsub_rn(S.px[p], Q.u), dy = __fsub_rn(S.py[p], Q.v);
            if (!(fabsf(dx) < Q.radius && fabsf(dy) < Q.radius)) continue;
        } else {
            const double mxp = 0.5 * (double)__fadd_rn(Q.u, Q.u2) - (double)S.px[p], myp = 0.5 * (double)__fadd_rn(Q.v, Q.v2) - (double)S.py[p];
            const float distance = (float)(mxp * mxp + myp * myp);
            if (distance > __fmul_rn(Q.radius, Q.radius)) continue;
            const float slope = __fsub_rn(__fdiv_rn(__fsub_rn(Q.v, Q.v2), __fsub_rn(Q.u, Q.u2)), S.ang[p]);
            if ((double)slope > (double)Q.radius * 0.01) continue;
            if (Q.min_level > 0 || Q.max_level > 0) {
                if (oct < Q.min_level) continue;
                if (Q.max_level >= 0 && oct > Q.max_level) continue;
            }
        }
        any = true;                              // vIndices non-empty
        if (S.occ[p]) continue;
        if (A.kind == 0 && A.uright) {
            const float ur = S.ur[p];
            if (ur > 0 && fabsf(__fsub_rn(Q.ur, ur)) > Q.radius) continue;
        }
        const unsigned long long kk = ((unsigned long long)hamming256(q0, q1, S.desc[2 * p], S.desc[2 * p + 1]) << 35) | ((unsigned long long)(unsigned)p << 4) | (unsigned)(oct & 15);
        if (kk < b) { s = b; b = kk; } else if (kk < s) s = kk;
    }
    anyOut = __ballot(any) != 0;
    best = wave_min_u64(b);
    second = wave_min_u64(b == best ? s : b);
}

// the reference's accept / reject logic on (best, second): thresholds, same-level ratio test (mode 0), rotation bin (mode 1)
__device__ __forceinline__ ProjDecision proj_decide(const ProjArgs& A, const ProjLds& S, float qAngle, unsigned long long b, unsigned long long s2) {
    ProjDecision D; D.acc = 0; D.bestP = -1; D.secondP = -2; D.bin = -1;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1;
    if (b != ~0ull) D.bestP = (int)((b >> 4) & 0x7FFFFFFF);
    if (s2 != ~0ull) D.secondP = (int)((s2 >> 4) & 0x7FFFFFFF);
    if (b != ~0ull && (int)(b >> 35) < 256) { bestDist = (int)(b >> 35); bestLevel = (int)(b & 15); }
    if (A.mode == 0 && s2 != ~0ull && (int)(s2 >> 35) < 256) { bestDist2 = (int)(s2 >> 35); bestLevel2 = (int)(s2 & 15); }
    if (bestDist <= A.thDist && b != ~0ull && (int)(b >> 35) < 256) {
        if (A.mode == 0 && bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(A.nnratio, (float)bestDist2)) return D;
        D.acc = 1;
        if (A.mode == 1 && A.checkOri) {
            float rot = __fsub_rn(qAngle, S.ang[D.bestP]);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
            if (bin == HISTO_LENGTH) bin = 0;
            D.bin = bin;
        }
    }
    return D;
}

__global__ __launch_bounds__(PROJ_WAVES * 64) void k_search_proj_lds(ProjArgs A) {
    extern __shared__ __align__(16) uint8_t dyn[];
    constexpr int NT = PROJ_WAVES * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = A.n, nq = A.nq;
    int* key2 = A.scratch + n; int* qbin = key2 + n; int* qidx = qbin + nq;      // same scratch layout as k_search_proj
    int N2 = 64; while (N2 < n) N2 <<= 1;
    ProjLds S;
    S.desc = (uint4*)dyn;
    S.px = (float*)(S.desc + 2 * (size_t)n); S.py = S.px + n; S.ang = S.py + n; S.ur = S.ang + n;
    S.oct = (int*)(S.ur + n); S.occ = S.oct + n; S.ord = (unsigned*)(S.occ + n); S.colStart = (int*)(S.ord + N2);
    S.stamp = S.colStart + GRID_COLS + 2;                  // [n] per sorted feature: (round << 5 | 31 - wave) of the earliest wave of the latest round that accepted it
    __shared__ int rAcc[PROJ_WAVES], rBestP[PROJ_WAVES], rSecondP[PROJ_WAVES], rBin[PROJ_WAVES], rObs[PROJ_WAVES];
    __shared__ int hist[HISTO_LENGTH];
    __shared__ int sh_nmatches;
    const sslam_keypoint* kps = (const sslam_keypoint*)A.feats;
    const sslam_keyline* kls = (const sslam_keyline*)A.feats;
    if (tid < HISTO_LENGTH) hist[tid] = 0;
    if (tid == 0) sh_nmatches = 0;
    const float invW = __fdiv_rn((float)GRID_COLS, __fsub_rn(A.maxX, A.minX));
    const float invH = __fdiv_rn((float)GRID_ROWS, __fsub_rn(A.maxY, A.minY));
    // sort keys: (grid column * ROWS + grid row) << 19 | index; features outside the grid (and the padding) sort last
    for (int i = tid; i < N2; i += NT) {
        unsigned k = 0xFFFFFFFFu;
        if (i < n) {
            A.assigned[i] = -1;
            if (A.kind == 0) {
                const int gx = (int)roundf(__fmul_rn(__fsub_rn(kps[i].x, A.minX), invW));
                const int gy = (int)roundf(__fmul_rn(__fsub_rn(kps[i].y, A.minY), invH));
                if (gx >= 0 && gx < GRID_COLS && gy >= 0 && gy < GRID_ROWS) k = ((unsigned)(gx * GRID_ROWS + gy) << 19) | (unsigned)i;
            } else k = (unsigned)i;                 // lines: GetLinesInArea scans in index order
        }
        S.ord[i] = k;
    }
    for (int i = tid; i < nq; i += NT) qbin[i] = -1;
    for (int i = tid; i < n; i += NT) S.stamp[i] = 0;
    __syncthreads();
    if (A.kind == 0) {
        for (int k = 2; k <= N2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < N2; i += NT) {
                    const int l = i ^ j;
                    if (l > i) {
                        const unsigned a = S.ord[i], b2 = S.ord[l];
                        const bool up = (i & k) == 0;
                        if ((a > b2) == up) { S.ord[i] = b2; S.ord[l] = a; }
                    }
                }
                __syncthreads();
            }
    }
    // stage the features in sorted order; A.n shrinks to the features that are in the grid (the rest can never be a candidate)
    int nIn = n;
    if (A.kind == 0) {
        int c = 0;
        for (int i = tid; i < n; i += NT) c += S.ord[i] != 0xFFFFFFFFu ? 1 : 0;
        c = wave_sum(c);
        if (lane == 0) atomicAdd(&sh_nmatches, c);
        __syncthreads();
        nIn = sh_nmatches;
        __syncthreads();
        if (tid == 0) sh_nmatches = 0;
    }
    for (int p = tid; p < nIn; p += NT) {
        const int i = (int)(S.ord[p] & 0x7FFFFu);
        if (A.kind == 0) { const sslam_keypoint kp = kps[i]; S.px[p] = kp.x; S.py[p] = kp.y; S.ang[p] = kp.angle; S.oct[p] = kp.octave; }
        else { const sslam_keyline kl = kls[i]; S.px[p] = kl.pt_x; S.py[p] = kl.pt_y; S.ang[p] = kl.angle; S.oct[p] = kl.octave; }
        S.occ[p] = A.occIn ? (int)A.occIn[i] : 0;
        S.ur[p] = A.uright ? A.uright[i] : -1.f;
        S.desc[2 * p] = ((const uint4*)A.desc)[2 * i]; S.desc[2 * p + 1] = ((const uint4*)A.desc)[2 * i + 1];
    }
    if (A.kind == 0) {        // colStart[c] = first sorted position whose grid column is >= c
        for (int p = tid; p <= nIn; p += NT) {
            const int colPrev = p == 0 ? -1 : (int)(S.ord[p - 1] >> 19) / GRID_ROWS;
            const int colCur = p == nIn ? GRID_COLS : (int)(S.ord[p] >> 19) / GRID_ROWS;
            for (int c = colPrev + 1; c <= colCur; ++c) S.colStart[c] = p;
        }
    }
    __syncthreads();
    ProjArgs B = A; B.n = nIn;
    // this wave's query of the first round
    sslam_proj_query Q; uint4 q0, q1;
    Q.valid = 0; Q.obs_positive = 0; Q.angle = 0.f;
    q0 = q1 = make_uint4(0, 0, 0, 0);
    if (wave < nq) { Q = A.q[wave]; q0 = ((const uint4*)(A.qdesc + (size_t)wave * 32))[0]; q1 = ((const uint4*)(A.qdesc + (size_t)wave * 32))[1]; }
    for (int base = 0; base < nq; base += PROJ_WAVES) {
        const int iq = base + wave;
        // prefetch the next round's query while this one is evaluated
        sslam_proj_query Qn; uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0;
        Qn.valid = 0; Qn.obs_positive = 0; Qn.angle = 0.f;
        const int iqn = iq + PROJ_WAVES;
        if (iqn < nq) { Qn = A.q[iqn]; n0 = ((const uint4*)(A.qdesc + (size_t)iqn * 32))[0]; n1 = ((const uint4*)(A.qdesc + (size_t)iqn * 32))[1]; }
        ProjDecision D; D.acc = 0; D.bestP = -1; D.secondP = -2; D.bin = -1;
        if (iq < nq && Q.valid) {
            unsigned long long best, second; bool any;
            proj_scan(B, S, Q, q0, q1, lane, invW, best, second, any);
            if (any) D = proj_decide(B, S, Q.angle, best, second);
        }
        if (lane == 0) { rAcc[wave] = D.acc; rBestP[wave] = D.bestP; rSecondP[wave] = D.secondP; rBin[wave] = D.bin; rObs[wave] = Q.obs_positive; }
        __syncthreads();
        if (wave == 0) {
            // lane w holds what wave w decided; the serial loop broadcasts with v_readlane, and lane t remembers the t-th
            // feature occupied during this round
            const int cnt = min(PROJ_WAVES, nq - base);
            const int myAcc = lane < cnt ? rAcc[lane] : 0, myBP = lane < cnt ? rBestP[lane] : -1, mySP = lane < cnt ? rSecondP[lane] : -2;
            const int myBin = lane < cnt ? rBin[lane] : -1, myObs = lane < cnt ? rObs[lane] : 0;
            int myTaken = -3, nTaken = 0, accepted = 0;
            // Fast path (round 3): the sixteen decisions are independent unless an earlier wave of this round accepted a feature that is this
            // wave's best or second-best (the only way an earlier query can change a later one, as above).  Every accepting wave stamps its
            // feature with (round, wave); a wave that finds an earlier wave's stamp of this round on its best or second-best sends the round
            // down the serial path below.  Otherwise all sixteen commit at once.
            const int roundTag = base / PROJ_WAVES + 1;
            if (lane < cnt && myAcc) atomicMax(&S.stamp[myBP], (roundTag << 5) | (31 - lane));
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            bool clash = false;
            if (lane < cnt) {
                if (myBP >= 0) { const int st = S.stamp[myBP]; clash |= (st >> 5) == roundTag && 31 - (st & 31) < lane; }
                if (mySP >= 0) { const int st = S.stamp[mySP]; clash |= (st >> 5) == roundTag && 31 - (st & 31) < lane; }
            }
            if (!__ballot(clash)) {
                if (lane < cnt && myAcc) {
                    const int fi = (int)(S.ord[myBP] & 0x7FFFFu), jq = base + lane;
                    A.assigned[fi] = jq;
                    if (myObs) S.occ[myBP] = 1;
                    if (A.mode == 1 && A.checkOri) { qbin[jq] = myBin; qidx[jq] = fi; atomicAdd(&hist[myBin], 1); }
                }
                accepted = __popcll(__ballot(lane < cnt && myAcc));
            } else
            for (int w = 0; w < cnt; ++w) {
                int acc = __builtin_amdgcn_readlane(myAcc, w), bP = __builtin_amdgcn_readlane(myBP, w), bin = __builtin_amdgcn_readlane(myBin, w);
                const int sP = __builtin_amdgcn_readlane(mySP, w);
                const int jq = base + w;
                if (nTaken > 0 && __ballot(lane < nTaken && (myTaken == bP || myTaken == sP))) {      // rare: re-evaluate against the updated occupancy
                    const sslam_proj_query Qw = A.q[jq];
                    const uint4 w0 = ((const uint4*)(A.qdesc + (size_t)jq * 32))[0], w1 = ((const uint4*)(A.qdesc + (size_t)jq * 32))[1];
                    unsigned long long b, s2; bool any2;
                    proj_scan(B, S, Qw, w0, w1, lane, invW, b, s2, any2);
                    ProjDecision R; R.acc = 0; R.bestP = -1; R.bin = -1;
                    if (any2) R = proj_decide(B, S, Qw.angle, b, s2);
                    acc = R.acc; bP = R.bestP; bin = R.bin;
                }
                if (!acc) continue;
                ++accepted;
                const int obsPositive = __builtin_amdgcn_readlane(myObs, w);
                if (lane == 0) {
                    const int fi = (int)(S.ord[bP] & 0x7FFFFu);
                    A.assigned[fi] = jq;
                    if (obsPositive) S.occ[bP] = 1;
                    if (A.mode == 1 && A.checkOri) { qbin[jq] = bin; qidx[jq] = fi; hist[bin]++; }
                }
                if (obsPositive) { if (lane == nTaken) myTaken = bP; ++nTaken; }
            }
            if (lane == 0) sh_nmatches += accepted;
        }
        __syncthreads();
        Q = Qn; q0 = n0; q1 = n1;
    }
    if (wave == 0) {
        int nmatches = sh_nmatches;
        if (A.mode == 1 && A.checkOri) {
            int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
            for (int i = 0; i < HISTO_LENGTH; ++i) {
                const int c = hist[i];
                if (c > max1) { max3 = max2; max2 = max1; max1 = c; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (c > max2) { max3 = max2; max2 = c; ind3 = ind2; ind2 = i; }
                else if (c > max3) { max3 = c; ind3 = i; }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
            int removed = 0;
            for (int i0 = 0; i0 < nq; i0 += 64) {
                const int i = i0 + lane;
                bool rm = false;
                if (i < nq) { const int bn = qbin[i]; rm = bn >= 0 && bn != ind1 && bn != ind2 && bn != ind3; if (rm) A.assigned[qidx[i]] = -2; }      // matched, then removed by the rotation check: the reference NULLs the pointer (src/ORBmatcher.cc:1465)
                removed += __popcll(__ballot(rm));
            }
            nmatches -= removed;
        }
        if (lane == 0) *A.nmatches = nmatches;
    }
}
