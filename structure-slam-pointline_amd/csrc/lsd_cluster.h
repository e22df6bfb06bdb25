// LSD sequential core, CLUSTER form: up to 64 frames per call, each with helper waves on SEVERAL compute units.  Part of lines.hip (included
// after lsd_regions.h, same anonymous namespace).  Not a standalone header.  docs/history/DESIGN_rounds_1-4.md 5e has the measurements.
//
// Its predecessor, the multi-wave form of rounds 2-4 (history 5c; removed in round 5), kept everything in one workgroup's LDS and was bound by what six helper
// waves next to the main wave could grow.  This form takes the helpers out of the main wave's workgroup: a frame owns nWG workgroups of one XCD (blockIdx % 8 == frame % 8: the observed dispatch rule, those share an L2; nothing
// below depends on it for correctness).  Workgroup 0 of a frame runs the MAIN wave and a FEEDER wave, the others three HELPER waves each.
// Shared state lives in global memory.
//
// Protocol (what is different from that multi-wave form, whose records and validation rule (b) it keeps: lsd_regions.h):
//   * THE PIXEL MAP IS MONOTONIC.  The main wave never releases a pixel in it: a seed it has to grow itself is grown on a private bitmap
//     that covers the whole frame (MARK_PRIV), refine() / reduce_region_radius release and re-mark there, and only the pixels that end up
//     USED are committed to the map.  A taken result commits its last list, as before.
//   * Therefore a helper's view may be arbitrarily stale (its L1, another XCD's L2): a pixel it saw USED is used now; a pixel it saw unused
//     and rejected by angle is rejected by the sequential run whatever its state; a pixel it accepted is in list A or B.  Check (b) of the
//     multi-wave form -- every point of A and B is unused NOW, read by the main wave from its own map -- is the whole validation; there are
//     no release events and no check (c).  (tests/sim/mw_proto.cpp, mode 1, is this protocol with real threads and stale views.)
//   * Results travel through global memory with L1-bypassing (sc1) stores and loads on both sides: per SUB-CHUNK of CL_SUB seed positions
//     {state, flag = doneLane | nres << 8} and up to CL_RES write-once result records, per helper a bump-allocated arena for the lists
//     (never reused within a frame: no ring, no waiting for space).  Publication order: lists, result record, s_waitcnt vmcnt(0), flag.
//     The main wave reads flag, then records, then lists.
//   * Sub-chunks are claimed in order by the helpers (global cursor, at most `window` sub-chunks ahead of the main wave); a claim is a CAS
//     on the state, and the main wave, arriving at a sub-chunk nobody has started, CASes it for itself -- it never waits for a helper that
//     is not there.  Every wait of the main wave is bounded.
//   * A helper with nothing to claim looks at its sub-chunk again (seeds without a result by now) and re-validates what it published; a
//     seed whose result has been overtaken by a commit is grown again and published as a second record (the later one counts).
//   * The feeder wave stages chunk states, records, lists and map values in LDS ahead of the main wave (ClSlot); the main wave numbers its
//     commits so that a staged value that may be out of date is read again.
#pragma once

#ifndef SSLAM_CL_SUB_SHIFT
#define SSLAM_CL_SUB_SHIFT 2          // sub-chunks of 16 / 8 / 4 / 2 positions: 6.69 / 6.49 / 6.29 / 6.90 ms per frame (2: the claims cost more than the finer dealing returns)
#endif
constexpr int CL_SUB_SHIFT = SSLAM_CL_SUB_SHIFT;
constexpr int CL_SUB = 1 << CL_SUB_SHIFT, CL_NSUB = 64 / CL_SUB;      // helpers claim sub-chunks of 16 seed positions (four helpers share a chunk of the main wave: the dense head of the seed list is where it waits)
constexpr int CL_RES = 2 * CL_SUB;         // result records per sub-chunk: one per position (the main wave's lanes hold the first CL_SUB) + as many for results a helper publishes AGAIN
                                           // after finding its first version overtaken by a commit (records are written once, never rewritten)
constexpr int CL_ARENA = 1 << 16;          // list words per helper per frame (MwRes::off is 16 bits)
constexpr int CL_LIST = 3072;              // LDS words per helper for the region in progress (lists A, B, F and the rectangle)
constexpr int CL_WAVES = 4;                // waves per workgroup (one per SIMD)
#ifndef SSLAM_CL_HELPERS_PER_WG
#define SSLAM_CL_HELPERS_PER_WG 3
#endif
constexpr int CL_HPW = SSLAM_CL_HELPERS_PER_WG;      // helper waves per helper workgroup: 3 with a 512 x 512 torus each (reach 254 pixels: the regions helpers used to give up on were a fifth of what the main wave grew itself), 4 with 256 x 256
typedef std::conditional<CL_HPW == 4, TorusHelper, TorusWide>::type ClTorus;
constexpr int CL_MAXWG = 16;               // workgroups per frame at most
#ifndef SSLAM_CL_GROUP
#define SSLAM_CL_GROUP 4
#endif
constexpr int CL_SCAN = 2 * 64 * SSLAM_CL_GROUP;          // LDS words of the main wave's look-ahead over the seed list (order entries + map values of one group of chunks)
constexpr int CL_SPIN_LIMIT = 1 << 18;     // polls before the main wave stops waiting for a helper (each poll is an L2 round trip)
struct ClSub { int state, flag; };         // per sub-chunk, zeroed per launch; state: 0 free, 1 the main wave's, 2 + h helper h's; flag = doneLane (0..16) | nres << 8
struct alignas(16) ClRec { MwRes m; unsigned pad[3]; };           // 32 bytes; records of sub-chunk sc: rec[sc * CL_RES ..)
static_assert(sizeof(ClRec) == 32, "result record layout");
struct ClCtl { int mainPos, finished, cursor, pad; long long stat[8]; };
static_assert(sizeof(ClCtl) <= NFA_STREAM_CTL_OFF && NFA_STREAM_CTL_OFF + sizeof(NfaStreamCtl) <= 512, "control block layout (lsd_plan.h: the streaming hand-over has a line of its own in the slot's zeroed head)");
__device__ __forceinline__ NfaStreamCtl* cl_ns(ClCtl* ctl) { return (NfaStreamCtl*)((uint8_t*)ctl + NFA_STREAM_CTL_OFF); }      // (k_lsd_regions_cl_stream only; zero otherwise)
// The main wave's workgroup runs ONE more wave, the FEEDER: it walks the seed list a few chunks ahead of the main wave and stages in LDS what
// the main wave would otherwise fetch from global memory with one dependent round trip after the other -- states and flags of the chunk's
// sub-chunks, the result records, and for every result of up to CL_STG points its list and the map values of its points (LDS-DMA loads,
// all of a chunk in flight at once).  A staged map value can be out of date: the main wave counts its commits, keeps the boxes of the last
// 64 in its lanes, the feeder notes the counter BEFORE it gathers, and a result whose box meets a commit younger than its gather is
// gathered again (same compute unit, same L1: a gather issued after the counter was read sees every store issued before the counter was
// written).  The main wave never waits for the feeder: a chunk that is not staged when it arrives takes the global path.
#ifndef SSLAM_CL_RING
#define SSLAM_CL_RING 3
#endif
#ifndef SSLAM_CL_STG
#define SSLAM_CL_STG 32
#endif
constexpr int CL_RING = SSLAM_CL_RING;     // chunks staged ahead
constexpr int CL_STG = SSLAM_CL_STG;       // list entries (A then B) staged per result; longer results take the global path
struct ClSlot {
    int chunk, ready, pad0, pad1;
    int st[CL_NSUB], fl[CL_NSUB];          // state / flag of the sub-chunks as staged (st 1: no result expected)
    MwRes rec[64];                         // record (l & (CL_SUB - 1)) of sub-chunk (l >> CL_SUB_SHIFT)
    int gseq[64];                          // the main wave's commit counter before the record's points were gathered; -1: not staged
    unsigned e[64][CL_STG];
    float v[64][CL_STG];
};
struct ClLocal { int mainChunk, commitSeq, finished, pad; };      // LDS: main wave -> feeder
constexpr int CL_RING_WORDS = (int)((sizeof(ClSlot) * CL_RING + sizeof(ClLocal) + 3) / 4);
struct ClShared {
    ClCtl* ctl; ClSub* sub; ClRec* rec; unsigned* arena; unsigned* specMap; unsigned* bigBm; int specW, specShift, nHelpers, window;
    __device__ __forceinline__ int cell(int x, int y) const { return (y >> specShift) * specW + (x >> specShift); }
};
// L1-bypassing accesses (global_load / global_store ... sc1): served by the L2 / memory, which is where the other compute units' stores are
__device__ __forceinline__ int g_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned g_ldu(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void g_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void g_stu(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cl_compiler_fence() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ void cl_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ bool cl_spec(const ClShared& cl, int x, int y) {
    if (cl.specShift < 0) return false;
    const int c = cl.cell(x, y);
    return (g_ldu(&cl.specMap[c >> 5]) >> (c & 31)) & 1u;
}

// stage clocks of the main wave (tools/cl_probe.py --cycles): compiled in with -DSSLAM_CL_CYCLES only
#ifdef SSLAM_CL_CYCLES
#define CL_CLK() __builtin_readcyclecounter()
#define CL_STAT(i, v) do { if (lane == 0) atomicAdd((unsigned long long*)&ctl->stat[i], (unsigned long long)(v)); } while (0)
#else
#define CL_CLK() 0ll
#define CL_STAT(i, v)
#endif
// ------------------------------------------------------------------ the main wave
// STREAM (k_lsd_regions_cl_stream: the default since round 5, SSLAM_NFA_STREAM=0 takes k_lsd_regions_cl): rectangle records go to the slot's staging array with L1-bypassing stores and a counter of the complete ones is
// published with every rectangle -- lsd_nfa.h's k_nfa_stream runs the NFA stage on them while this wave goes on.  Nothing else differs, and this wave never waits for it.
template <class G, int STREAM>      // G: the main wave's private bitmap: TorusFrame (LDS) or TorusGlobal (larger frames)
__device__ void cl_main(uint8_t* __restrict__ ws, const LsdPlan& P, int b, unsigned* __restrict__ qLds, unsigned* __restrict__ bmMain, unsigned* __restrict__ scanBuf,
                        double* __restrict__ red, float4* __restrict__ seedStash, const ClShared& cl, ClSlot* __restrict__ ring, ClLocal* __restrict__ loc) {
    const int lane = threadIdx.x & 63;
    uint8_t* base = ws + (size_t)b * P.frameBytes;
    Planes pl; pl.T = (float*)(base + P.offT); pl.Cs = (const float2*)(base + P.offCs); pl.S = (const int*)(base + P.offS); pl.tW = P.tW; pl.cW = P.cW;
    const unsigned* order = (const unsigned*)(base + P.offOrder);
    double* candOut = (double*)(base + P.offCand);
    Misc* misc = (Misc*)(base + P.offMisc);
    const int sw = P.sw, sh = P.sh;
    RegQ rq; rq.lds = qLds; rq.glb = (unsigned*)(base + P.offReg);
    const int nOrd = misc->nDefined;
    const double prec = P.prec;
    int nSeg = 0;
    long long clTaken = 0, clOwn = 0, clBad = 0, clWait = 0, clOwnChunks = 0, clRefused = 0;
    long long cWait = 0, cTake = 0, cOwn = 0, cRect = 0, clOwnRefused = 0;
    const long long cStart = CL_CLK();
    // The seed list is scanned CL_GROUP chunks at a time, one group ahead: order entries and their map values of group g + 1 are loaded
    // while group g is processed (two dependent round trips per chunk were a fifth of the main wave's time -- most chunks hold no unused
    // seed at all).  A value loaded early can be out of date; everything marked since its load lies inside the union of the boxes of
    // the commits since (accCur), and only the candidates inside that box are read again.
    constexpr int CL_GROUP = SSLAM_CL_GROUP;
    unsigned* scanIdx = scanBuf; float* scanT = (float*)(scanBuf + 64 * CL_GROUP);
    unsigned nIdx[CL_GROUP]; float nT[CL_GROUP];
    auto load_group = [&](int p0) {
#pragma unroll
        for (int j = 0; j < CL_GROUP; ++j) { const int q = p0 + 64 * j + lane; nIdx[j] = q < nOrd ? order[q] : 0xFFFFFFFFu; }
#pragma unroll
        for (int j = 0; j < CL_GROUP; ++j) nT[j] = nIdx[j] != 0xFFFFFFFFu ? pl.T[pl.ti(nIdx[j])] : NOTDEF_F;
    };
    load_group(0);
    unsigned accCurLo = 0xFFFFFFFFu, accCurHi = 0u, accNextLo = 0xFFFFFFFFu, accNextHi = 0u;      // packed x | y << 16 minima / maxima; Lo > Hi: empty
    bool accCurAny = false, accNextAny = false;
    int commitSeq = 0, logSeq = -1; unsigned logLo = 0u, logHi = 0u;      // lane (seq & 63) keeps the box of commit seq
    long long clStagedChunks = 0, clStagedTakes = 0, clRegather = 0, clLate = 0;
    for (int pos0 = 0; pos0 < nOrd; pos0 += 64) {
        const int gj = (pos0 >> 6) & (CL_GROUP - 1);
        if (gj == 0) {
#pragma unroll
            for (int j = 0; j < CL_GROUP; ++j) { scanIdx[64 * j + lane] = nIdx[j]; scanT[64 * j + lane] = nT[j]; }      // (every lane reads back its own entries only)
            accCurLo = accNextLo; accCurHi = accNextHi; accCurAny = accNextAny;
            accNextLo = 0xFFFFFFFFu; accNextHi = 0u; accNextAny = false;
            if (pos0 + 64 * CL_GROUP < nOrd) load_group(pos0 + 64 * CL_GROUP);
        }
        if (lane == 0) g_st(&cl.ctl->mainPos, pos0);
        lds_st(&loc->mainChunk, pos0 >> 6);
        const unsigned idx = scanIdx[64 * gj + lane];
        const bool have = idx != 0xFFFFFFFFu;
        const int tiSeed = have ? pl.ti(idx) : 0;
        float a0 = scanT[64 * gj + lane];
        if (accCurAny) {
            const int ix = (int)(idx & 0xFFFF), iy = (int)(idx >> 16);
            if (have && t_free(a0) && ix >= (int)(accCurLo & 0xFFFF) && ix <= (int)(accCurHi & 0xFFFF) && iy >= (int)(accCurLo >> 16) && iy <= (int)(accCurHi >> 16)) a0 = pl.T[tiSeed];
        }
        unsigned long long unM = __ballot(t_free(a0));
        if (!unM) continue;
        bool stashReady = false;
        auto fill_stash = [&]() {
            const double ar = (double)a0 * DEG2RAD;
            seedStash[lane] = make_float4(a0, (float)cos(ar), (float)sin(ar), __int_as_float((int)idx));
            stashReady = true;
        };
        // whose sub-chunks?  A helper that has started one owns it; the main wave claims the others (with unused seeds) for itself.  Lanes
        // 0..CL_NSUB-1 fetch state and flag of the chunk's headers in one round trip; lane l then holds record (l % CL_SUB) of sub-chunk (l / CL_SUB) -- and,
        // when the record carries its points itself (a small region: most of them), validates it ON ITS OWN: every lane gathers the map
        // values of its record's points, all records of the chunk in one round trip, instead of one list load + one gather per take.
        const int sc0 = (pos0 >> 6) * CL_NSUB;
        ClSub* S4 = &cl.sub[sc0];
        const int mySub = lane >> CL_SUB_SHIFT, myK = lane & (CL_SUB - 1);
        const ClRec* myRecPtr = &cl.rec[(size_t)(sc0 + mySub) * CL_RES + myK];
        int stv = 1, flv = 0, nsv = 0;               // per sub-chunk, in lane s: state, last flag read, records loaded
        ClSlot* SL = &ring[(pos0 >> 6) % CL_RING];
        // (LDS and vector loads at wave-uniform addresses still land in vector registers: without the readfirstlane the compiler treats what
        // depends on them -- here every branch on `staged`, further down the record of a late result and with it n, took and the loop's own
        // mask of unused seeds -- as divergent and wraps the whole seed loop in exec-mask bookkeeping)
        const bool staged = __builtin_amdgcn_readfirstlane(lds_ld(&SL->chunk)) == (pos0 >> 6) && __builtin_amdgcn_readfirstlane(lds_ld(&SL->ready)) == 1;      // the feeder has this chunk in LDS
        int myGseq = -1;                             // staged: the commit counter at the gather of my record's points (-1: not staged)
        if (staged) ++clStagedChunks;
        if (!staged) {
            const bool need = lane < CL_NSUB && ((unM >> (CL_SUB * lane)) & ((1ull << CL_SUB) - 1)) != 0;
            if (need) {
                stv = g_ld(&S4[lane].state); flv = g_ld(&S4[lane].flag);
                if (stv == 0) { stv = atomicCAS(&S4[lane].state, 0, 1); if (stv == 0) stv = 1; }
                if (stv < 2) { flv = 0; ++clOwnChunks; }
            }
        }
        cl_compiler_fence();
        MwRes myRes; myRes.w0 = myRes.w1 = myRes.w2 = myRes.lo = myRes.hi = 0u;
        auto fetch_records = [&](bool mine) {        // the calling lanes load their record (a record is complete before the flag counts it)
            if (mine) {
                const unsigned* r = (const unsigned*)myRecPtr;
                myRes.w0 = g_ldu(r); myRes.w1 = g_ldu(r + 1); myRes.w2 = g_ldu(r + 2); myRes.lo = g_ldu(r + 3); myRes.hi = g_ldu(r + 4);
            }
        };
        if (staged) {   // everything the global path fetches below, from LDS
            if (lane < CL_NSUB) { stv = SL->st[lane]; flv = SL->fl[lane]; nsv = min(flv >> 8, CL_SUB); }
            myRes = SL->rec[lane]; myGseq = SL->gseq[lane];
        } else {   // what is published already, all four sub-chunks at once
            const int myFlag = __builtin_amdgcn_ds_bpermute(mySub << 2, flv);      // the flag lane mySub holds
            fetch_records(myK < min(myFlag >> 8, CL_SUB));
            if (lane < CL_NSUB) nsv = min(flv >> 8, CL_SUB);
        }
        auto load_records = [&](int s, int nres) {      // records [nsv_s, nres) of sub-chunk s
            const int had = __builtin_amdgcn_readlane(nsv, s);
            if (mySub == s && myK >= had && myK < nres) myGseq = -1;
            fetch_records(mySub == s && myK >= had && myK < nres);
            if (lane == s) nsv = nres;
        };
        while (unM) {
            // (wave-uniform by construction -- ballots -- but carried through branches the compiler cannot prove uniform: pin it to scalar
            // registers, or the whole seed loop runs under exec-mask bookkeeping)
            unM = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unM >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)unM);
            const int first = __ffsll((long long)unM) - 1;
            unM &= unM - 1;
            const int s = first >> CL_SUB_SHIFT, f16 = first & (CL_SUB - 1);
            ClSub* H = &S4[s];
            const int stS = __builtin_amdgcn_readlane(stv, s);
            int owner = stS >= 2 ? stS - 2 : -1;
            int flag = __builtin_amdgcn_readlane(flv, s);
            float4 sd = make_float4(0.f, 0.f, 0.f, 0.f);
            double regAngle = 0;
            int n = -1;
            bool took = false, tookEmit = false; RectD tookRec;
            unsigned e0 = 0u;                            // lane i: point i of a taken region's first list (i < 64)
            bool wasRefused = false;
            unsigned bxLo = 0u, bxHi = 0xFFFFFFFFu;
            const long long c0 = CL_CLK();
            if (owner >= 0) {
                int spin = 0;
                while ((flag & 0xFF) <= f16) {
                    flag = __builtin_amdgcn_readfirstlane(g_ld(&H->flag));
                    if ((flag & 0xFF) > f16) break;
                    if (++spin > CL_SPIN_LIMIT) { owner = -1; ++clBad; if (lane == s) stv = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                clWait += spin;
                if (lane == s) flv = flag;
            }
            const long long c1 = CL_CLK(); cWait += c1 - c0;
            // validate + commit one published result.  gs / hl: the feeder's staging of it (gs < 0: not staged); returns whether it was taken
            auto try_take = [&](const MwRes& r, int gs, int hl) -> bool {
                const int nA = r.nA(), nB = r.nB(), nF = r.nF(), flags = r.flags();
                const unsigned* lstA = cl.arena + (size_t)owner * CL_ARENA + r.off();
                const unsigned* lstB = lstA + nA;
                const unsigned* lstF = (flags & MW_REDUCED) ? lstB + nB : (flags & MW_REFINED) ? lstB : lstA;
                bool ok = true;
                float v0 = 0.f; int ti0 = 0;
                // the rectangle's 24 words travel with the first list load (one round trip instead of two)
                const unsigned* rw = lstB + nB + ((flags & MW_REDUCED) ? nF : 0);
                const unsigned wv = ((flags & MW_EMIT) && lane < 24) ? g_ldu(rw + lane) : 0u;
                bool viaStage = false;
                if (gs >= 0 && nA + nB > 1) {
                    // staged result: list and map values are in LDS.  Values gathered before commit gs + 1 .. commitSeq can be out of date
                    // only inside those commits' boxes
                    const bool young = logSeq > gs && boxes_meet(logLo, logHi, r.lo, r.hi, 0);
                    const bool dirty = commitSeq - gs > 64 || __ballot(young) != 0;
                    const int i = lane;
                    bool usedNow = false;
                    if (i < nA + nB) {
                        const unsigned e = SL->e[hl][i]; const int ti = pl.ti(e);
                        const float v = dirty ? pl.T[ti] : SL->v[hl][i];
                        usedNow = !t_free(v); v0 = v; ti0 = ti; e0 = e;
                    }
                    ok = __ballot(usedNow) == 0;
                    viaStage = true; ++clStagedTakes; if (dirty) ++clRegather;
                }
                for (int bs = 0; !viaStage && ok && nA + nB > 1 && bs < nA + nB; bs += 64) {      // (b): everything the helper accepted on the way is unused now
                    const int i = bs + lane;
                    bool usedNow = false;
                    if (i < nA + nB) { const unsigned e = g_ldu(lstA + i); const int ti = pl.ti(e); const float v = pl.T[ti]; usedNow = !t_free(v); if (bs == 0) { v0 = v; ti0 = ti; e0 = e; } }
                    ok = __ballot(usedNow) == 0;
                }
                if (!ok) return false;
                if (nA + nB == 1) { if (lane == first) pl.T[tiSeed] = t_used(a0); }
                else if (!(flags & MW_REFINED) && nF <= 64) { if (lane < nF) pl.T[ti0] = t_used(v0); }
                else for (int i = lane; i < nF; i += 64) { unsigned* t = pl.Tb() + pl.ti(g_ldu(lstF + i)); *t |= USED_BIT; }
                took = true; n = nA; bxLo = r.lo; bxHi = r.hi;
                tookEmit = (flags & MW_EMIT) != 0;
                if (tookEmit) {
                    double* rd = (double*)&tookRec;
#pragma unroll
                    for (int j = 0; j < 12; ++j) rd[j] = __hiloint2double(__builtin_amdgcn_readlane((int)wv, 2 * j + 1), __builtin_amdgcn_readlane((int)wv, 2 * j));
                }
                clTaken += 1 + ((long long)n << 32);
                return true;
            };
            if (owner >= 0) {
                cl_compiler_fence();
                int nres = min(flag >> 8, CL_RES);
                if (min(nres, CL_SUB) > __builtin_amdgcn_readlane(nsv, s)) load_records(s, min(nres, CL_SUB));
                // the lanes hold records 0 .. CL_SUB-1 of the sub-chunk; a seed can have two of them (the helper published again): the later one counts
                const unsigned long long hit = __ballot(mySub == s && myK < min(nres, CL_SUB) && myRes.lane() == first);
                int triedK = -1;
                bool done = false;
                if (hit) {
                    const int hl = 63 - __clzll((long long)hit);
                    MwRes r;
                    r.w0 = (unsigned)__builtin_amdgcn_readlane((int)myRes.w0, hl); r.w1 = (unsigned)__builtin_amdgcn_readlane((int)myRes.w1, hl);
                    r.w2 = (unsigned)__builtin_amdgcn_readlane((int)myRes.w2, hl); r.lo = (unsigned)__builtin_amdgcn_readlane((int)myRes.lo, hl);
                    r.hi = (unsigned)__builtin_amdgcn_readlane((int)myRes.hi, hl);
                    triedK = hl & (CL_SUB - 1);
                    done = try_take(r, __builtin_amdgcn_readlane(myGseq, hl), hl);
                    if (!done) { ++clRefused; wasRefused = true; }
                }
                if (!done) {
                    // nothing for this seed, or what there was has been overtaken by a commit: the helper may have published (again) since the
                    // flag was read -- a second look, a re-validation.  One poll, then every record of the sub-chunk behind the one tried
                    flag = __builtin_amdgcn_readfirstlane(g_ld(&H->flag));
                    if (lane == s) flv = flag;
                    cl_compiler_fence();
                    nres = min(flag >> 8, CL_RES);
                    if (nres > triedK + 1) {
                        const unsigned* rb = (const unsigned*)&cl.rec[(size_t)(sc0 + s) * CL_RES];
                        MwRes best; best.w0 = best.w1 = best.w2 = best.lo = best.hi = 0u; int bestK = -1;
#pragma unroll
                        for (int kk = 0; kk < CL_RES; ++kk) {
                            if (kk > triedK && kk < nres) {
                                const unsigned* q = rb + kk * (sizeof(ClRec) / 4);
                                MwRes c;
                                c.w0 = (unsigned)__builtin_amdgcn_readfirstlane((int)g_ldu(q)); c.w1 = (unsigned)__builtin_amdgcn_readfirstlane((int)g_ldu(q + 1));
                                c.w2 = (unsigned)__builtin_amdgcn_readfirstlane((int)g_ldu(q + 2)); c.lo = (unsigned)__builtin_amdgcn_readfirstlane((int)g_ldu(q + 3));
                                c.hi = (unsigned)__builtin_amdgcn_readfirstlane((int)g_ldu(q + 4));
                                if (c.lane() == first) { best = c; bestK = kk; }
                            }
                        }
                        if (bestK >= 0) {
                            done = try_take(best, -1, 0);
                            if (done) { ++clLate; wasRefused = false; } else if (!wasRefused) { ++clRefused; wasRefused = true; }
                        }
                    }
                }
            }
            const long long c2 = CL_CLK(); cTake += c2 - c1;
            n = __builtin_amdgcn_readfirstlane(n); took = __builtin_amdgcn_readfirstlane((int)took) != 0; tookEmit = __builtin_amdgcn_readfirstlane((int)tookEmit) != 0;
            if (n < 0) {      // the main wave's own growth: private marks, the map is written at commit time only
                if (!stashReady) fill_stash();
                sd = seedStash[first];
                const int sxy = __float_as_int(sd.w), sx = sxy & 0xFFFF, sy = sxy >> 16;
                n = region_grow_m<true, MARK_PRIV, G>(sx, sy, sd.x, sd.y, sd.z, sw, sh, pl, rq, prec, regAngle, nullptr, bmMain);
                clOwn += 1 + ((long long)n << 32);
                if (wasRefused) clOwnRefused += 1 + ((long long)n << 32);
            }
            const long long c3 = CL_CLK(); cOwn += c3 - c2;
            n = __builtin_amdgcn_readfirstlane(n);
            RectD rec;
            bool emit = false;
            const bool tooSmall = n < P.minRegSize;       // (of the region as first grown: refine() may shrink a kept region below the minimum)
            if (took) {
                if (!tooSmall) { emit = tookEmit; rec = tookRec; }
            } else {
                if (!tooSmall) {
                    bool refined = false; unsigned evLo = 0xFFFFFFFFu, evHi = 0u;
                    long long cycs[3] = {0, 0, 0};
                    SpecLists sl; sl.bm = bmMain; sl.free = nullptr; sl.cap = 0; sl.nB = 0; sl.reduced = false; sl.gaveUp = false;
                    emit = rect_refine<true, MARK_PRIV, false, G>(P, sd, n, regAngle, rq, pl, red, rec, refined, evLo, evHi, cycs, &sl, nullptr);
                }
                // commit: what is left in the list stays USED (a region below the minimum size, a region refine() gave up on: all of it)
                for (int i = lane; i < n; i += 64) {
                    const unsigned e = rq.get_n(i, n);
                    const int px = (int)(e & 0xFFFF), py = (int)(e >> 16);
                    unsigned* t = pl.Tb() + pl.ti(px, py); *t |= USED_BIT;
                    const int bi = bm_bit<G>(px, py); atomicAnd(&bmMain[bi >> 5], ~(1u << (bi & 31)));
                }
                if (n <= QCAP) list_bbox(rq.lds, n, lane, bxLo, bxHi); else { bxLo = 0u; bxHi = 0xFFFFFFFFu; }
            }
            n = __builtin_amdgcn_readfirstlane(n); emit = __builtin_amdgcn_readfirstlane((int)emit) != 0;
            if (!took) cRect += CL_CLK() - c3;
            accCurLo = pk_min_u16(accCurLo, bxLo); accCurHi = pk_max_u16(accCurHi, bxHi); accNextLo = pk_min_u16(accNextLo, bxLo); accNextHi = pk_max_u16(accNextHi, bxHi);
            accCurAny = accNextAny = true;
            ++commitSeq;                                  // (after the commit's stores were issued: the feeder reads the counter before it gathers)
            if (lane == (commitSeq & 63)) { logLo = bxLo; logHi = bxHi; logSeq = commitSeq; }
            lds_st(&loc->commitSeq, commitSeq);
            if (tooSmall) {
                // too small: rejected, its pixels stay used.  Which candidates of this chunk did it take?
                for (int k = 1; k < n; ++k) {
                    const unsigned e = took ? (unsigned)__builtin_amdgcn_readlane((int)e0, k) : rq.lds[k];      // (minRegSize < 64)
                    unM &= ~__ballot(idx == e);
                }
                continue;
            }
            {   // candidates of this chunk inside the box of what was just marked: are they still unused?
                const int ix = (int)(idx & 0xFFFF), iy = (int)(idx >> 16);
                const bool chk = have && ((unM >> lane) & 1ull) && ix >= (int)(bxLo & 0xFFFF) && ix <= (int)(bxHi & 0xFFFF) && iy >= (int)(bxLo >> 16) && iy <= (int)(bxHi >> 16);
                bool usedNow = false;
                if (chk) usedNow = !t_free(pl.T[tiSeed]);
                unM &= ~__ballot(usedNow);
            }
            if (!emit) continue;
            if (STREAM) {
                if (nSeg < MAX_SEG) {
                    // Publication lags by one record: the stores of record nSeg - 1 were issued a region ago, so the wait in front of the counter costs
                    // nothing, and every rectangle but the frame's last is handed over as soon as the next one exists (the last one with candFinal).
                    cl_stores_done();
                    if (lane == 0) {
                        g_st(&cl_ns(cl.ctl)->candReady, nSeg);                 // records [0, nSeg) are complete
                        unsigned long long* o = (unsigned long long*)(cl.arena + (size_t)CL_ARENA * (size_t)max(1, cl.nHelpers)) + (size_t)nSeg * 12;      // the staging array lies behind the list arenas (lines.hip: stageOff)
                        const double v[12] = {rec.x1, rec.y1, rec.x2, rec.y2, rec.width, rec.x, rec.y, rec.theta, rec.dx, rec.dy, rec.prec, rec.p};
#pragma unroll
                        for (int q = 0; q < 12; ++q) __hip_atomic_store(o + q, (unsigned long long)__double_as_longlong(v[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            } else
            if (nSeg < MAX_SEG && lane == 0) {
                double* o = candOut + (size_t)nSeg * 12;
                o[0] = rec.x1; o[1] = rec.y1; o[2] = rec.x2; o[3] = rec.y2; o[4] = rec.width; o[5] = rec.x; o[6] = rec.y;
                o[7] = rec.theta; o[8] = rec.dx; o[9] = rec.dy; o[10] = rec.prec; o[11] = rec.p;
            }
            ++nSeg;
        }
    }
    lds_st(&loc->finished, 1);
    if (STREAM) { cl_stores_done(); if (lane == 0) g_st(&cl_ns(cl.ctl)->candFinal, 1 + min(nSeg, MAX_SEG)); }
    if (lane == 0) {
        g_st(&cl.ctl->finished, 1);
        misc->nCand = min(nSeg, MAX_SEG); if (nSeg > MAX_SEG) misc->overflow = 1;
        misc->cyc[0] = clWait; misc->cyc[1] = clOwnChunks; misc->cyc[2] = clRefused; misc->cyc[3] = clStagedChunks | (clRegather << 32); misc->cyc[4] = clStagedTakes | (clLate << 32);
#ifdef SSLAM_CL_CYCLES
        misc->cyc[0] = cWait; misc->cyc[1] = cTake; misc->cyc[2] = cOwn; misc->cyc[3] = cRect; misc->cyc[4] = CL_CLK() - cStart;
        cl.ctl->stat[4] = clOwnRefused; cl.ctl->stat[5] = clOwn;
#endif
        misc->cyc[5] = clTaken; misc->cyc[6] = clOwn; misc->cyc[7] = clBad;
    }
}

// ------------------------------------------------------------------ the feeder wave (second wave of the main wave's workgroup)
__device__ void cl_feeder(uint8_t* __restrict__ ws, const LsdPlan& P, int b, const ClShared& cl, ClSlot* __restrict__ ring, ClLocal* __restrict__ loc) {
    const int lane = threadIdx.x & 63;
    uint8_t* base = ws + (size_t)b * P.frameBytes;
    Planes pl; pl.T = (float*)(base + P.offT); pl.Cs = (const float2*)(base + P.offCs); pl.S = (const int*)(base + P.offS); pl.tW = P.tW; pl.cW = P.cW;
    const unsigned* order = (const unsigned*)(base + P.offOrder);
    const Misc* misc = (const Misc*)(base + P.offMisc);
    const int nOrd = misc->nDefined, nChunks = (nOrd + 63) >> 6;
    const int mySub = lane >> CL_SUB_SHIFT, myK = lane & (CL_SUB - 1);
    if (cl.window < 0) return;
    for (int c = 1; c < nChunks; ++c) {
        // the slot of chunk c is free once the main wave is past chunk c - CL_RING
        int mc = 0;
        for (int spin = 0;; ++spin) {
            if (lds_ld(&loc->finished) || spin > (1 << 24)) return;
            mc = lds_ld(&loc->mainChunk);
            if (c < mc + CL_RING) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (c <= mc) continue;                                   // the main wave is there already: it takes the global path
        const int q = c * 64 + lane;
        const unsigned idx = q < nOrd ? order[q] : 0xFFFFFFFFu;
        const float a0 = idx != 0xFFFFFFFFu ? pl.T[pl.ti(idx)] : NOTDEF_F;
        const unsigned long long unM = __ballot(t_free(a0));
        if (!unM) continue;
        ClSlot* S = &ring[c % CL_RING];
        if (lane == 0) { S->ready = 0; S->chunk = c; }
        // states and flags of the sub-chunks with unused seeds; a helper still in its first pass is waited for until the main wave gets here
        const ClSub* S4 = &cl.sub[c * CL_NSUB];
        const bool need = lane < CL_NSUB && ((unM >> (CL_SUB * lane)) & ((1ull << CL_SUB) - 1)) != 0;
        int stv = 1, flv = 0;
        bool pending = true;
        for (int spin = 0; spin < (1 << 22); ++spin) {
            if (need) { stv = g_ld(&S4[lane].state); flv = g_ld(&S4[lane].flag); }
            pending = __ballot(need && (stv < 2 || (flv & 0xFF) < CL_SUB)) != 0;      // nobody has started it, or its helper is still in pass 0
            if (!pending || lds_ld(&loc->mainChunk) >= c || lds_ld(&loc->finished)) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (pending) continue;                                   // not staged: the main wave sorts it out itself
        cl_compiler_fence();
        if (lane < CL_NSUB) { S->st[lane] = need ? stv : 1; S->fl[lane] = need ? flv : 0; }
        const int myFlag = __builtin_amdgcn_ds_bpermute(mySub << 2, flv);
        const int myOwner = __builtin_amdgcn_ds_bpermute(mySub << 2, stv) - 2;
        MwRes R; R.w0 = R.w1 = R.w2 = R.lo = R.hi = 0u;
        const bool valid = myK < min(myFlag >> 8, CL_SUB);
        if (valid) {
            const unsigned* r = (const unsigned*)&cl.rec[(size_t)(c * CL_NSUB + mySub) * CL_RES + myK];
            R.w0 = g_ldu(r); R.w1 = g_ldu(r + 1); R.w2 = g_ldu(r + 2); R.lo = g_ldu(r + 3); R.hi = g_ldu(r + 4);
        }
        S->rec[lane] = R; S->gseq[lane] = -1;
        const int cnt = valid ? R.nA() + R.nB() : 0;
        const bool stageable = cnt > 1 && cnt <= CL_STG;
        const unsigned long long stM = __ballot(stageable);
        const unsigned* myList = cl.arena + (size_t)max(myOwner, 0) * CL_ARENA + R.off();
        // the lists: LDS-DMA, every result of the chunk in flight at once
        for (unsigned long long m = stM; m; m &= m - 1) {
            const int r = __ffsll((long long)m) - 1;
            const int cr = __builtin_amdgcn_readlane(cnt, r);
            const unsigned long long lp = (unsigned long long)(size_t)myList;
            const unsigned* lst = (const unsigned*)(size_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(lp >> 32), r) << 32) | (unsigned)__builtin_amdgcn_readlane((int)lp, r));
            if (lane < cr) __builtin_amdgcn_global_load_lds(lst + lane, &S->e[r][0], 4, 0, 16);
        }
        cl_stores_done();
        // the map values of their points; the main wave's commit counter first
        const int seq = lds_ld(&loc->commitSeq);
        cl_compiler_fence();
        for (unsigned long long m = stM; m; m &= m - 1) {
            const int r = __ffsll((long long)m) - 1;
            const int cr = __builtin_amdgcn_readlane(cnt, r);
            if (lane < cr) { const unsigned e = ((volatile unsigned*)&S->e[r][0])[lane]; __builtin_amdgcn_global_load_lds(pl.T + pl.ti(e), &S->v[r][0], 4, 0, 0); }
        }
        cl_stores_done();
        // (two commits of margin: the counter is an LDS store, the commit's marks are vector stores issued before it -- the gather was issued
        // after the counter was read and follows them through the compute unit's one vector-memory pipe, but nothing is lost by not relying on it)
        if (stageable) S->gseq[lane] = max(seq - 2, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        lds_st(&S->ready, 1);
    }
}

// ------------------------------------------------------------------ a helper wave
// Claims a chunk, grows its unused seeds in order (each seed on its own, on its view of the pixel map + private marks), publishes.  While
// it cannot claim another chunk (window) and the main wave is still in front of this one it looks at the chunk again: whatever is unused by
// then and has no result -- seeds the shared map had talked it out of, seeds it gave up on -- is grown as well.  What its results would mark
// stays in the shared map until the main wave has passed the chunk (reap).
constexpr int CL_FIFO = 64;                // sub-chunks a helper can have published and not yet retired from the shared map
__device__ void cl_helper(int h, uint8_t* __restrict__ ws, const LsdPlan& P, int b, const ClShared& cl, unsigned* __restrict__ listBuf, unsigned* __restrict__ bm,
                          float4* __restrict__ stash, double* __restrict__ red) {
    typedef ClTorus G;
    const int lane = threadIdx.x & 63;
    uint8_t* base = ws + (size_t)b * P.frameBytes;
    Planes pl; pl.T = (float*)(base + P.offT); pl.Cs = (const float2*)(base + P.offCs); pl.S = (const int*)(base + P.offS); pl.tW = P.tW; pl.cW = P.cW;
    const unsigned* order = (const unsigned*)(base + P.offOrder);
    const Misc* misc = (const Misc*)(base + P.offMisc);
    const int sw = P.sw, sh = P.sh, nOrd = misc->nDefined, nSubs = (nOrd + CL_SUB - 1) / CL_SUB;
    ClCtl* ctl = cl.ctl;
    unsigned* arena = cl.arena + (size_t)h * CL_ARENA;
    int ah = 0;                                                   // arena words used
    if (cl.window < 0) return;                                    // (test knob: the main wave alone)
    constexpr int LISTCAP = CL_LIST - CL_FIFO;                    // the tail of the list buffer holds the FIFO of published chunks: chunk << 8 | results
    unsigned* fifo = listBuf + LISTCAP;
    int fHead = 0, fTail = 0;                                     // [fTail, fHead) outstanding
    // retire the chunks the main wave has passed: their regions leave the shared map
    auto reap = [&]() {
        if (cl.specShift < 0) { fTail = fHead; return; }
        const int mc = g_ld(&ctl->mainPos) >> 6;
        while (fTail < fHead) {
            const unsigned ent = fifo[fTail & (CL_FIFO - 1)];
            const int sc = (int)(ent >> 8), k = (int)(ent & 0xFF);
            if (sc / CL_NSUB >= mc && !g_ld(&ctl->finished)) break;
            for (int kk = 0; kk < k; ++kk) {
                const unsigned* r = (const unsigned*)&cl.rec[(size_t)sc * CL_RES + kk];
                MwRes R; R.w0 = g_ldu(r); R.w1 = g_ldu(r + 1); R.w2 = g_ldu(r + 2); R.lo = 0; R.hi = 0;
                const unsigned* lstF = arena + R.off() + ((R.flags() & MW_REDUCED) ? R.nA() + R.nB() : (R.flags() & MW_REFINED) ? R.nA() : 0);
                for (int i = lane; i < R.nF(); i += 64) { const unsigned e = g_ldu(lstF + i); const int clc = cl.cell((int)(e & 0xFFFF), (int)(e >> 16)); atomicAnd(&cl.specMap[clc >> 5], ~(1u << (clc & 31))); }
            }
            ++fTail;
        }
    };
    for (;;) {
        // ---- the next chunk nobody has, not further than `window` chunks in front of the main wave
        int sc = 0;                                               // the sub-chunk; c = the main wave's chunk it belongs to
        if (lane == 0) sc = atomicAdd(&ctl->cursor, 1);
        sc = __builtin_amdgcn_readfirstlane(sc);
        const int c = sc / CL_NSUB;
        if (sc >= nSubs) { for (int spin = 0; fTail < fHead && spin < (1 << 20) && !g_ld(&ctl->finished); ++spin) { reap(); __builtin_amdgcn_s_sleep(16); } return; }
        bool gone = false;
        for (int spin = 0;; ++spin) {
            if (g_ld(&ctl->finished)) return;
            reap();
            const int mc = g_ld(&ctl->mainPos) >> 6;
            if (c < mc) { gone = true; if (lane == 0) atomicMax(&ctl->cursor, mc * CL_NSUB); break; }      // the main wave is already past it: the cursor jumps to where it is
            if (sc <= mc * CL_NSUB + cl.window && fHead - fTail < CL_FIFO) break;
            if (spin > (1 << 22)) return;
            __builtin_amdgcn_s_sleep(8);
        }
        if (gone) continue;
        ClSub* H = &cl.sub[sc];
        int got = 0;
        if (lane == 0) got = atomicCAS(&H->state, 0, 2 + h);
        if (__builtin_amdgcn_readfirstlane(got) != 0) continue;  // the main wave took it
        // ---- the sub-chunk's seed candidates (lanes 0..15)
        const int q = sc * CL_SUB + lane, laneBase = (sc % CL_NSUB) * CL_SUB;      // results are numbered by the lane of the main wave's chunk
        const bool have = lane < CL_SUB && q < nOrd;
        const unsigned idx = have ? order[q] : 0u;
        const int cy = idx >> 16, cx = idx & 0xFFFF, tiSeed = pl.ti(idx);
        unsigned long long haveRes = 0;      // seeds with a result, or that this helper gave up on
        int k = 0; bool room = true;
        // what this helper has published for the sub-chunk (for the re-validation of later passes): seed lane, list offset, points of A and B
        int pubLane[CL_RES], pubOff[CL_RES], pubCnt[CL_RES]; unsigned pubLive = 0u;
#pragma unroll
        for (int j = 0; j < CL_RES; ++j) { pubLane[j] = 0; pubOff[j] = 0; pubCnt[j] = 0; }
        for (int pass = 0; room; ++pass) {
            if (pass > 0) {
                // another look only while there is nothing else to do (the next chunk is outside the window) and the main wave is still in front
                reap();
                const int mp = g_ld(&ctl->mainPos);
                if (mp >= c * 64 || g_ld(&ctl->finished)) break;
                if (g_ld(&ctl->cursor) <= (mp >> 6) * CL_NSUB + cl.window && fHead - fTail < CL_FIFO - 1) break;
                __builtin_amdgcn_s_sleep(8);
                asm volatile("buffer_inv sc1" ::: "memory");
                // Re-validation: a published result that a commit has overtaken since (one of its points is used now) would be refused by
                // the main wave, which then grows the seed itself.  While this helper has nothing else to do it checks its results the way
                // the main wave will, and a seed whose result died is grown again below and published as a NEW record (the later one counts).
#pragma unroll
                for (int j = 0; j < CL_RES; ++j) {
                    if (j < k && ((pubLive >> j) & 1u) && pubCnt[j] > 1) {
                        bool usedAny = false;
                        for (int bs = 0; bs < pubCnt[j]; bs += 64) {
                            const int i = bs + lane;
                            bool u = false;
                            if (i < pubCnt[j]) u = !t_free(pl.T[pl.ti(g_ldu(arena + pubOff[j] + i))]);
                            usedAny = usedAny || __ballot(u) != 0;
                        }
                        if (usedAny) { pubLive &= ~(1u << j); haveRes &= ~(1ull << pubLane[j]); CL_STAT(6, 1); }
                    }
                }
            }
            const float a0 = have ? pl.T[tiSeed] : NOTDEF_F;
            unsigned long long unM = __ballot(t_free(a0) && (pass > 0 || !cl_spec(cl, cx, cy))) & ~haveRes;
            if (unM) {
                const double ar = (double)a0 * DEG2RAD;
                stash[lane] = make_float4(a0, (float)cos(ar), (float)sin(ar), __int_as_float(cx | (cy << 16)));
            }
            while (unM) {
                if (g_ld(&ctl->mainPos) > c * 64 || g_ld(&ctl->finished)) { CL_STAT(3, 1 + __popcll(unM)); room = false; break; }      // the main wave has passed this chunk
                const int first = __ffsll((long long)unM) - 1;
                unM &= unM - 1;
                const float4 sd = stash[first];
                const int sxy = __float_as_int(sd.w), sx = sxy & 0xFFFF, sy = sxy >> 16;
                asm volatile("buffer_inv sc1" ::: "memory");      // this compute unit's L1 may hold lines from before the main wave's latest marks: start from the L2's view
                if (!t_free(pl.T[pl.ti(sx, sy)])) continue;       // taken since the chunk was scanned
                if (pass == 0 && cl_spec(cl, sx, sy)) continue;   // ... or about to be
                if (k >= CL_RES || ah + 3 * QCAP + 24 > CL_ARENA) { CL_STAT(0, 1 + __popcll(unM)); room = false; break; }      // no room left: the main wave grows the rest itself
                RegQ rq; rq.lds = listBuf; rq.glb = nullptr;
                double regAngle = 0;
                const int capN = min(QCAP, LISTCAP - 24);
                int n = region_grow_w<true, false, MARK_SPEC, G>(sx, sy, sd.x, sd.y, sd.z, sw, sh, pl, rq, P.prec, regAngle, nullptr, bm, capN);
                if (n < 0) {                                      // too long for a helper: release its marks, the main wave grows this one
                    for (int i = lane; i < -n; i += 64) { const unsigned e = rq.lds[i]; const int bi = bm_bit<G>((int)(e & 0xFFFF), (int)(e >> 16)); atomicAnd(&bm[bi >> 5], ~(1u << (bi & 31))); }
                    haveRes |= 1ull << first; CL_STAT(1, 1 + ((long long)(-n) << 32));
                    continue;
                }
                const int nA = n;
                unsigned* lstA = rq.lds;
                SpecLists sl; sl.bm = bm; sl.free = lstA + nA; sl.cap = LISTCAP - 24 - nA; sl.nB = 0; sl.reduced = false; sl.gaveUp = false;
                bool emit = false, refined = false; unsigned dLo = 0, dHi = 0; long long cycs[3];
                RectD rec;
                if (n >= P.minRegSize) {
                    emit = rect_refine<true, MARK_SPEC, false, G>(P, sd, n, regAngle, rq, pl, red, rec, refined, dLo, dHi, cycs, &sl, nullptr);
                    if (sl.gaveUp) { haveRes |= 1ull << first; CL_STAT(2, 1); continue; }      // its marks are released; the main wave handles this seed
                }
                // marks still set: the final list (rq.lds[0..n)).  They go -- the next region is grown on its own.
                for (int i = lane; i < n; i += 64) { const unsigned e = rq.lds[i]; const int bi = bm_bit<G>((int)(e & 0xFFFF), (int)(e >> 16)); atomicAnd(&bm[bi >> 5], ~(1u << (bi & 31))); }
                const int nAB = nA + sl.nB;
                int total = nAB + (sl.reduced ? n : 0);
                if (emit) {
                    if (lane == 0) {
                        int* rw = (int*)(lstA + total); const double* rd = (const double*)&rec;
#pragma unroll
                        for (int j = 0; j < 12; ++j) { rw[2 * j] = __double2loint(rd[j]); rw[2 * j + 1] = __double2hiint(rd[j]); }
                    }
                    total += 24;
                }
                unsigned lo, hi;
                list_bbox(lstA, nAB, lane, lo, hi);
                // ---- publish: lists, record, (stores complete), flag
                for (int i = lane; i < total; i += 64) g_stu(arena + ah + i, lstA[i]);
                {
                    const unsigned flags = (refined ? MW_REFINED : 0) | (sl.reduced ? MW_REDUCED : 0) | (emit ? MW_EMIT : 0);
                    unsigned* r = (unsigned*)&cl.rec[(size_t)sc * CL_RES + k];
                    unsigned w = 0u;
                    if (lane == 0) w = (unsigned)(laneBase + first) | (flags << 8);
                    else if (lane == 1) w = (unsigned)ah | ((unsigned)nA << 16);
                    else if (lane == 2) w = (unsigned)sl.nB | ((unsigned)n << 16);
                    else if (lane == 3) w = lo;
                    else if (lane == 4) w = hi;
                    if (lane < 5) g_stu(r + lane, w);
                }
                cl_stores_done();
#pragma unroll
                for (int j = 0; j < CL_RES; ++j) if (j == k) { pubLane[j] = first; pubOff[j] = ah; pubCnt[j] = nAB; }
                pubLive |= 1u << k;
                ++k; ah += total; haveRes |= 1ull << first;
                if (lane == 0) g_st(&H->flag, (pass == 0 ? first + 1 : CL_SUB) | (k << 8));
                // what the region leaves USED enters the shared map that steers seed choice; candidates of this chunk inside it are dropped
                if (cl.specShift >= 0) {
                    for (int i = lane; i < n; i += 64) {
                        const unsigned e = rq.lds[i]; const int clc = cl.cell((int)(e & 0xFFFF), (int)(e >> 16));
                        atomicOr(&cl.specMap[clc >> 5], 1u << (clc & 31));
                    }
                    if (n > 1 && pass == 0) {
                        cl_stores_done();
                        unM &= ~__ballot(have && cl_spec(cl, cx, cy));
                    }
                }
            }
            if (pass == 0 && lane == 0) g_st(&H->flag, CL_SUB | (k << 8));
        }
        if (k > 0 && cl.specShift >= 0) { if (lane == 0) fifo[fHead & (CL_FIFO - 1)] = ((unsigned)sc << 8) | (unsigned)k; ++fHead; }
    }
}

// One frame = nWG workgroups of one XCD (blockIdx % 8 == frame % 8; role = (blockIdx / 8) % nWG; up to eight frames per XCD); dynamic LDS: per wave a list buffer of CL_LIST words and a
// 256 x 256-bit torus, in front of them (role 0 only) the main wave's region queue and its frame-wide bitmap.
__global__ __launch_bounds__(64 * CL_WAVES) void k_lsd_regions_cl(uint8_t* __restrict__ ws, LsdPlan P, uint8_t* __restrict__ clArea, size_t clFrameBytes, int nframes, int nWG,
                                                                 int specWords, int specShift, int window) {
    extern __shared__ __align__(16) unsigned dynLds[];
    __shared__ double red[CL_WAVES][3 * 64];
    __shared__ float4 stashes[CL_WAVES][64];
    // blocks with the same (blockIdx % 8) share an XCD; an XCD hosts frames xcd, xcd + 8, xcd + 16, ... (nWG workgroups each)
    const int j = blockIdx.x >> 3, b = (blockIdx.x & 7) + 8 * (j / nWG), role = j % nWG, wave = threadIdx.x >> 6;
    const bool nofeed = (window & (1 << 20)) != 0;      // (experiment knob: no feeder wave)
    if (window >= 0) window &= (1 << 20) - 1;
    if (b >= nframes) return;
    uint8_t* area = clArea + (size_t)b * clFrameBytes;
    ClShared cl;
    cl.ctl = (ClCtl*)area;
    const size_t maxSubs = ((size_t)P.npx + CL_SUB - 1) / CL_SUB;
    cl.sub = (ClSub*)(area + 512);
    cl.specMap = (unsigned*)(area + 512 + ((maxSubs * sizeof(ClSub) + 511) & ~(size_t)511));
    cl.bigBm = cl.specMap + ((specWords + 127) & ~127);      // (zeroed with the shared map when the frame is too large for the LDS bitmap; empty otherwise)
    cl.rec = (ClRec*)(cl.bigBm + ((P.sw > TorusFrame::XMASK + 1 || P.sh > TorusFrame::YMASK + 1) ? TorusGlobal::WORDS : 0));
    cl.arena = (unsigned*)(cl.rec + maxSubs * CL_RES);
    cl.specShift = specShift; cl.specW = specShift >= 0 ? (P.sw + (1 << specShift) - 1) >> specShift : 0;
    cl.nHelpers = (nWG - 1) * CL_HPW; cl.window = window;
    unsigned* mine = dynLds + (size_t)wave * (CL_LIST + ClTorus::WORDS);      // (helper workgroups)
    if (role != 0 && wave < CL_HPW) for (int i = threadIdx.x & 63; i < ClTorus::WORDS; i += 64) mine[CL_LIST + i] = 0u;
    const bool bigFrame = P.sw > TorusFrame::XMASK + 1 || P.sh > TorusFrame::YMASK + 1;      // the main wave's bitmap lives in global memory (zeroed by the host)
    const int bmWords = bigFrame ? 0 : TorusFrame::WORDS;
    if (role == 0) {
        for (int i = threadIdx.x; i < bmWords; i += blockDim.x) dynLds[QCAP + 4 + i] = 0u;
        ClSlot* ring = (ClSlot*)(dynLds + QCAP + 4 + bmWords + CL_SCAN);
        if (threadIdx.x < CL_RING) { ring[threadIdx.x].chunk = -1; ring[threadIdx.x].ready = 0; }
        if (threadIdx.x == 0) { ClLocal* loc = (ClLocal*)(ring + CL_RING); loc->mainChunk = 0; loc->commitSeq = 0; loc->finished = 0; }
    }
    __syncthreads();
    if (role == 0) {
        // the main wave's workgroup: main wave + feeder.  No helpers here: their L1 invalidations cost the main wave 4 % (7.21 -> 6.89 ms)
        ClSlot* ring = (ClSlot*)(dynLds + QCAP + 4 + bmWords + CL_SCAN);
        ClLocal* loc = (ClLocal*)(ring + CL_RING);
        if (wave == 0) {
            if (bigFrame) cl_main<TorusGlobal, 0>(ws, P, b, dynLds, cl.bigBm, dynLds + QCAP + 4, red[0], stashes[0], cl, ring, loc);
            else cl_main<TorusFrame, 0>(ws, P, b, dynLds, dynLds + QCAP + 4, dynLds + QCAP + 4 + TorusFrame::WORDS, red[0], stashes[0], cl, ring, loc);
        }
        else if (wave == 1 && !nofeed) cl_feeder(ws, P, b, cl, ring, loc);
        return;
    }
    else if (wave < CL_HPW) cl_helper((role - 1) * CL_HPW + wave, ws, P, b, cl, mine, mine + CL_LIST, stashes[wave], red[wave]);
}

// The same with the rectangles streamed to a concurrent NFA stage (the default since round 5: cl_main<G, true>, lsd_nfa.h: k_nfa_stream).  A copy of the kernel above rather
// than a shared body: wrapped into a common inline function the default kernel came out with other register assignments, and the default kernel is the measured one.
__global__ __launch_bounds__(64 * CL_WAVES) void k_lsd_regions_cl_stream(uint8_t* __restrict__ ws, LsdPlan P, uint8_t* __restrict__ clArea, size_t clFrameBytes, int nframes, int nWG,
                                                                        int specWords, int specShift, int window) {
    extern __shared__ __align__(16) unsigned dynLds[];
    __shared__ double red[CL_WAVES][3 * 64];
    __shared__ float4 stashes[CL_WAVES][64];
    // blocks with the same (blockIdx % 8) share an XCD; an XCD hosts frames xcd, xcd + 8, xcd + 16, ... (nWG workgroups each)
    const int j = blockIdx.x >> 3, b = (blockIdx.x & 7) + 8 * (j / nWG), role = j % nWG, wave = threadIdx.x >> 6;
    const bool nofeed = (window & (1 << 20)) != 0;      // (experiment knob: no feeder wave)
    if (window >= 0) window &= (1 << 20) - 1;
    if (b >= nframes) return;
    uint8_t* area = clArea + (size_t)b * clFrameBytes;
    ClShared cl;
    cl.ctl = (ClCtl*)area;
    const size_t maxSubs = ((size_t)P.npx + CL_SUB - 1) / CL_SUB;
    cl.sub = (ClSub*)(area + 512);
    cl.specMap = (unsigned*)(area + 512 + ((maxSubs * sizeof(ClSub) + 511) & ~(size_t)511));
    cl.bigBm = cl.specMap + ((specWords + 127) & ~127);      // (zeroed with the shared map when the frame is too large for the LDS bitmap; empty otherwise)
    cl.rec = (ClRec*)(cl.bigBm + ((P.sw > TorusFrame::XMASK + 1 || P.sh > TorusFrame::YMASK + 1) ? TorusGlobal::WORDS : 0));
    cl.arena = (unsigned*)(cl.rec + maxSubs * CL_RES);
    cl.specShift = specShift; cl.specW = specShift >= 0 ? (P.sw + (1 << specShift) - 1) >> specShift : 0;
    cl.nHelpers = (nWG - 1) * CL_HPW; cl.window = window;
    unsigned* mine = dynLds + (size_t)wave * (CL_LIST + ClTorus::WORDS);      // (helper workgroups)
    if (role != 0 && wave < CL_HPW) for (int i = threadIdx.x & 63; i < ClTorus::WORDS; i += 64) mine[CL_LIST + i] = 0u;
    const bool bigFrame = P.sw > TorusFrame::XMASK + 1 || P.sh > TorusFrame::YMASK + 1;      // the main wave's bitmap lives in global memory (zeroed by the host)
    const int bmWords = bigFrame ? 0 : TorusFrame::WORDS;
    if (role == 0) {
        for (int i = threadIdx.x; i < bmWords; i += blockDim.x) dynLds[QCAP + 4 + i] = 0u;
        ClSlot* ring = (ClSlot*)(dynLds + QCAP + 4 + bmWords + CL_SCAN);
        if (threadIdx.x < CL_RING) { ring[threadIdx.x].chunk = -1; ring[threadIdx.x].ready = 0; }
        if (threadIdx.x == 0) { ClLocal* loc = (ClLocal*)(ring + CL_RING); loc->mainChunk = 0; loc->commitSeq = 0; loc->finished = 0; }
    }
    __syncthreads();
    if (role == 0) {
        // the main wave's workgroup: main wave + feeder.  No helpers here: their L1 invalidations cost the main wave 4 % (7.21 -> 6.89 ms)
        ClSlot* ring = (ClSlot*)(dynLds + QCAP + 4 + bmWords + CL_SCAN);
        ClLocal* loc = (ClLocal*)(ring + CL_RING);
        if (wave == 0) {
            if (bigFrame) cl_main<TorusGlobal, 1>(ws, P, b, dynLds, cl.bigBm, dynLds + QCAP + 4, red[0], stashes[0], cl, ring, loc);
            else cl_main<TorusFrame, 1>(ws, P, b, dynLds, dynLds + QCAP + 4, dynLds + QCAP + 4 + TorusFrame::WORDS, red[0], stashes[0], cl, ring, loc);
        }
        else if (wave == 1 && !nofeed) cl_feeder(ws, P, b, cl, ring, loc);
        return;
    }
    else if (wave < CL_HPW) cl_helper((role - 1) * CL_HPW + wave, ws, P, b, cl, mine, mine + CL_LIST, stashes[wave], red[wave]);
}

