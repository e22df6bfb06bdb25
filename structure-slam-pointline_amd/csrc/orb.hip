// ORB front-end for MI355X (gfx950): pyramid, per-cell FAST-9 + NMS + threshold
// fallback, quadtree distribution, intensity-centroid orientation, 7x7 blur and
// 256-bit rBRIEF — the device side of StructureSLAM::ORBextractor
// (reference src/ORBextractor.cc; behavioural spec in SURVEY.md Appendix D).
//
// Design:
//   * frames are batched: every launch covers all frames (and, where the stage has
//     no cross-level dependency, all pyramid levels) of the batch;
//   * pyramid levels are stored UNPADDED (the reference's 19-px reflect border is
//     never read by any consumer: keypoints live in [19,w-19) and the widest
//     consumer, rBRIEF, reaches 18 px);
//   * FAST scores are threshold independent (score = best 9-arc contrast - 1), so
//     one LDS-staged pass per 30x30-ish cell reproduces "FAST(20) then FAST(7) if
//     empty" including the cell-local NMS, without a score map in HBM;
//   * the quadtree is simulated exactly by one wave per (frame, level);
//   * orientation + blur + rBRIEF run fused, one wave per keypoint, on a 43x43 patch
//     staged in LDS (no blurred image in HBM).
#include "common.h"
#include <cmath>
#include <algorithm>

using namespace sslam;

namespace {

constexpr int MAX_LEVELS = 16;
constexpr int MAX_ROOTS = 64;            // quadtree root strips = round(W/H) of a level (src/ORBextractor.cc:543)
constexpr int EDGE = 19;          // EDGE_THRESHOLD, src/ORBextractor.cc:74
constexpr int MINB = 16;          // EDGE_THRESHOLD-3, :773
constexpr int HALF_PATCH = 15;

struct LevelInfo {
    int w, h, pitch;
    unsigned off;                 // byte offset of the level inside a frame's pyramid block
    int cellBeg, nCells;          // range in the frame's cell table
    int candOff, candCap;         // range in the frame's candidate arrays (u32 units)
    int nfeat;                    // mnFeaturesPerLevel
    int selOff, selCap;           // range in the frame's selected-keypoint array
    int nIni;                     // root nodes of the quadtree
    float hX;
    int W, H;                     // maxBorder-minBorder extents
    float scale;                  // mvScaleFactor
    float size;                   // float(int(31*scale))
    int tabX, tabY;               // entry (int16x4) offsets into the resize tables
    float rcpGroups;              // 1 / ((w + 3) / 4): k_resize's row index
};

struct Plan {
    int nlevels;
    int nCellsFrame, candFrame, selFrame;
    unsigned blurPack0, blurPack1;      // taps 0-3 and 4-6 of the 7x7 sigma-2 blur as bytes (q8)
    const uint4* blurToep;              // the horizontal pass of k_describe as matrix-core operands (build_blur_toeplitz)
    int blurTapSum;
    size_t pyrFrame;              // bytes
    // Level 0 of the pyramid is the caller's image (src/ORBextractor.cc:1107-1132 copies it into a padded buffer; the padding is reflect-101 addressing here).  When base,
    // pitch and frame stride are 4-byte aligned the consumers read it IN PLACE (img0 != nullptr: per call, set by sslam_orb_extract_batch_dev); otherwise k_copy_level0
    // copies it into the frame's pyramid block first (rounds 1-4 always did: 0.6 MB per frame read and written for nothing).
    const uint8_t* img0; size_t img0Stride; int img0Pitch;
    int maxCellW, maxCellH, maxCellsLevel, maxNodeCap;
    LevelInfo L[MAX_LEVELS];
};

struct CellInfo {
    short level, x0, y0, x1, y1;  // detection rectangle [x0,x1) x [y0,y1) in level coordinates
    short pad;
    int candOff;                  // offset inside the frame's candidate array
    unsigned magic, gmagic;       // 2^19 / cell width + 1 and 2^19 / (dword groups per row) + 1: k_fast_cells' exact divisions as one 24-bit multiply and a shift (i < 4400 < 2^19 / 64; the product stays below 2^32)
};

// base pointer and row pitch of level `level` of frame b
__device__ __forceinline__ const uint8_t* level_image(const Plan& P, const uint8_t* __restrict__ pyr, size_t pyrFrame, int b, int level, int& pitch) {
    if (level == 0 && P.img0) { pitch = P.img0Pitch; return P.img0 + (size_t)b * P.img0Stride; }
    pitch = P.L[level].pitch;
    return pyr + (size_t)b * pyrFrame + P.L[level].off;
}

__device__ const signed char kPat[1024] = {
#include "orb_pattern.inc"
};
__constant__ int kUmax[16];
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s16x2(unsigned v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ unsigned as_u32(s16x2 v) { return __builtin_bit_cast(unsigned, v); }
// (the 7 blur taps travel in the Plan, packed as bytes for v_dot4_u32_u8: they belong to the extractor -- sslam_orb_set_blur_variant -- not to the module)
__constant__ unsigned kDiscMask[31 * 8];  // byte masks of the r=15 disc: row v, dword m covers u = 4m-15 .. 4m-12

// ------------------------------------------------------------------ level 0 copy
__global__ void k_copy_level0(const uint8_t* __restrict__ in, size_t pitch, size_t imageStride,
                              uint8_t* __restrict__ pyr, size_t pyrFrame, int w, int h, int dpitch) {
    const int b = blockIdx.z;
    const int y = blockIdx.y;
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (x >= w) return;
    const uint8_t* s = in + (size_t)b * imageStride + (size_t)y * pitch + x;
    uint8_t* d = pyr + (size_t)b * pyrFrame + (size_t)y * dpitch + x;
    if (x + 16 <= w && (((uintptr_t)s) & 15) == 0) {
        *(uint4*)d = *(const uint4*)s;
    } else {
        int n = min(16, w - x);
        for (int i = 0; i < n; ++i) d[i] = s[i];
    }
}

// ------------------------------------------------------------------ bilinear /1.2
// cv::resize(INTER_LINEAR) for CV_8UC1: 11-bit fixed-point taps.
// tabX/tabY entries: {src offset, coef0, coef1, 0} as int16x4 (tabX padded to a multiple of 4 entries).
// One thread = 4 horizontally adjacent output pixels: two 16-byte table loads, then the two source rows as three aligned
// dwords each (sub-dword global loads cost the texture path four times a dword load), bytes picked with 64-bit shifts,
// one dword store.  A group whose taps span more than the 12 loaded bytes (scale factors above ~2) takes byte loads.
__device__ __forceinline__ unsigned pick2(unsigned d0, unsigned d1, unsigned d2, int o) {      // bytes o, o+1 of d0:d1:d2 (o <= 10)
    const unsigned long long w01 = (unsigned long long)d0 | ((unsigned long long)d1 << 32);
    const unsigned long long w12 = (unsigned long long)d1 | ((unsigned long long)d2 << 32);
    return (unsigned)((o < 4 ? w01 : w12) >> (8 * (o < 4 ? o : o - 4)));
}
// src / srcStride / srcPitch: the source level (a level of the pyramid block, or the caller's image for level 1: Plan::img0); srcGuard: the source has no spare bytes
// behind its LAST row (the caller's image: behind any other row lies the next one), so a group of the last source row whose three dwords would reach past the pitch takes
// the byte path.  (Guarding every row put one byte-path lane into almost every wave: 8.9 -> 9.6 ms per 12 288 frames, GPU call B.)
__global__ __launch_bounds__(256) void k_resize(const uint8_t* __restrict__ src, size_t srcStride, int srcPitch, int srcGuard, uint8_t* __restrict__ pyr, size_t pyrFrame,
                                                LevelInfo S, LevelInfo D, const short4* __restrict__ tabs, int nframes) {
    for (int b = blockIdx.y; b < nframes; b += gridDim.y) {      // gridDim.y may be smaller than the batch: the workgroups walk the frames, and what follows up to the loads is loop-invariant (the compiler hoists it)
    const int ngroups = (D.w + 3) >> 2;                 // flattened (row, 4-pixel group) index: full waves whatever the level width
    const int t = blockIdx.x * 256 + threadIdx.x;
    // t / ngroups without the ~35-instruction division: the quotient of the float product is off by at most one for t < 2^24
    int y = (int)((float)t * D.rcpGroups);
    { const int r = t - __mul24(y, ngroups); y += r >= ngroups ? 1 : (r < 0 ? -1 : 0); }
    const int x4 = (t - __mul24(y, ngroups)) * 4;
    if (y >= D.h) return;                               // (the same for every frame)
    const short4 ty = tabs[D.tabY + y];
    const uint4 ta = *(const uint4*)(tabs + D.tabX + x4), tb = *(const uint4*)(tabs + D.tabX + x4 + 2);
    const int sy0 = min(max((int)ty.x, 0), S.h - 1), sy1 = min(max((int)ty.x + 1, 0), S.h - 1);
    const uint8_t* sbase = src + (size_t)b * srcStride;
    const uint8_t* g0 = sbase + (size_t)sy0 * srcPitch;
    const uint8_t* g1 = sbase + (size_t)sy1 * srcPitch;
    const unsigned tw[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
    int sx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sx[i] = (short)(tw[2 * i] & 0xFFFF);
    const int a = sx[0] & ~3;
    unsigned out = 0;
    const int b0 = ty.y, b1 = ty.z;
    if (sx[3] - a <= 10 && sx[1] - sx[0] <= 2 && sx[3] - sx[2] <= 2 && !(srcGuard && a + 12 > srcPitch && sy1 >= S.h - 1)) {      // pyramid levels: pitch % 64 == 0 and the frame block has 16 spare bytes, the dwords are readable; the caller's image: inside the row
        // Two outputs at a time: their four source bytes per row lie inside one 8-byte window of the three loaded dwords (window start =
        // dword 0 or 1), so ONE v_perm per row fetches [left_i, right_i, left_i+1, right_i+1]; a second v_perm widens a pair to 16-bit
        // lanes for v_dot2_i32_i16 with the (a0, a1) coefficient pair taken from the table words with one v_alignbit.
        const unsigned* q0 = (const unsigned*)(g0 + a);
        const unsigned* q1 = (const unsigned*)(g1 + a);
        const unsigned u0 = q0[0], u1 = q0[1], u2 = q0[2], v0 = q1[0], v1 = q1[1], v2 = q1[2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int oA = sx[2 * pr] - a, oB = sx[2 * pr + 1] - a;
            const bool hi = oA >= 4;
            const int base = hi ? 4 : 0;
            const unsigned w = (unsigned)(oA - base) | ((unsigned)(oB - base) << 16);
            const unsigned sel = w * 0x0101u + 0x01000100u;                                  // bytes q, q+1 for both outputs
            const unsigned r0 = __builtin_amdgcn_perm(hi ? u2 : u1, hi ? u1 : u0, sel);
            const unsigned r1 = __builtin_amdgcn_perm(hi ? v2 : v1, hi ? v1 : v0, sel);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int i = 2 * pr + e;
                const unsigned wsel = e == 0 ? 0x0c010c00u : 0x0c030c02u;                    // (byte, 0, byte + 1, 0): two zero-extended 16-bit lanes
                const s16x2 p0 = as_s16x2(__builtin_amdgcn_perm(0u, r0, wsel)), p1 = as_s16x2(__builtin_amdgcn_perm(0u, r1, wsel));
                const s16x2 cf = as_s16x2(__builtin_amdgcn_alignbit(tw[2 * i + 1], tw[2 * i], 16));      // (a0, a1)
                const int h0 = __builtin_amdgcn_sdot2(p0, cf, 0, false), h1 = __builtin_amdgcn_sdot2(p1, cf, 0, false);
                const int v = ((__mul24(b0, h0 >> 4) >> 16) + (__mul24(b1, h1 >> 4) >> 16) + 2) >> 2;      // coefficients <= 2^11, row sums <= 2^15: v_mul_i32_i24 (v_mul_hi_i32 is quarter rate)
                out |= (unsigned)(v & 255) << (8 * i);
            }
        }
    } else {
        int p00[4], p01[4], p10[4], p11[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sx1 = min(sx[i] + 1, S.w - 1);
            p00[i] = g0[sx[i]]; p01[i] = g0[sx1]; p10[i] = g1[sx[i]]; p11[i] = g1[sx1];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int a0 = (short)(tw[2 * i] >> 16), a1 = (short)(tw[2 * i + 1] & 0xFFFF);
            const int r0 = p00[i] * a0 + p01[i] * a1;
            const int r1 = p10[i] * a0 + p11[i] * a1;
            const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
            out |= (unsigned)(v & 255) << (8 * i);
        }
    }
    *(unsigned*)(pyr + (size_t)b * pyrFrame + D.off + (size_t)y * D.pitch + x4) = out;   // pitch%64==0, pad columns are scratch
    }
}

// ------------------------------------------------------------------ FAST per cell
// FAST-9/16 score, threshold independent: (max over the 16 nine-pixel arcs of the
// arc's minimum signed contrast, both polarities) - 1   (cv cornerScore<16>, A.1).
// Returns 0 when the pixel is not a corner at `minTh`.
__device__ __forceinline__ int fast_score16(const uint8_t* c, int tp, int minTh) {
    const int v = c[0];
    int r[16];
    r[0] = c[3 * tp];      r[1] = c[3 * tp + 1];   r[2] = c[2 * tp + 2];   r[3] = c[tp + 3];
    r[4] = c[3];           r[5] = c[-tp + 3];      r[6] = c[-2 * tp + 2];  r[7] = c[-3 * tp + 1];
    r[8] = c[-3 * tp];     r[9] = c[-3 * tp - 1];  r[10] = c[-2 * tp - 2]; r[11] = c[-tp - 3];
    r[12] = c[-3];         r[13] = c[tp - 3];      r[14] = c[2 * tp - 2];  r[15] = c[3 * tp - 1];
    // necessary condition for a 9-arc at minTh: every opposite pair holds one arc member
    bool cb = true, cd = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        cb = cb && ((r[k] > v + minTh) || (r[k + 8] > v + minTh));
        cd = cd && ((r[k] < v - minTh) || (r[k + 8] < v - minTh));
    }
    if (!(cb || cd)) return 0;
    // One polarity per evaluation.  A 9-arc whose pixels are all darker than v - t and one whose pixels are all brighter than v + t would
    // have to share two ring positions, so at most one polarity can reach a score >= minTh >= 1: the lane picks the polarity its pair
    // test left open (d = v - r for a darker ring, r - v for a brighter one) and runs ONE max-of-arc-minimum network instead of the two
    // of cv::cornerScore; the rare pixel that passed both pair tests runs the network a second time with the other sign.
    // d = signed contrast in the chosen polarity; the minimum over a 9-arc is a min3 of three min3's (v_min3_i32), the maximum over the sixteen arcs a max3 tree.
    // brighter ring: d = r - v.  Darker ring: ~(r - v) = (v - r) - 1 -- a bitwise NOT reverses the order like a negation does, costs one xor per element instead of a
    // negate + select, and the constant 1 comes back after the network.  The network works IN PLACE: m3[k] = min(d[k], d[k+1], d[k+2]) overwrites d[k] (d[0], d[1] kept
    // for the wrap-around), m9[k] = min(m3[k], m3[k+3], m3[k+6]) goes straight into the running maximum -- sixteen live values instead of forty-eight; and the ring itself
    // dies with the first evaluation (the rare second one reads it again): once the staging loop no longer set the kernel's register count the network did, and that count
    // decides how many of the kernel's waves fit beside the sequential LSD core in its guest form.
    auto network = [&](int (&d)[16], int sg) -> int {
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k] = (d[k] - v) ^ sg;
        const int d0 = d[0], d1 = d[1];
#pragma unroll
        for (int k = 0; k < 14; ++k) d[k] = min(min(d[k], d[k + 1]), d[k + 2]);
        d[14] = min(min(d[14], d[15]), d0);
        d[15] = min(min(d[15], d0), d1);
        int A = min(min(d[0], d[3]), d[6]);
#pragma unroll
        for (int k = 1; k < 16; ++k) A = max(A, min(min(d[k], d[(k + 3) & 15]), d[(k + 6) & 15]));
        return A - sg;                                       // + 1 for the darker polarity (see above)
    };
    int best = max(0, network(r, cd ? -1 : 0));             // the darker ring if that test passed, else the brighter one
    const bool both = cb && cd;
    if (__builtin_amdgcn_ballot_w64(both)) {                // rare: the brighter ring as well
        const uint8_t* c2 = c;
        asm volatile("" : "+v"(c2));                        // (read again: the first evaluation consumed the ring in place)
        int q[16];
        q[0] = c2[3 * tp];      q[1] = c2[3 * tp + 1];   q[2] = c2[2 * tp + 2];   q[3] = c2[tp + 3];
        q[4] = c2[3];           q[5] = c2[-tp + 3];      q[6] = c2[-2 * tp + 2];  q[7] = c2[-3 * tp + 1];
        q[8] = c2[-3 * tp];     q[9] = c2[-3 * tp - 1];  q[10] = c2[-2 * tp - 2]; q[11] = c2[-tp - 3];
        q[12] = c2[-3];         q[13] = c2[tp - 3];      q[14] = c2[2 * tp - 2];  q[15] = c2[3 * tp - 1];
        const int b2 = network(q, 0);
        if (both) best = max(best, b2);
    }
    int s = best - 1;
    return s >= minTh ? s : 0;
}

// One wave per (cell, frame).  LDS: image tile with 3-px halo (staged with aligned dword loads), score
// tile with a zero rim (NMS neighbours outside the cell's detection rectangle count as 0, D.1), and a
// keep-code per pixel.  Candidates leave in raster order, packed (score<<24 | y<<12 | x) with x,y
// relative to minBorder (=16), :820-825.  i -> (row, col) uses an exact magic-multiply division.
__global__ __launch_bounds__(64) void k_fast_cells(const uint8_t* __restrict__ pyr, size_t pyrFrame, Plan P,
                                                   const CellInfo* __restrict__ cells,
                                                   unsigned* __restrict__ cand, int* __restrict__ cellCount,
                                                   int iniTh, int minTh, int tileP, int scP) {
    extern __shared__ __align__(16) uint8_t lds[];
    // Round 4: which cell a workgroup takes follows the dispatch order.  Workgroups are dealt round-robin over the eight XCDs, so with
    // cellId = blockIdx.x horizontally adjacent cells -- which share their 6-pixel overlap and the 128-byte lines both tiles straddle -- were
    // fetched through eight different L2s (4.34 MB fetched per frame for 0.95 MB of pyramid, PMC).  Now the workgroups of one XCD
    // (blockIdx.x % 8) walk one contiguous eighth of the frame's cell list, in order; which eighth rotates with the frame so that
    // every XCD sees every pyramid level (the levels differ in corner density).  gridDim.x = 8 * ceil(nCells / 8).
    const int b = blockIdx.y, per = gridDim.x >> 3;
    const int cellId = (((blockIdx.x & 7) + b) & 7) * per + (blockIdx.x >> 3);
    if (cellId >= P.nCellsFrame) return;
    const int lane = threadIdx.x;
    const CellInfo ci = cells[cellId];
    const LevelInfo& L = P.L[ci.level];
    const int cw = ci.x1 - ci.x0, ch = ci.y1 - ci.y0;
    uint8_t* tile = lds;                                       // (maxCellH+6) x tileP, tileP % 4 == 0
    uint8_t* sc = tile + (P.maxCellH + 6) * tileP;             // (maxCellH+2) x scP
    uint8_t* code = sc + (P.maxCellH + 2) * scP;               // maxCellH x maxCellW
    int pitch;
    const uint8_t* img = level_image(P, pyr, pyrFrame, b, ci.level, pitch);
    // stage rows [y0-3, y1+3) from the 4-byte aligned column ax <= x0-3; cells are interior (x0 >= 19, x1 + 3 <= w - 13) and the
    // row pitch is a multiple of 4, so the aligned dwords stay inside the row
    const int ax = (ci.x0 - 3) & ~3, off = (ci.x0 - 3) - ax;
    const int ndw = (off + cw + 6 + 3) >> 2, th = ch + 6;
    {   // buffer loads: the tile's first byte in a scalar resource descriptor, ONE 32-bit offset register per load (the loop's sixteen loads in flight with 64-bit addresses
        // set the kernel's register count: 51 -> how many of its waves fit beside the sequential LSD core in its guest form)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(img + (size_t)(ci.y0 - 3) * pitch + ax), 0, 0x7FFFFFFF, 0x00020000);
        const int q0 = lane & 15;
        for (int ry = lane >> 4; ry < th; ry += 4) {
            const int rowOff = __mul24(ry, pitch);
            for (int q = q0; q < ndw; q += 16) ((unsigned*)(tile + ry * tileP))[q] = __builtin_amdgcn_raw_buffer_load_b32(rs, rowOff + 4 * q, 0, 0);
        }
    }
    for (int i = lane; i < (((ch + 2) * scP + 3) >> 2); i += 64) ((unsigned*)sc)[i] = 0;
    __syncthreads();
    const unsigned magic = ci.magic;      // (1 << 19) / cw + 1: floor(i/cw) == (i*magic)>>19 for i < 4400, cw <= 64
    // pass A: necessary condition for a 9-arc at minTh, four pixels per lane on whole dwords.  Nine contiguous ring positions
    // always contain two ADJACENT compass points (ring 0/4/8/12 = S/E/N/W at distance 3), i.e. one vertical and one
    // horizontal one, and both must lie on the arc's side of the threshold:
    //   ((S|N brighter) & (E|W brighter)) | ((S|N darker) & (E|W darker)).
    // Bytes are widened to two 16-bit halves per dword and compared with packed i16 arithmetic (sign bit of th -/+ (r - v)).
    // Survivors are compacted in raster order so that the expensive ring evaluation of pass B runs with full lanes on the few
    // pixels that can matter.
    unsigned short* clist = (unsigned short*)(lds + (((P.maxCellH + 6) * tileP + (P.maxCellH + 2) * scP + P.maxCellH * P.maxCellW + 7) & ~3));
    int ncl = 0;
    {
        const int T0 = off + 3;                                    // tile column of pixel 0
        const int gmin = T0 >> 2, ng = ((T0 + cw - 1) >> 2) - gmin + 1;
        const unsigned gmagic = ci.gmagic;                          // (1 << 19) / ng + 1: ng <= 17, items < 4400
        const int nitems = ch * ng;
        const s16x2 th2 = {(short)minTh, (short)minTh};
        for (int it0 = 0; it0 < nitems; it0 += 64) {
            const int it = it0 + lane;
            unsigned flags = 0;
            int py = 0, g = 0;
            if (it < nitems) {
                py = (int)(__umul24((unsigned)it, gmagic) >> 19);
                g = gmin + (it - __mul24(py, ng));
                const int rowN = __mul24(py, tileP);               // (24-bit multiplies throughout: v_mul_lo_u32 is quarter rate)
                const unsigned* rc = (const unsigned*)(tile + rowN + 3 * tileP) + g;
                const unsigned c1 = rc[0], c0 = rc[-1], c2 = rc[1];
                const unsigned n4 = ((const unsigned*)(tile + rowN))[g], s4 = ((const unsigned*)(tile + rowN + 6 * tileP))[g];
                unsigned res = 0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {                      // h = 0: bytes 0,2   h = 1: bytes 1,3
                    // bytes h and h + 2 of a dword as two zero-extended 16-bit lanes: one v_perm each; the east / west neighbours (three bytes to the right / left) come
                    // straight out of the dword pairs (c2:c1 bytes 3 + h, 5 + h; c1:c0 bytes 1 + h, 3 + h) -- no v_alignbyte in between
                    const unsigned wsel = h == 0 ? 0x0c020c00u : 0x0c030c01u;
                    const s16x2 v = as_s16x2(__builtin_amdgcn_perm(0u, c1, wsel));
                    const s16x2 rn = as_s16x2(__builtin_amdgcn_perm(0u, n4, wsel)), rs = as_s16x2(__builtin_amdgcn_perm(0u, s4, wsel));
                    const s16x2 re = as_s16x2(__builtin_amdgcn_perm(c2, c1, h == 0 ? 0x0c050c03u : 0x0c060c04u)), rw = as_s16x2(__builtin_amdgcn_perm(c1, c0, h == 0 ? 0x0c030c01u : 0x0c040c02u));
                    // (N or S brighter than v + t) and (E or W brighter) == min(max(N, S), max(E, W)) > v + t; darker: max(min(N, S), min(E, W)) < v - t: packed max / min,
                    // then one packed subtraction per polarity whose sign bit is the answer
                    const s16x2 vp = v + th2, vm = v - th2;
                    const s16x2 mxV = __builtin_elementwise_max(rn, rs), mnV = __builtin_elementwise_min(rn, rs);
                    const s16x2 mxH = __builtin_elementwise_max(re, rw), mnH = __builtin_elementwise_min(re, rw);
                    const unsigned bright = as_u32(vp - __builtin_elementwise_min(mxV, mxH));      // sign set <=> max > v + t on both axes
                    const unsigned dark = as_u32(__builtin_elementwise_max(mnV, mnH) - vm);        // sign set <=> min < v - t on both axes
                    const unsigned m = (bright | dark) & 0x80008000u;
                    res |= m >> (15 - h);                          // bit 0+h from byte h, bit 16+h from byte 2+h
                }
                // res bits: 0 -> byte 0, 1 -> byte 1, 16 -> byte 2, 17 -> byte 3
                flags = (res & 3u) | ((res >> 14) & 12u);
                const int px0 = 4 * g - T0;                        // pixel of byte 0
                flags &= (0xFu << max(0, -px0)) & (0xFu >> max(0, px0 + 4 - cw));      // bytes that are pixels of this cell
            }
            const int cntf = __popc(flags);
            const int incl = wave_incl_scan(cntf);
            int pos = ncl + incl - cntf;
            const int ibase = __mul24(py, cw) + 4 * g - T0;
#pragma unroll
            for (int k = 0; k < 4; ++k) if (flags & (1u << k)) clist[pos++] = (unsigned short)(ibase + k);
            ncl += __builtin_amdgcn_readlane(incl, 63);
        }
    }
    __syncthreads();
    for (int j = lane; j < ncl; j += 64) {
        const int i = clist[j];
        const int py = (int)(__umul24((unsigned)i, magic) >> 19), px = i - __mul24(py, cw);
        const int s = fast_score16(tile + __mul24(py + 3, tileP) + (off + px + 3), tileP, minTh);
        if (s) sc[__mul24(py + 1, scP) + (px + 1)] = (uint8_t)s;        // sc is pre-zeroed
    }
    __syncthreads();
    // NMS + thresholds + emission only visit the compacted list (it is in raster order, and so is what it emits)
    int cnt20 = 0;
    for (int j0 = 0; j0 < ncl; j0 += 64) {
        const int j = j0 + lane;
        int cde = 0;
        if (j < ncl) {
            const int i = clist[j];
            const int py = (int)(__umul24((unsigned)i, magic) >> 19), px = i - __mul24(py, cw);
            const uint8_t* c = sc + __mul24(py + 1, scP) + (px + 1);
            // strict maximum of the 3 x 3 neighbourhood: the nine bytes are read together (one LDS round trip; as a chain of && the compiler made it eight dependent ones,
            // a branch and a wait per neighbour) and "s > every neighbour" is "s > their maximum"
            const int s = c[0];
            const int n0 = c[-1], n1 = c[1], n2 = c[-scP - 1], n3 = c[-scP], n4 = c[-scP + 1], n5 = c[scP - 1], n6 = c[scP], n7 = c[scP + 1];
            const int mx = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
            cde = (s > mx) ? (s >= iniTh ? 2 : 1) : 0;      // (s > mx >= 0 implies s > 0)
            code[j] = (uint8_t)cde;
        }
        cnt20 += __popcll(__ballot(cde == 2));
    }
    __syncthreads();
    const int need = cnt20 > 0 ? 2 : 1;      // FAST(iniTh) empty -> FAST(minTh), :809-816
    unsigned* out = cand + (size_t)b * P.candFrame + ci.candOff;
    int n = 0;
    for (int j0 = 0; j0 < ncl; j0 += 64) {
        const int j = j0 + lane;
        bool keep = false;
        if (j < ncl) keep = code[j] >= need;
        unsigned long long m = __ballot(keep);
        if (keep) {
            const int i = clist[j];
            const int py = (int)(__umul24((unsigned)i, magic) >> 19), px = i - __mul24(py, cw);
            int s = sc[__mul24(py + 1, scP) + (px + 1)];
            out[n + mbcnt(m)] = ((unsigned)s << 24) | ((unsigned)(ci.y0 + py - MINB) << 12) | (unsigned)(ci.x0 + px - MINB);
        }
        n += __popcll(m);
    }
    if (lane == 0) cellCount[(size_t)b * P.nCellsFrame + cellId] = n;
}

// ------------------------------------------------------------------ quadtree
// Exact simulation of ORBextractor::DistributeOctTree (src/ORBextractor.cc:539-763,
// spec SURVEY D.2) by one wave per (level, frame).  Keypoints live in two global
// ping-pong arrays; a node owns a contiguous segment, and a split is a stable
// 4-way partition of that segment (ballot + prefix popcount).  The node list is a
// doubly linked list in LDS; tie-break D1 = creation sequence.
struct OctLds {
    int* pre;                 // cell prefix sums            [maxCellsLevel+1]
    unsigned* box0;           // x0 | x1<<16                 [NC]
    unsigned* box1;           // y0 | y1<<16
    unsigned* start;          // segment start, bit31 = lives in buffer B
    int* count;
    unsigned* links;          // next | prev<<16 (0xFFFF = nil)
    unsigned long long* expA; // expandable lists (size<<40 | seq<<16 | node)  [NCp2]
    unsigned long long* expB;
};

#define NIL 0xFFFFu

__global__ __launch_bounds__(64) void k_octree(const unsigned* __restrict__ cand, const int* __restrict__ cellCount,
                                               const CellInfo* __restrict__ cells, unsigned* __restrict__ bufA,
                                               unsigned* __restrict__ bufB, unsigned* __restrict__ sel,
                                               int* __restrict__ selCount, Plan P, int NC, int NCp2) {
    extern __shared__ __align__(16) uint8_t lds[];
    // linear workgroup id = frame * nlevels + x: with eight levels the XCD would be the level; rotate by the frame instead
    // (digit-sum rotation: XCDs, CUs and SIMDs are all dealt out round-robin, a plain +frame would still be periodic per CU)
    const int b = blockIdx.y, level = (int)((blockIdx.x + b + (b >> 3) + (b >> 6) + (b >> 9)) % gridDim.x), lane = threadIdx.x;
    const LevelInfo& L = P.L[level];
    OctLds S;
    S.expA = (unsigned long long*)lds;
    S.expB = S.expA + NCp2;
    S.pre = (int*)(S.expB + NCp2);
    S.box0 = (unsigned*)(S.pre + P.maxCellsLevel + 1);
    S.box1 = S.box0 + NC;
    S.start = S.box1 + NC;
    S.count = (int*)(S.start + NC);
    S.links = (unsigned*)(S.count + NC);

    const int nc = L.nCells;
    const int* cc = cellCount + (size_t)b * P.nCellsFrame + L.cellBeg;
    int total = 0;
    for (int c0 = 0; c0 < nc; c0 += 64) {
        int c = c0 + lane;
        int cnt = c < nc ? cc[c] : 0;
        int inc = wave_incl_scan(cnt);
        if (c < nc) S.pre[c] = total + inc - cnt;
        total += __shfl(inc, 63, 64);
    }
    if (lane == 0) S.pre[nc] = total;
    __syncthreads();
    int* selCnt = selCount + (size_t)b * P.nlevels + level;
    if (total == 0 || L.nIni <= 0) { if (lane == 0) *selCnt = 0; return; }      // build_plan rejects aspect ratios beyond MAX_ROOTS

    unsigned* A = bufA + (size_t)b * P.candFrame + L.candOff;
    unsigned* B = bufB + (size_t)b * P.candFrame + L.candOff;
    const unsigned* cbase = cand + (size_t)b * P.candFrame;
    // gather the cells' candidates in cell-row-major order (the order FAST emitted them, :789-829)
    for (int p = lane; p < total; p += 64) {
        int lo = 0, hi = nc;                       // find c: pre[c] <= p < pre[c+1]
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (S.pre[mid] <= p) lo = mid; else hi = mid; }
        A[p] = cbase[cells[L.cellBeg + lo].candOff + (p - S.pre[lo])];
    }
    __syncthreads();

    // uniform bookkeeping (every lane holds the same values)
    unsigned head = NIL, tail = NIL;
    int nNodes = 0, nAlloc = 0, seq = 0;

    auto push_front = [&](unsigned slot) {
        if (lane == 0) {
            S.links[slot] = head | (NIL << 16);
            if (head != NIL) S.links[head] = (S.links[head] & 0xFFFFu) | (slot << 16);
        }
        if (head == NIL) tail = slot;
        head = slot;
        ++nNodes;
    };
    auto push_back = [&](unsigned slot) {
        if (lane == 0) {
            S.links[slot] = NIL | (tail << 16);
            if (tail != NIL) S.links[tail] = (S.links[tail] & 0xFFFF0000u) | slot;
        }
        if (tail == NIL) head = slot;
        tail = slot;
        ++nNodes;
    };

    // roots (:545-584): nIni = round(W/H) vertical strips; kp -> strip int(x/hX)
    {
        int src_is_B = 0;
        if (L.nIni > 1) {
            int o = 0;
            for (int r = 0; r < L.nIni; ++r) {
                int begin = o;
                for (int i0 = 0; i0 < total; i0 += 64) {
                    int i = i0 + lane;
                    unsigned v = i < total ? A[i] : 0;
                    bool mine = i < total && (int)__fdiv_rn((float)(v & 0xFFF), L.hX) == r;
                    unsigned long long m = __ballot(mine);
                    if (mine) B[o + mbcnt(m)] = v;
                    o += __popcll(m);
                }
                if (o > begin) {
                    unsigned slot = nAlloc++;
                    if (lane == 0) {
                        S.box0[slot] = (unsigned)(int)__fmul_rn(L.hX, (float)r) | ((unsigned)(int)__fmul_rn(L.hX, (float)(r + 1)) << 16);
                        S.box1[slot] = 0u | ((unsigned)L.H << 16);
                        S.start[slot] = (unsigned)begin | 0x80000000u;
                        S.count[slot] = o - begin;
                    }
                    __syncthreads();
                    push_back(slot);
                    ++seq;
                }
            }
            src_is_B = 1;
        } else {
            unsigned slot = nAlloc++;
            if (lane == 0) {
                S.box0[slot] = 0u | ((unsigned)(int)L.hX << 16);
                S.box1[slot] = 0u | ((unsigned)L.H << 16);
                S.start[slot] = 0u;
                S.count[slot] = total;
            }
            push_back(slot);
            ++seq;
        }
        (void)src_is_B;
        __syncthreads();
    }

    unsigned long long* expCur = S.expA;
    unsigned long long* expPrev = S.expB;
    int nExp = 0, nToExpand = 0;

    // DivideNode (:481-537) + child insertion (:615-660): stable partition of the
    // node's segment into UL, UR, BL, BR; non-empty children pushed to the FRONT in
    // that order; children with >1 keypoints recorded as expandable.
    auto split = [&](unsigned it) {
        const unsigned b0 = S.box0[it], b1 = S.box1[it], st = S.start[it];
        const int n = S.count[it];
        const unsigned lk = S.links[it];
        const int x0 = b0 & 0xFFFF, x1 = b0 >> 16, y0 = b1 & 0xFFFF, y1 = b1 >> 16;
        const int midx = x0 + ((x1 - x0 + 1) >> 1), midy = y0 + ((y1 - y0 + 1) >> 1);
        const bool inB = st >> 31;
        const int s = st & 0x7FFFFFFF;
        const unsigned* src = (inB ? B : A) + s;
        unsigned* dst = (inB ? A : B) + s;
        int c[4] = {0, 0, 0, 0};
        if (n <= 64) {
            unsigned v = lane < n ? src[lane] : 0;
            int cls = lane < n ? (((int)(v & 0xFFF) >= midx) | (((int)((v >> 12) & 0xFFF) >= midy) << 1)) : 4;
            unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1), m2 = __ballot(cls == 2), m3 = __ballot(cls == 3);
            c[0] = __popcll(m0); c[1] = __popcll(m1); c[2] = __popcll(m2); c[3] = __popcll(m3);
            if (cls < 4) {
                int pos = cls == 0 ? mbcnt(m0) : cls == 1 ? c[0] + mbcnt(m1) : cls == 2 ? c[0] + c[1] + mbcnt(m2) : c[0] + c[1] + c[2] + mbcnt(m3);
                dst[pos] = v;
            }
        } else {
            for (int i0 = 0; i0 < n; i0 += 64) {
                int i = i0 + lane;
                unsigned v = i < n ? src[i] : 0;
                int cls = i < n ? (((int)(v & 0xFFF) >= midx) | (((int)((v >> 12) & 0xFFF) >= midy) << 1)) : 4;
                c[0] += __popcll(__ballot(cls == 0)); c[1] += __popcll(__ballot(cls == 1));
                c[2] += __popcll(__ballot(cls == 2)); c[3] += __popcll(__ballot(cls == 3));
            }
            int o[4] = {0, c[0], c[0] + c[1], c[0] + c[1] + c[2]};
            for (int i0 = 0; i0 < n; i0 += 64) {
                int i = i0 + lane;
                unsigned v = i < n ? src[i] : 0;
                int cls = i < n ? (((int)(v & 0xFFF) >= midx) | (((int)((v >> 12) & 0xFFF) >= midy) << 1)) : 4;
                unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1), m2 = __ballot(cls == 2), m3 = __ballot(cls == 3);
                if (cls < 4) {
                    int pos = cls == 0 ? o[0] + mbcnt(m0) : cls == 1 ? o[1] + mbcnt(m1) : cls == 2 ? o[2] + mbcnt(m2) : o[3] + mbcnt(m3);
                    dst[pos] = v;
                }
                o[0] += __popcll(m0); o[1] += __popcll(m1); o[2] += __popcll(m2); o[3] += __popcll(m3);
            }
        }
        // unlink the parent
        {
            unsigned nx = lk & 0xFFFF, pv = lk >> 16;
            if (lane == 0) {
                if (pv != NIL) S.links[pv] = (S.links[pv] & 0xFFFF0000u) | nx;
                if (nx != NIL) S.links[nx] = (S.links[nx] & 0xFFFFu) | (pv << 16);
            }
            if (pv == NIL) head = nx;
            if (nx == NIL) tail = pv;
            --nNodes;
        }
        __syncthreads();
        const int cx0[4] = {x0, midx, x0, midx}, cx1[4] = {midx, x1, midx, x1};
        const int cy0[4] = {y0, y0, midy, midy}, cy1[4] = {midy, midy, y1, y1};
        int off = 0;
        bool reuse = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (c[k] > 0) {
                unsigned slot = reuse ? it : (unsigned)nAlloc++;
                reuse = false;
                if (lane == 0) {
                    S.box0[slot] = (unsigned)cx0[k] | ((unsigned)cx1[k] << 16);
                    S.box1[slot] = (unsigned)cy0[k] | ((unsigned)cy1[k] << 16);
                    S.start[slot] = (unsigned)(s + off) | (inB ? 0u : 0x80000000u);
                    S.count[slot] = c[k];
                }
                __syncthreads();
                push_front(slot);
                if (c[k] > 1) {
                    ++nToExpand;
                    if (lane == 0) expCur[nExp] = ((unsigned long long)c[k] << 40) | ((unsigned long long)seq << 16) | slot;
                    ++nExp;
                }
                ++seq;
            }
            off += c[k];
        }
        __syncthreads();
    };

    const int N = L.nfeat;
    bool finish = false;
    while (!finish) {
        const int prevSize = nNodes;
        nToExpand = 0; nExp = 0;
        unsigned it = head;
        while (it != NIL) {
            unsigned nx = S.links[it] & 0xFFFF;
            if (S.count[it] > 1) split(it);
            it = nx;
        }
        if (nNodes >= N || nNodes == prevSize) finish = true;
        else if (nNodes + nToExpand * 3 > N) {
            while (!finish) {
                const int prevSize2 = nNodes;
                // sort the expandable nodes ascending by (size, creation seq) — :684, tie-break D1
                int P2 = 1; while (P2 < nExp) P2 <<= 1;
                for (int i = nExp + lane; i < P2; i += 64) expCur[i] = ~0ull;
                __syncthreads();
                for (int k = 2; k <= P2; k <<= 1)
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        for (int i = lane; i < P2; i += 64) {
                            int ixj = i ^ j;
                            if (ixj > i) {
                                unsigned long long a = expCur[i], bb = expCur[ixj];
                                bool up = (i & k) == 0;
                                if ((a > bb) == up) { expCur[i] = bb; expCur[ixj] = a; }
                            }
                        }
                        __syncthreads();
                    }
                unsigned long long* t = expCur; expCur = expPrev; expPrev = t;
                const int nPrev = nExp;
                nExp = 0;
                for (int j = nPrev - 1; j >= 0; --j) {
                    split((unsigned)(expPrev[j] & 0xFFFF));
                    if (nNodes >= N) break;
                }
                if (nNodes >= N || nNodes == prevSize2) finish = true;
            }
        }
    }

    // retain the best-response keypoint of each node, list order (:742-760)
    unsigned* order = (unsigned*)S.expA;
    if (lane == 0) { int k = 0; for (unsigned it = head; it != NIL; it = S.links[it] & 0xFFFF) order[k++] = it; }
    __syncthreads();
    unsigned* out = sel + (size_t)b * P.selFrame + L.selOff;
    const int nOut = min(nNodes, L.selCap);
    for (int k = lane; k < nOut; k += 64) {
        unsigned node = order[k];
        unsigned st = S.start[node];
        const unsigned* src = ((st >> 31) ? B : A) + (st & 0x7FFFFFFF);
        int n = S.count[node];
        unsigned best = src[0];
        for (int i = 1; i < n; ++i) { unsigned v = src[i]; if ((v >> 24) > (best >> 24)) best = v; }
        out[k] = best;
    }
    if (lane == 0) *selCnt = nOut;
}

// ------------------------------------------------------------------ describe
// One wave per selected keypoint: IC_Angle (:77-104) on the unblurred level,
// GaussianBlur 7x7 sigma 2 (:1086, fixed-point D6) evaluated only where rBRIEF
// taps land, computeOrbDescriptor (:108-147), and the KeyPoint record (:837-847,
// :1095-1101).  The 43x43 source patch is staged in LDS with reflect-101.
constexpr int PR = 21;               // patch radius: 18 (taps) + 3 (blur)
constexpr int PW = 2 * PR + 1;       // 43
constexpr int PP = 48;               // LDS pitch: 12 aligned dwords per row

// TAP = true is the stage tap of the blur (a7, sslam_orb_debug_blur_patches): the same staging and the same two blur passes, but instead
// of the 512 rBRIEF taps every position of the 37x37 window |dx|,|dy| <= 18 is evaluated and written to descOut (1369 bytes per keypoint).
constexpr int TAPW = 37;
#ifndef SSLAM_DESCRIBE_MFMA
#define SSLAM_DESCRIBE_MFMA 1
#endif
template <bool TAP>
__global__ __launch_bounds__(64) void k_describe(const uint8_t* __restrict__ pyr, size_t pyrFrame, Plan P,
                                                 const unsigned* __restrict__ sel, const int* __restrict__ selCount,
                                                 sslam_keypoint* __restrict__ kpOut, uint8_t* __restrict__ descOut,
                                                 int* __restrict__ counts, int cap) {
    __shared__ __align__(16) uint8_t patch[PW * PP + 16];
    __shared__ __align__(16) unsigned short hb[PW * 40];
    const int b = blockIdx.y, lane = threadIdx.x;
    int slot = blockIdx.x;
    // locate (level, index) of this slot
    int level = 0;
    while (level < P.nlevels - 1 && slot >= P.L[level].selOff + P.L[level].selCap) ++level;
    const LevelInfo& L = P.L[level];
    const int idx = slot - L.selOff;
    const int* sc = selCount + (size_t)b * P.nlevels;
    int before = 0, totalKp = 0;
    for (int l = 0; l < P.nlevels; ++l) { int c = sc[l]; if (l < level) before += c; totalKp += c; }
    if (slot == 0 && lane == 0) counts[b] = min(totalKp, cap);
    if (idx >= sc[level]) return;
    const int outIdx = before + idx;
    if (outIdx >= cap) return;
    const unsigned pk = sel[(size_t)b * P.selFrame + slot];
    const int kx = (int)(pk & 0xFFF) + MINB, ky = (int)((pk >> 12) & 0xFFF) + MINB, score = pk >> 24;
    int pitch;
    const uint8_t* img = level_image(P, pyr, pyrFrame, b, level, pitch);
    // stage the 43x43 patch so that patch column c sits at LDS byte c + 2 of its row: the disc of IC_Angle (columns 6..36)
    // and the four-output groups of the horizontal blur then start on dword boundaries and can be consumed as whole dwords
    // (v_dot4_u32_u8 does four multiply-adds per instruction).  Interior keypoints (almost all): 12 aligned global dwords per
    // row, shifted into place with v_alignbyte against the neighbouring lane's dword (DPP); five rows per pass so that a row
    // never straddles the wave.  Keypoints within 21 px of the level border: byte loads with reflect-101.
    constexpr int SH = 2;
    const int ax = (kx - PR) & ~3;
    const bool interior = kx - PR >= 0 && ky - PR >= 0 && ky + PR < L.h && kx + PR < L.w && ax + 48 <= pitch;
    if (interior) {
        const int delta = (kx - PR) - ax - SH;                     // LDS byte b of a row = global byte ax + delta + b
        const uint8_t* src = img + (size_t)(ky - PR) * pitch + ax;
        const int rr = lane / 12, q = lane - rr * 12;
        // all nine loads first (rows past the patch re-read its last row: no branch around a load, so none of them waits for the one before --
        // as a loop of load / shift / store the kernel spent nine dependent memory round trips per keypoint here), then the shifts and LDS stores
        unsigned gl[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) gl[k] = *(const unsigned*)(src + (__umul24((unsigned)min(5 * k + rr, PW - 1), (unsigned)pitch) + 4u * (unsigned)min(q, 11)));      // 32-bit offset from the wave's base (a 64-bit row product is three quarter-rate multiplies)
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int r = 5 * k + rr;
            const bool on = lane < 60 && r < PW;
            const unsigned g = gl[k];
            const unsigned nxt = (unsigned)__builtin_amdgcn_mov_dpp((int)g, 0x130, 0xF, 0xF, false);      // wave_shl:1 = lane + 1
            const unsigned prv = (unsigned)__builtin_amdgcn_mov_dpp((int)g, 0x138, 0xF, 0xF, false);      // wave_shr:1 = lane - 1
            unsigned word;
            if (delta >= 0) word = __builtin_amdgcn_alignbyte(q < 11 ? nxt : 0u, g, (unsigned)delta);
            else word = __builtin_amdgcn_alignbyte(g, q > 0 ? prv : 0u, (unsigned)(delta + 4));
            if (on) ((unsigned*)(patch + r * PP))[q] = word;
        }
    } else {
        for (int i = lane; i < PW * PW; i += 64) {
            int r = i / PW, c = i - r * PW;
            int yy = reflect101(ky - PR + r, L.h), xx = reflect101(kx - PR + c, L.w);
            patch[r * PP + SH + c] = img[(size_t)yy * pitch + xx];
        }
    }
    __syncthreads();
    // intensity centroid over the r=15 disc: lane = (row v, dword m); weights u + 16 keep the dot product unsigned
    int m10 = 0, m01 = 0;
    for (int i = lane; i < 31 * 8; i += 64) {
        const int r = i >> 3, m = i & 7;
        const unsigned I4 = ((const unsigned*)(patch + (PR - HALF_PATCH + r) * PP + 8))[m] & kDiscMask[i];
        const unsigned w0 = (unsigned)(4 * m + 1);
        const unsigned W4 = w0 | ((w0 + 1) << 8) | ((w0 + 2) << 16) | ((w0 + 3) << 24);
        const int sum = (int)__builtin_amdgcn_udot4(I4, 0x01010101u, 0u, false);
        m10 += (int)__builtin_amdgcn_udot4(I4, W4, 0u, false) - 16 * sum;
        m01 += (r - HALF_PATCH) * sum;
    }
    m10 = wave_sum(m10); m01 = wave_sum(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    // horizontal blur pass: all 43 rows, columns x-18..x+21
    const unsigned T0 = P.blurPack0, T1 = P.blurPack1;
    unsigned kBlurTaps[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) kBlurTaps[q] = ((q < 4 ? T0 : T1) >> (8 * (q & 3))) & 255u;
#if SSLAM_DESCRIBE_MFMA
    // ... as a product with the banded matrix of the taps on the matrix cores (v_mfma_i32_32x32x32_i8; the vector unit is what this step is short
    // of, docs/history/DESIGN_rounds_1-4.md 5f): hb[r][c] = sum_k patch[r][k] * Toep[k][c], Toep[k][c] = tap[k - c - 2].  A = 32 patch rows, 16 bytes per lane straight
    // from LDS (as signed bytes p - 128: the accumulator starts at 128 * sum(taps), every column of the band sums to that); B = the band, three
    // 32 x 32 blocks precomputed on the host in operand order (the block k < 32, c >= 32 is zero); the k a byte stands for is the same in both
    // operands by construction.  D: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    {
        typedef int v4i __attribute__((ext_vector_type(4)));
        typedef int v16i __attribute__((ext_vector_type(16)));
        const int j = lane & 31, h = lane >> 5;
        const uint4 b0 = P.blurToep[lane], b1 = P.blurToep[64 + lane], b2 = P.blurToep[128 + lane];
        const v4i B0 = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w}, B1 = {(int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w}, B2 = {(int)b2.x, (int)b2.y, (int)b2.z, (int)b2.w};
        // the accumulators are sums of (p - 128) * tap: + 128 * sum(taps) per result.  Two results are packed per dword (v_perm of the low
        // halves), then the bias goes on with a packed 16-bit add -- no carry between the halves, and every true sum fits 16 bits
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        const unsigned biasU = (unsigned)(128 * P.blurTapSum) * 0x00010001u;
        const u16x2 bias2 = __builtin_bit_cast(u16x2, biasU);
        auto pack2 = [&](int a0, int a1) -> unsigned { return __builtin_bit_cast(unsigned, __builtin_bit_cast(u16x2, __builtin_amdgcn_perm((unsigned)a1, (unsigned)a0, 0x05040100u)) + bias2); };
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int rowA = min(32 * mt + j, PW - 1);
            const uint4 a0 = *(const uint4*)(patch + rowA * PP + 16 * h), a1 = *(const uint4*)(patch + rowA * PP + 32 + 16 * h);
            const v4i A0 = {(int)(a0.x ^ 0x80808080u), (int)(a0.y ^ 0x80808080u), (int)(a0.z ^ 0x80808080u), (int)(a0.w ^ 0x80808080u)};
            const v4i A1 = {(int)(a1.x ^ 0x80808080u), (int)(a1.y ^ 0x80808080u), (int)(a1.z ^ 0x80808080u), (int)(a1.w ^ 0x80808080u)};
            // transposed product (band^T x patch^T): the lane's 16 results of a block are output columns (r & 3) + 8 (r >> 2) + 4 h of ITS patch
            // row -- four runs of four adjacent columns, one 8-byte LDS store each
            const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // (an inline constant as the C operand: no registers)
            const int row = 32 * mt + j;
            unsigned short* hr = hb + min(row, PW - 1) * 40 + 4 * h;
            v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(B0, A0, zero, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(B1, A1, acc, 0, 0, 0);
            if (row < PW) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 w2; w2.x = pack2(acc[4 * g], acc[4 * g + 1]); w2.y = pack2(acc[4 * g + 2], acc[4 * g + 3]);
                    *(uint2*)(hr + 8 * g) = w2;
                }
            }
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(B2, A1, zero, 0, 0, 0);
            if (row < PW) {
                uint2 w2; w2.x = pack2(acc[0], acc[1]); w2.y = pack2(acc[2], acc[3]);
                *(uint2*)(hr + 32) = w2;          // columns 32 + 4 h .. + 3 (the rest of the second block is past column 39)
            }
        }
    }
#else
    for (int i = lane; i < PW * 10; i += 64) {
        const int r = i / 10, g4 = i - r * 10;
        const unsigned* p = (const unsigned*)(patch + r * PP) + g4;
        const unsigned d0 = p[0], d1 = p[1], d2 = p[2];
        // output j uses LDS bytes 4*g4 + 2 + j .. + 8 + j
        const unsigned lo0 = __builtin_amdgcn_alignbyte(d1, d0, 2), hi0 = __builtin_amdgcn_alignbyte(d2, d1, 2);
        const unsigned lo1 = __builtin_amdgcn_alignbyte(d1, d0, 3), hi1 = __builtin_amdgcn_alignbyte(d2, d1, 3);
        const unsigned lo2 = d1, hi2 = d2;
        const unsigned lo3 = __builtin_amdgcn_alignbyte(d2, d1, 1), hi3 = d2 >> 8;     // the fourth byte of hi has tap weight 0
        const unsigned o0 = __builtin_amdgcn_udot4(lo0, T0, __builtin_amdgcn_udot4(hi0, T1, 0u, false), false);
        const unsigned o1 = __builtin_amdgcn_udot4(lo1, T0, __builtin_amdgcn_udot4(hi1, T1, 0u, false), false);
        const unsigned o2 = __builtin_amdgcn_udot4(lo2, T0, __builtin_amdgcn_udot4(hi2, T1, 0u, false), false);
        const unsigned o3 = __builtin_amdgcn_udot4(lo3, T0, __builtin_amdgcn_udot4(hi3, T1, 0u, false), false);
        uint2 w2; w2.x = o0 | (o1 << 16); w2.y = o2 | (o3 << 16);
        *(uint2*)(hb + r * 40 + g4 * 4) = w2;
    }
#endif
    __syncthreads();
    if (TAP) {
        for (int i = lane; i < TAPW * TAPW; i += 64) {
            const int yy = i / TAPW - 18, xx = i - (i / TAPW) * TAPW - 18;
            const unsigned short* h = hb + __mul24(PR + yy - 3, 40) + (18 + xx);
            unsigned acc = 0;
#pragma unroll
            for (int q = 0; q < 7; ++q) acc = __umul24((unsigned)h[q * 40], kBlurTaps[q]) + acc;
            descOut[((size_t)b * cap + outIdx) * (TAPW * TAPW) + i] = (uint8_t)min((acc + 32768u) >> 16, 255u);      // (saturation only bites with taps that sum to 257: blur variant 1)
        }
    }
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float ang = __fmul_rn(angle, factorPI);
    double snD, csD;
    sincos_0_2pi((double)ang, snD, csD);                                       // D5 (common.h: identical to the library's after the rounding to float, for every float argument)
    const float a = (float)csD, bsn = (float)snD;
    unsigned nib = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int t[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const signed char* pp = kPat + ((lane * 4 + k) * 2 + e) * 2;
            float px = (float)pp[0], py = (float)pp[1];
            int yy = cv_roundf(__fadd_rn(__fmul_rn(px, bsn), __fmul_rn(py, a)));
            int xx = cv_roundf(__fsub_rn(__fmul_rn(px, a), __fmul_rn(py, bsn)));
            const unsigned short* h = hb + __mul24(PR + yy - 3, 40) + (18 + xx);
            unsigned acc = 0;
#pragma unroll
            for (int q = 0; q < 7; ++q) acc = __umul24((unsigned)h[q * 40], kBlurTaps[q]) + acc;
            t[e] = (int)min((acc + 32768u) >> 16, 255u);
        }
        nib |= (unsigned)(t[0] < t[1]) << k;
    }
    unsigned word = nib << (4 * (lane & 7));
    word |= __shfl_xor((int)word, 1, 64);
    word |= __shfl_xor((int)word, 2, 64);
    word |= __shfl_xor((int)word, 4, 64);
    if (!TAP && (lane & 7) == 0) ((unsigned*)descOut)[((size_t)b * cap + outIdx) * 8 + (lane >> 3)] = word;
    if (lane < 7) {
        float fx = (float)kx, fy = (float)ky;
        if (level != 0) { fx = __fmul_rn(fx, L.scale); fy = __fmul_rn(fy, L.scale); }
        unsigned w;
        switch (lane) {
            case 0: w = __float_as_uint(fx); break;
            case 1: w = __float_as_uint(fy); break;
            case 2: w = __float_as_uint(L.size); break;
            case 3: w = __float_as_uint(angle); break;
            case 4: w = __float_as_uint((float)score); break;
            case 5: w = (unsigned)level; break;
            default: w = 0xFFFFFFFFu; break;
        }
        ((unsigned*)kpOut)[((size_t)b * cap + outIdx) * 7 + lane] = w;
    }
}

// batch status: frames whose keypoint total exceeds the caller's capacity (k_describe drops the rows past it).  One workgroup.
// status[0] = number of such frames, status[1] = the first one (INT_MAX if none), status[2..3] = 0 (the line extractor's words)
__global__ __launch_bounds__(256) void k_orb_status(const int* __restrict__ selCount, int nlevels, int nframes, int cap, int* __restrict__ status) {
    __shared__ int cnt, first;
    if (threadIdx.x == 0) { cnt = 0; first = 0x7FFFFFFF; }
    __syncthreads();
    for (int b = threadIdx.x; b < nframes; b += 256) {
        int t = 0;
        for (int l = 0; l < nlevels; ++l) t += selCount[(size_t)b * nlevels + l];
        if (t > cap) { atomicAdd(&cnt, 1); atomicMin(&first, b); }
    }
    __syncthreads();
    if (threadIdx.x == 0) { status[0] = cnt; status[1] = first; status[2] = 0; status[3] = 0x7FFFFFFF; }
}

}  // namespace

// =============================================================== host side
struct sslam_orb {
    sslam_ctx* ctx;
    int nfeatures, nlevels, iniTh, minTh;
    float scaleFactorF;
    double scaleFactor;
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> perLevel;
    int umax[16];
    int planW = 0, planH = 0;
    Plan plan;
    const uint8_t* lastImg0 = nullptr; size_t lastImg0Pitch = 0, lastImg0Stride = 0; bool lastInPlace = false;      // the image of the last call (debug taps of level 0)
    std::vector<CellInfo> cells;
    std::vector<short> tabs;
    DevBuf dCells, dTabs, dPyr, dCellCount, dCand, dBufA, dBufB, dSel, dSelCount;
    DevBuf dImg, dKp, dDesc, dCounts;       // single-frame host path staging
    HostPinned hImg, hOut;
    int wsFrames = 0;
    int lastFrames = 0;
    int lastN = -1;                 // keypoints of the last sslam_orb_extract (still resident in dKp/dDesc)
    bool constsUploaded = false;
    int blurVariant = 0;            // sslam_orb_set_blur_variant
    hipEvent_t gateEvent = nullptr;         // sslam_orb_set_gate_event
    DevBuf dToep; int toepVariant = -1, toepSum = 0;      // the blur's band matrix as matrix-core operands, for the taps of blurVariant
};

static inline int cvRoundF(float v) { return (int)lrintf(v); }

// OpenCV 3.4.0's 8-bit Gaussian taps (decision D6's alternative, oracle/cvleaf.h gauss_taps_340): the float kernel times 256, every tap rounded
static std::vector<int> blur_taps_340(int n, double sigma) {
    std::vector<float> k(n); double sum = 0; const double s2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; k[i] = (float)std::exp(s2 * x * x); sum += k[i]; }
    sum = 1. / sum;
    std::vector<int> t(n);
    for (int i = 0; i < n; ++i) { k[i] = (float)(k[i] * sum); t[i] = (int)lrint((double)k[i] * 256.0); }
    return t;
}
static std::vector<int> blur_taps_q8(int n, double sigma) {
    // bit-exact 8.8 Gaussian taps with error diffusion, sum == 256 (decision D6)
    std::vector<double> k(n);
    double sum = 0, s2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; ++i) { double x = i - (n - 1) * 0.5; k[i] = std::exp(s2 * x * x); sum += k[i]; }
    std::vector<int> t(n);
    double err = 0; long isum = 0;
    for (int i = 0; i < n / 2; ++i) {
        double adj = k[i] / sum * 256.0 + err;
        int v = (int)lrint(adj);
        err = adj - v; t[i] = t[n - 1 - i] = v; isum += v;
    }
    t[n / 2] = (int)(256 - 2 * isum);
    return t;
}

static int build_plan(sslam_orb* o, int w, int h) {
    Plan& P = o->plan;
    memset(&P, 0, sizeof(P));
    P.nlevels = o->nlevels;
    o->cells.clear(); o->tabs.clear();
    size_t off = 0;
    int candOff = 0, selOff = 0;
    for (int l = 0; l < o->nlevels; ++l) {
        LevelInfo& L = P.L[l];
        float s = o->invScale[l];
        L.w = cvRoundF((float)w * s); L.h = cvRoundF((float)h * s);     // ComputePyramid :1112
        L.rcpGroups = 1.0f / (float)((L.w + 3) >> 2);
        if (L.w < 1 || L.h < 1 || L.w > 4095 + MINB || L.h > 4095 + MINB) { set_error("image size %dx%d unsupported at level %d", w, h, l); return SSLAM_ERR_UNSUPPORTED; }
        L.pitch = (L.w + 63) & ~63;
        L.off = (unsigned)off;
        off += (size_t)L.pitch * L.h;
        off = (off + 255) & ~(size_t)255;
        L.scale = o->scale[l];
        L.size = (float)(int)(31 * o->scale[l]);                          // :835
        L.nfeat = o->perLevel[l];
        // FAST cell grid, :771-829 / SURVEY D.1
        const int maxBX = L.w - EDGE + 3, maxBY = L.h - EDGE + 3;
        const float width = (float)(maxBX - MINB), height = (float)(maxBY - MINB);
        L.W = maxBX - MINB; L.H = maxBY - MINB;
        L.cellBeg = (int)o->cells.size();
        L.candOff = candOff;
        const int nCols = (int)(width / 30.f), nRows = (int)(height / 30.f);
        if (nCols > 0 && nRows > 0 && L.W > 0 && L.H > 0) {
            const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
            for (int i = 0; i < nRows; ++i) {
                const float iniY = (float)(MINB + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBY - 3) continue;
                if (maxY > maxBY) maxY = (float)maxBY;
                for (int j = 0; j < nCols; ++j) {
                    const float iniX = (float)(MINB + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBX - 6) continue;
                    if (maxX > maxBX) maxX = (float)maxBX;
                    CellInfo c;
                    c.level = (short)l; c.pad = 0;
                    c.x0 = (short)((int)iniX + 3); c.x1 = (short)((int)maxX - 3);
                    c.y0 = (short)((int)iniY + 3); c.y1 = (short)((int)maxY - 3);
                    if (c.x1 <= c.x0 || c.y1 <= c.y0) continue;       // view narrower than 7: FAST finds nothing
                    c.candOff = candOff;
                    int cw = c.x1 - c.x0, chh = c.y1 - c.y0;
                    candOff += ((cw + 1) / 2) * ((chh + 1) / 2);      // strict 3x3 NMS keeps <= 1 per 2x2
                    P.maxCellW = std::max(P.maxCellW, cw); P.maxCellH = std::max(P.maxCellH, chh);
                    {   // as k_fast_cells lays the tile out: tile column of pixel 0 = off + 3, off = (x0 - 3) & 3
                        const int T0 = ((c.x0 - 3) & 3) + 3, gmin = T0 >> 2, ng = ((T0 + cw - 1) >> 2) - gmin + 1;
                        c.magic = (1u << 19) / (unsigned)cw + 1; c.gmagic = (1u << 19) / (unsigned)ng + 1;
                    }
                    o->cells.push_back(c);
                }
            }
        }
        L.nCells = (int)o->cells.size() - L.cellBeg;
        L.candCap = candOff - L.candOff;
        P.maxCellsLevel = std::max(P.maxCellsLevel, L.nCells);
        // quadtree roots, :543-545
        if (L.W > 0 && L.H > 0) {
            L.nIni = (int)std::round((float)L.W / (float)L.H);
            if (L.nCells == 0) L.nIni = 0;            // no FAST cell fits: the level yields no keypoints whatever its aspect ratio
            if (L.nIni > MAX_ROOTS) { set_error("image %dx%d: level %d would start the quadtree with %d root nodes (limit %d)", w, h, l, L.nIni, MAX_ROOTS); return SSLAM_ERR_UNSUPPORTED; }
            L.hX = L.nIni > 0 ? (float)L.W / (float)L.nIni : 0.f;
        } else { L.nIni = 0; L.hX = 0; }
        L.selOff = selOff;
        L.selCap = std::max(L.nfeat + 3, 4 * std::max(L.nIni, 1) + 1);
        selOff += L.selCap;
        P.maxNodeCap = std::max(P.maxNodeCap, L.selCap + 2);
        // resize tables (level l from level l-1), cv::resize INTER_LINEAR 8u, A.5
        if (l > 0) {
            const LevelInfo& S = P.L[l - 1];
            const double inv_sx = (double)L.w / S.w, inv_sy = (double)L.h / S.h;
            const double sx_ = 1. / inv_sx, sy_ = 1. / inv_sy;
            L.tabX = (int)o->tabs.size() / 4;
            for (int dx = 0; dx < ((L.w + 3) & ~3); ++dx) {      // padded entries repeat the last column (never stored past pitch)
                const int dxe = std::min(dx, L.w - 1);
                float fx = (float)((dxe + 0.5) * sx_ - 0.5);
                int sx = (int)std::floor(fx);
                fx -= sx;
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx >= S.w - 1) { fx = 0; sx = S.w - 1; }
                o->tabs.push_back((short)sx);
                o->tabs.push_back((short)cvRoundF((1.f - fx) * 2048));
                o->tabs.push_back((short)cvRoundF(fx * 2048));
                o->tabs.push_back(0);
            }
            L.tabY = (int)o->tabs.size() / 4;
            for (int dy = 0; dy < L.h; ++dy) {
                float fy = (float)((dy + 0.5) * sy_ - 0.5);
                int sy = (int)std::floor(fy);
                fy -= sy;
                o->tabs.push_back((short)sy);
                o->tabs.push_back((short)cvRoundF((1.f - fy) * 2048));
                o->tabs.push_back((short)cvRoundF(fy * 2048));
                o->tabs.push_back(0);
            }
            while ((o->tabs.size() / 4) % 4) o->tabs.insert(o->tabs.end(), 4, (short)0);      // keep every tabX 32-byte aligned
        }
    }
    P.pyrFrame = (off + 16 + 255) & ~(size_t)255;      // 16 spare bytes: k_resize reads whole dwords past a row's last pixel
    P.nCellsFrame = (int)o->cells.size();
    P.candFrame = std::max(candOff, 1);
    P.selFrame = selOff;
    if (o->dCells.ensure(std::max<size_t>(o->cells.size(), 1) * sizeof(CellInfo)) != SSLAM_OK) return SSLAM_ERR_HIP;
    if (o->dTabs.ensure(std::max<size_t>(o->tabs.size(), 1) * sizeof(short)) != SSLAM_OK) return SSLAM_ERR_HIP;
    if (!o->cells.empty()) SSLAM_HIP(hipMemcpy(o->dCells.p, o->cells.data(), o->cells.size() * sizeof(CellInfo), hipMemcpyHostToDevice));
    if (!o->tabs.empty()) SSLAM_HIP(hipMemcpy(o->dTabs.p, o->tabs.data(), o->tabs.size() * sizeof(short), hipMemcpyHostToDevice));
    o->planW = w; o->planH = h;
    o->wsFrames = 0;
    return SSLAM_OK;
}

static int ensure_workspace(sslam_orb* o, int nframes) {
    if (nframes <= o->wsFrames) return SSLAM_OK;
    const Plan& P = o->plan;
    int rc;
    if ((rc = o->dPyr.ensure(P.pyrFrame * nframes))) return rc;
    if ((rc = o->dCellCount.ensure(sizeof(int) * (size_t)std::max(P.nCellsFrame, 1) * nframes))) return rc;
    if ((rc = o->dCand.ensure(sizeof(unsigned) * (size_t)P.candFrame * nframes))) return rc;
    if ((rc = o->dBufA.ensure(sizeof(unsigned) * (size_t)P.candFrame * nframes))) return rc;
    if ((rc = o->dBufB.ensure(sizeof(unsigned) * (size_t)P.candFrame * nframes))) return rc;
    if ((rc = o->dSel.ensure(sizeof(unsigned) * (size_t)P.selFrame * nframes))) return rc;
    if ((rc = o->dSelCount.ensure(sizeof(int) * (size_t)P.nlevels * nframes))) return rc;
    o->wsFrames = nframes;
    return SSLAM_OK;
}

extern "C" int sslam_orb_create(sslam_ctx* ctx, int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh, sslam_orb** out) {
    if (!ctx || !out || nlevels < 1 || nlevels > MAX_LEVELS || nfeatures < 0 || !(scaleFactor > 1.0f) || minTh < 1 || iniTh < minTh || iniTh > 255) {
        set_error("sslam_orb_create: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    sslam_orb* o = new sslam_orb();
    o->ctx = ctx; o->nfeatures = nfeatures; o->nlevels = nlevels; o->iniTh = iniTh; o->minTh = minTh;
    o->scaleFactorF = scaleFactor; o->scaleFactor = (double)scaleFactor;     // member is a double holding the float, include/ORBextractor.h:97
    // src/ORBextractor.cc:415-446
    o->scale.resize(nlevels); o->sigma2.resize(nlevels); o->invScale.resize(nlevels); o->invSigma2.resize(nlevels);
    o->scale[0] = 1.0f; o->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; ++i) { o->scale[i] = (float)(o->scale[i - 1] * o->scaleFactor); o->sigma2[i] = o->scale[i] * o->scale[i]; }
    for (int i = 0; i < nlevels; ++i) { o->invScale[i] = 1.0f / o->scale[i]; o->invSigma2[i] = 1.0f / o->sigma2[i]; }
    o->perLevel.resize(nlevels);
    float factor = (float)(1.0f / o->scaleFactor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) { o->perLevel[l] = cvRoundF(nDesired); sum += o->perLevel[l]; nDesired *= factor; }
    o->perLevel[nlevels - 1] = std::max(nfeatures - sum, 0);
    // umax, :454-469
    {
        int v, v0, vmax = (int)std::floor(HALF_PATCH * std::sqrt(2.f) / 2 + 1);
        int vmin = (int)std::ceil(HALF_PATCH * std::sqrt(2.f) / 2);
        const double hp2 = HALF_PATCH * HALF_PATCH;
        for (v = 0; v <= vmax; ++v) o->umax[v] = (int)lrint(std::sqrt(hp2 - v * v));
        for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) { while (o->umax[v0] == o->umax[v0 + 1]) ++v0; o->umax[v] = v0; ++v0; }
    }
    *out = o;
    return SSLAM_OK;
}

extern "C" int sslam_orb_set_blur_variant(sslam_orb* o, int variant) {
    if (!o || (variant != 0 && variant != 1)) { set_error("sslam_orb_set_blur_variant: invalid arguments"); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::recursive_mutex> lk(o->ctx->mu);
    o->blurVariant = variant;
    return SSLAM_OK;
}

extern "C" int sslam_orb_destroy(sslam_orb* o) {
    if (!o) return SSLAM_OK;
    (void)hipSetDevice(o->ctx->device);
    (void)hipStreamSynchronize(o->ctx->stream);
    DevBuf* bufs[] = {&o->dCells, &o->dTabs, &o->dPyr, &o->dCellCount, &o->dCand, &o->dBufA, &o->dBufB, &o->dSel, &o->dSelCount, &o->dImg, &o->dKp, &o->dDesc, &o->dCounts, &o->dToep};
    for (DevBuf* b : bufs) b->release();
    o->hImg.release(); o->hOut.release();
    delete o;
    return SSLAM_OK;
}

extern "C" int sslam_orb_get_scales(const sslam_orb* o, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int32_t* per_level) {
    if (!o) return SSLAM_ERR_INVALID;
    for (int i = 0; i < o->nlevels; ++i) {
        if (scale) scale[i] = o->scale[i];
        if (inv_scale) inv_scale[i] = o->invScale[i];
        if (sigma2) sigma2[i] = o->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = o->invSigma2[i];
        if (per_level) per_level[i] = o->perLevel[i];
    }
    return SSLAM_OK;
}

extern "C" int sslam_orb_set_gate_event(sslam_orb* o, void* hip_event) {
    if (!o) { set_error("sslam_orb_set_gate_event: null handle"); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::recursive_mutex> lk(o->ctx->mu);
    o->gateEvent = (hipEvent_t)hip_event;
    return SSLAM_OK;
}

extern "C" int sslam_orb_max_keypoints(const sslam_orb* o) {
    if (!o) return SSLAM_ERR_INVALID;
    int n = 0;
    for (int l = 0; l < o->nlevels; ++l) n += std::max(o->perLevel[l] + 3, 4 * 8 + 1);
    return n;
}

extern "C" int sslam_orb_extract_batch_dev(sslam_orb* o, const uint8_t* d_images, int w, int h, size_t pitch, size_t image_stride,
                                           int nframes, sslam_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts, int cap, void* stream_) {
    if (!o || !d_images || !d_kp || !d_desc || !d_counts || w <= 0 || h <= 0 || nframes <= 0 || cap <= 0 || pitch < (size_t)w) {
        set_error("sslam_orb_extract_batch_dev: invalid arguments"); return SSLAM_ERR_INVALID;
    }
    std::lock_guard<std::recursive_mutex> lk(o->ctx->mu);      // plan, workspace and profile records are shared state
    SSLAM_HIP(hipSetDevice(o->ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : o->ctx->stream;
    int rc;
    if (w != o->planW || h != o->planH) {
        SSLAM_HIP(hipStreamSynchronize(st));
        if ((rc = build_plan(o, w, h))) return rc;
    }
    {   // the blur taps of this extractor (decision D6, or its OpenCV-3.4.0 alternative: sslam_orb_set_blur_variant)
        const std::vector<int> taps = o->blurVariant == 1 ? blur_taps_340(7, 2.0) : blur_taps_q8(7, 2.0);
        for (int t : taps) if (t < 0 || t > 255) { set_error("blur taps do not fit a byte"); return SSLAM_ERR_UNSUPPORTED; }
        o->plan.blurPack0 = (unsigned)taps[0] | ((unsigned)taps[1] << 8) | ((unsigned)taps[2] << 16) | ((unsigned)taps[3] << 24);
        o->plan.blurPack1 = (unsigned)taps[4] | ((unsigned)taps[5] << 8) | ((unsigned)taps[6] << 16);
        if (o->toepVariant != o->blurVariant) {
            // k_describe's horizontal pass as int8 matrix-core operands: blocks (k < 32, c < 32), (k >= 32, c < 32), (k >= 32, c >= 32) of
            // Toep[k][c] = tap[k - c - 2] (k = LDS byte of the patch row, 48 of them; c = output column, 40 of them); lane l holds bytes
            // e = 0..15 for k = 32 ks + 16 (l >> 5) + e, c = 32 nt + (l & 31)
            std::vector<signed char> tb(3 * 64 * 16, 0);
            const int blk[3][2] = {{0, 0}, {0, 1}, {1, 1}};      // (nt, ks)
            int sum = 0;
            for (int t : taps) { if (t > 127) { set_error("blur taps do not fit a signed byte"); return SSLAM_ERR_UNSUPPORTED; } sum += t; }
            for (int f = 0; f < 3; ++f) for (int l = 0; l < 64; ++l) for (int e = 0; e < 16; ++e) {
                const int k = 32 * blk[f][1] + 16 * (l >> 5) + e, c = 32 * blk[f][0] + (l & 31), q = k - c - 2;
                if (q >= 0 && q <= 6 && c < 40 && k < 48) tb[((size_t)f * 64 + l) * 16 + e] = (signed char)taps[q];
            }
            if ((rc = o->dToep.ensure(tb.size()))) return rc;
            SSLAM_HIP(hipStreamSynchronize(st));
            SSLAM_HIP(hipMemcpy(o->dToep.p, tb.data(), tb.size(), hipMemcpyHostToDevice));
            o->toepVariant = o->blurVariant; o->toepSum = sum;
        }
        o->plan.blurToep = o->dToep.as<uint4>(); o->plan.blurTapSum = o->toepSum;
    }
    if (!o->constsUploaded) {
        SSLAM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(kUmax), o->umax, sizeof(int) * 16));
        unsigned mask[31 * 8];
        for (int r = 0; r < 31; ++r)
            for (int m = 0; m < 8; ++m) {
                unsigned w = 0;
                for (int k = 0; k < 4; ++k) { const int u = 4 * m + k - 15; if (std::abs(u) <= o->umax[std::abs(r - 15)]) w |= 0xFFu << (8 * k); }
                mask[r * 8 + m] = w;
            }
        SSLAM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(kDiscMask), mask, sizeof(mask)));
        o->constsUploaded = true;
    }
    if (nframes > o->wsFrames) { SSLAM_HIP(hipStreamSynchronize(st)); }
    if ((rc = ensure_workspace(o, nframes))) return rc;
    Plan P = o->plan;
    uint8_t* pyr = o->dPyr.as<uint8_t>();
    // level 0 in place when the caller's layout allows aligned dword loads (see Plan::img0); SSLAM_ORB_COPY_LEVEL0=1 forces the copy (A/B knob)
    // The aligned dword reads of the kernels may touch the padding bytes [w, pitch) of a row: with padded rows the frames must therefore be whole pitch x h blocks
    // (image_stride >= pitch * h; include/sslam_frontend.h states it, and that the LAST frame's buffer must hold pitch * h bytes) -- a layout that does not say so is copied.
    const bool inPlace = ((uintptr_t)d_images & 3) == 0 && (pitch & 3) == 0 && (image_stride & 3) == 0 && pitch <= 0x7FFFFFFF &&
                         (pitch == (size_t)w || image_stride >= pitch * (size_t)h) && !getenv("SSLAM_ORB_COPY_LEVEL0");
    P.img0 = inPlace ? d_images : nullptr; P.img0Stride = image_stride; P.img0Pitch = (int)pitch;
    o->lastImg0 = d_images; o->lastImg0Pitch = pitch; o->lastImg0Stride = image_stride; o->lastInPlace = inPlace;
    if (!inPlace) {
        dim3 blk(64), grd((w + 64 * 16 - 1) / (64 * 16), h, nframes);
        { sslam::ProfScope _ps(o->ctx, "k_copy_level0", st); hipLaunchKernelGGL(k_copy_level0, grd, blk, 0, st, d_images, pitch, image_stride, pyr, P.pyrFrame, w, h, P.L[0].pitch); }
    }
    for (int l = 1; l < P.nlevels; ++l) {
        const unsigned gx = (unsigned)((((P.L[l].w + 3) / 4) * P.L[l].h + 255) / 256);
        // Workgroups WALK THE FRAMES (round 6): everything a thread derives from its four output columns and its row -- three table entries, the source window, the byte
        // selectors and coefficient pairs -- is the same for every frame of the batch, so a grid of ~32 workgroups per compute unit whose workgroups step through the frames
        // computes it once: 9.0 -> 6.0 ms for the seven launches alone (10.2 -> 6.0 in the harness of call V), the two-stream step 158.3 -> 153.7 ms.
        // SSLAM_RESIZE_GRID_WGS=n: about n workgroups per launch (0: one per frame and tile, the form of rounds 1-5).
        int wantWgs = 32 * o->ctx->num_cus;
        if (const char* e = getenv("SSLAM_RESIZE_GRID_WGS")) wantWgs = atoi(e);
        const int gy = wantWgs > 0 ? std::max(1, std::min(nframes, wantWgs / (int)std::max(gx, 1u))) : nframes;
        dim3 blk(256), grd(gx, gy);
        const bool fromImage = l == 1 && inPlace;
        { sslam::ProfScope _ps(o->ctx, "k_resize", st);
          hipLaunchKernelGGL(k_resize, grd, blk, 0, st, fromImage ? d_images : pyr + P.L[l - 1].off, fromImage ? image_stride : P.pyrFrame, fromImage ? (int)pitch : P.L[l - 1].pitch, fromImage ? 1 : 0,
                             pyr, P.pyrFrame, P.L[l - 1], P.L[l], o->dTabs.as<short4>(), nframes); }
    }
    if (o->gateEvent) SSLAM_HIP(hipStreamWaitEvent(st, o->gateEvent, 0));      // sslam_orb_set_gate_event: the pyramid is built ahead, the rest waits (e.g. for the line branch's sequential core)
    if (P.nCellsFrame > 0) {
        int tileP = (P.maxCellW + 6 + 3 + 7) & ~3, scP = P.maxCellW + 2;
        size_t lds = (size_t)(P.maxCellH + 6) * tileP + (((size_t)(P.maxCellH + 2) * scP + 3) & ~(size_t)3) + 3 * (size_t)P.maxCellH * P.maxCellW + 32;
        dim3 grd(8 * ((P.nCellsFrame + 7) / 8), nframes);
        { sslam::ProfScope _ps(o->ctx, "k_fast_cells", st); hipLaunchKernelGGL(k_fast_cells, grd, dim3(64), lds, st, pyr, P.pyrFrame, P, o->dCells.as<CellInfo>(), o->dCand.as<unsigned>(),
                           o->dCellCount.as<int>(), o->iniTh, o->minTh, tileP, scP); }
    }
    {
        int NC = P.maxNodeCap + 2, NCp2 = 1;
        while (NCp2 < NC) NCp2 <<= 1;
        size_t lds = sizeof(unsigned long long) * 2 * NCp2 + sizeof(int) * (P.maxCellsLevel + 1) + sizeof(unsigned) * 5 * (size_t)NC;
        dim3 grd(P.nlevels, nframes);
        { sslam::ProfScope _ps(o->ctx, "k_octree", st); hipLaunchKernelGGL(k_octree, grd, dim3(64), lds, st, o->dCand.as<unsigned>(), o->dCellCount.as<int>(), o->dCells.as<CellInfo>(),
                           o->dBufA.as<unsigned>(), o->dBufB.as<unsigned>(), o->dSel.as<unsigned>(), o->dSelCount.as<int>(), P, NC, NCp2); }
    }
    {
        dim3 grd(P.selFrame, nframes);
        { sslam::ProfScope _ps(o->ctx, "k_describe", st); hipLaunchKernelGGL(k_describe<false>, grd, dim3(64), 0, st, pyr, P.pyrFrame, P, o->dSel.as<unsigned>(), o->dSelCount.as<int>(),
                           d_kp, d_desc, d_counts, cap); }
    }
    SSLAM_HIP(hipGetLastError());
    o->lastFrames = nframes;
    return SSLAM_OK;
}

extern "C" int sslam_orb_extract(sslam_orb* o, const uint8_t* gray, int w, int h, size_t stride,
                                 sslam_keypoint* kp_out, uint8_t* desc_out, int cap, int* n_out) {
    if (!o || !n_out) { set_error("sslam_orb_extract: null handle"); return SSLAM_ERR_INVALID; }
    if (w == 0 || h == 0 || !gray) { *n_out = 0; return SSLAM_OK; }      // empty image: return, outputs untouched (:1046-1047)
    if (w < 0 || h < 0 || stride < (size_t)w || !kp_out || !desc_out) { set_error("sslam_orb_extract: invalid arguments"); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::recursive_mutex> lk(o->ctx->mu);
    SSLAM_HIP(hipSetDevice(o->ctx->device));
    hipStream_t st = o->ctx->stream;
    const int icap = sslam_orb_max_keypoints(o);
    int rc;
    const size_t dpitch = ((size_t)w + 63) & ~(size_t)63;
    if ((rc = o->dImg.ensure(dpitch * h))) return rc;
    if ((rc = o->dKp.ensure(sizeof(sslam_keypoint) * (size_t)icap))) return rc;
    if ((rc = o->dDesc.ensure(32 * (size_t)icap))) return rc;
    if ((rc = o->dCounts.ensure(sizeof(int) * 4))) return rc;
    if ((rc = o->hOut.ensure((sizeof(sslam_keypoint) + 32) * (size_t)icap + 64))) return rc;
    SSLAM_HIP(hipMemcpy2DAsync(o->dImg.p, dpitch, gray, stride, w, h, hipMemcpyHostToDevice, st));
    if ((rc = sslam_orb_extract_batch_dev(o, o->dImg.as<uint8_t>(), w, h, dpitch, dpitch * h, 1, o->dKp.as<sslam_keypoint>(),
                                          o->dDesc.as<uint8_t>(), o->dCounts.as<int>(), icap, st))) return rc;
    uint8_t* hp = o->hOut.as<uint8_t>();
    SSLAM_HIP(hipMemcpyAsync(hp, o->dCounts.p, sizeof(int), hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(hp + 64, o->dKp.p, sizeof(sslam_keypoint) * (size_t)icap, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipMemcpyAsync(hp + 64 + sizeof(sslam_keypoint) * (size_t)icap, o->dDesc.p, 32 * (size_t)icap, hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    int n = *(int*)hp;
    *n_out = n;
    o->lastN = n;
    if (n > cap) { set_error("sslam_orb_extract: %d keypoints exceed caller capacity %d", n, cap); return SSLAM_ERR_CAPACITY; }
    memcpy(kp_out, hp + 64, sizeof(sslam_keypoint) * (size_t)n);
    memcpy(desc_out, hp + 64 + sizeof(sslam_keypoint) * (size_t)icap, 32 * (size_t)n);
    return SSLAM_OK;
}

extern "C" int sslam_orb_batch_status_dev(sslam_orb* o, int cap, int32_t* d_status4, void* stream_) {
    if (!o || !d_status4 || cap <= 0 || o->lastFrames <= 0) { set_error("sslam_orb_batch_status_dev: invalid arguments (or no batch yet)"); return SSLAM_ERR_INVALID; }
    SSLAM_HIP(hipSetDevice(o->ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : o->ctx->stream;
    hipLaunchKernelGGL(k_orb_status, dim3(1), dim3(256), 0, st, o->dSelCount.as<int>(), o->plan.nlevels, o->lastFrames, cap, d_status4);
    SSLAM_HIP(hipGetLastError());
    return SSLAM_OK;
}

extern "C" int sslam_orb_batch_status(sslam_orb* o, int cap, void* stream_, int* truncated_frames_out, int* first_frame_out) {
    if (!o) { set_error("sslam_orb_batch_status: null handle"); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::recursive_mutex> lk(o->ctx->mu);
    int rc;
    if ((rc = o->dCounts.ensure(sizeof(int) * 4))) return rc;
    hipStream_t st = stream_ ? (hipStream_t)stream_ : o->ctx->stream;
    if ((rc = sslam_orb_batch_status_dev(o, cap, o->dCounts.as<int32_t>(), st))) return rc;
    int h[4] = {0, 0, 0, 0};
    SSLAM_HIP(hipMemcpyAsync(h, o->dCounts.p, sizeof(h), hipMemcpyDeviceToHost, st));
    SSLAM_HIP(hipStreamSynchronize(st));
    if (truncated_frames_out) *truncated_frames_out = h[0];
    if (first_frame_out) *first_frame_out = h[0] ? h[1] : -1;
    if (h[0]) { set_error("sslam_orb_batch_status: %d frame(s) hold more keypoints than the capacity %d (first: frame %d)", h[0], cap, h[1]); return SSLAM_ERR_CAPACITY; }
    return SSLAM_OK;
}

extern "C" int sslam_orb_debug_level(sslam_orb* o, int frame, int level, uint8_t* out, int* w, int* h) {
    if (!o || frame < 0 || frame >= o->lastFrames || level < 0 || level >= o->nlevels) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(o->ctx->device));
    SSLAM_HIP(hipStreamSynchronize(o->ctx->stream));
    const LevelInfo& L = o->plan.L[level];
    if (w) *w = L.w;
    if (h) *h = L.h;
    // level 0 is the image of the last call itself (read in place when its layout allowed it: Plan::img0) -- the caller's buffer must still be alive
    if (out && level == 0 && o->lastInPlace) SSLAM_HIP(hipMemcpy2D(out, L.w, o->lastImg0 + (size_t)frame * o->lastImg0Stride, o->lastImg0Pitch, L.w, L.h, hipMemcpyDeviceToHost));
    else if (out) SSLAM_HIP(hipMemcpy2D(out, L.w, o->dPyr.as<uint8_t>() + (size_t)frame * o->plan.pyrFrame + L.off, L.pitch, L.w, L.h, hipMemcpyDeviceToHost));
    return SSLAM_OK;
}

extern "C" int sslam_orb_debug_candidates(sslam_orb* o, int frame, int level, int32_t* xys, int cap, int* n_out) {
    if (!o || frame < 0 || frame >= o->lastFrames || level < 0 || level >= o->nlevels || !n_out) return SSLAM_ERR_INVALID;
    SSLAM_HIP(hipSetDevice(o->ctx->device));
    SSLAM_HIP(hipStreamSynchronize(o->ctx->stream));
    const Plan& P = o->plan;
    const LevelInfo& L = P.L[level];
    std::vector<int> counts(std::max(L.nCells, 1));
    std::vector<unsigned> cand(std::max(L.candCap, 1));
    if (L.nCells) SSLAM_HIP(hipMemcpy(counts.data(), o->dCellCount.as<int>() + (size_t)frame * P.nCellsFrame + L.cellBeg, sizeof(int) * L.nCells, hipMemcpyDeviceToHost));
    if (L.candCap) SSLAM_HIP(hipMemcpy(cand.data(), o->dCand.as<unsigned>() + (size_t)frame * P.candFrame + L.candOff, sizeof(unsigned) * L.candCap, hipMemcpyDeviceToHost));
    int n = 0;
    for (int c = 0; c < L.nCells; ++c) {
        const CellInfo& ci = o->cells[L.cellBeg + c];
        for (int k = 0; k < counts[c]; ++k, ++n) {
            if (n < cap) {
                unsigned v = cand[ci.candOff - L.candOff + k];
                xys[n * 3] = v & 0xFFF; xys[n * 3 + 1] = (v >> 12) & 0xFFF; xys[n * 3 + 2] = v >> 24;
            }
        }
    }
    *n_out = n;
    return SSLAM_OK;
}

// Stage tap for a7 (GaussianBlur 7x7 sigma 2 on each level, src/ORBextractor.cc:1085-1086): the blurred image never exists in memory
// (the blur is evaluated inside k_describe where rBRIEF taps land), so the tap re-runs k_describe<true> over the selection of the last
// batch and returns, per keypoint in output order, the blurred 37x37 window around it plus its KeyPoint record.
extern "C" int sslam_orb_debug_blur_patches(sslam_orb* o, int frame, sslam_keypoint* kp_out, uint8_t* patches_out, int cap, int* n_out) {
    if (!o || frame < 0 || frame >= o->lastFrames || !kp_out || !patches_out || !n_out || cap <= 0) return SSLAM_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(o->ctx->mu);
    SSLAM_HIP(hipSetDevice(o->ctx->device));
    hipStream_t st = o->ctx->stream;
    SSLAM_HIP(hipStreamSynchronize(st));
    Plan P = o->plan;
    if (o->lastInPlace) { P.img0 = o->lastImg0 + (size_t)frame * o->lastImg0Stride; P.img0Stride = 0; P.img0Pitch = (int)o->lastImg0Pitch; }      // (the kernel below runs as frame 0)
    const int icap = sslam_orb_max_keypoints(o);
    DevBuf dk, dp, dc;
    int rc;
    if ((rc = dk.ensure(sizeof(sslam_keypoint) * (size_t)icap)) || (rc = dp.ensure((size_t)icap * TAPW * TAPW)) || (rc = dc.ensure(sizeof(int) * 4))) { dk.release(); dp.release(); dc.release(); return rc; }
    hipLaunchKernelGGL(k_describe<true>, dim3(P.selFrame, 1), dim3(64), 0, st, o->dPyr.as<uint8_t>() + (size_t)frame * P.pyrFrame, P.pyrFrame, P,
                       o->dSel.as<unsigned>() + (size_t)frame * P.selFrame, o->dSelCount.as<int>() + (size_t)frame * P.nlevels,
                       dk.as<sslam_keypoint>(), dp.as<uint8_t>(), dc.as<int>(), icap);
    int n = 0;
    hipError_t e = hipMemcpyAsync(&n, dc.p, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess && n <= cap) {
        e = hipMemcpy(kp_out, dk.p, sizeof(sslam_keypoint) * (size_t)n, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(patches_out, dp.p, (size_t)n * TAPW * TAPW, hipMemcpyDeviceToHost);
    }
    dk.release(); dp.release(); dc.release();
    if (e != hipSuccess) { set_error("sslam_orb_debug_blur_patches: %s", hipGetErrorString(e)); return SSLAM_ERR_HIP; }
    *n_out = n;
    if (n > cap) { set_error("sslam_orb_debug_blur_patches: %d keypoints exceed capacity %d", n, cap); return SSLAM_ERR_CAPACITY; }
    return SSLAM_OK;
}

// Keep the features of the last sslam_orb_extract call on the device as a frame handle (device-to-device snapshot of the
// keypoints + descriptors the extractor still holds): the Frame that ExtractORB just filled never has to be uploaded
// again for SearchByProjection / knn matching.
int sslam_frame_from_device(sslam_ctx* ctx, int kind, const void* d_feats, const uint8_t* d_desc, int n, const float bounds[4], sslam_frame** out);
extern "C" int sslam_frame_from_orb(sslam_orb* o, const float bounds[4], sslam_frame** out) {
    if (!o || !bounds || !out) { set_error("sslam_frame_from_orb: invalid arguments"); return SSLAM_ERR_INVALID; }
    if (o->lastN < 0) { set_error("sslam_frame_from_orb: no sslam_orb_extract call to snapshot"); return SSLAM_ERR_INVALID; }
    std::lock_guard<std::recursive_mutex> lk(o->ctx->mu);
    return sslam_frame_from_device(o->ctx, 0, o->dKp.p, o->dDesc.as<uint8_t>(), o->lastN, bounds, out);
}

extern "C" sslam_ctx* sslam_orb_context(sslam_orb* o) { return o ? o->ctx : nullptr; }
