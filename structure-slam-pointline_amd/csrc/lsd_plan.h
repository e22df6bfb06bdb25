// LSD / LBD: constants, per-frame workspace plan, rectangle record, angle helpers.
// Part of lines.hip (included there, inside its anonymous namespace: one translation unit, so device helpers are shared
// without relocatable device code).  Not a standalone header.
#pragma once

constexpr double kPI = 3.14159265358979323846;
constexpr double DEG2RAD = kPI / 180;
constexpr double M_3_2_PI_ = (3 * kPI) / 2, M_2PI_ = 2 * kPI;
constexpr float NOTDEF_F = -1024.0f;
// Per-pixel planes of the scaled image, all row-major (round 3: the 16-byte {angle, cos, sin, |g|^2} record and the separate angle map are gone):
//   T   4 B  level-line angle in degrees (fastAtan2: [0, 360), never -0) for DEFINED pixels, NOTDEF_F otherwise.  The SIGN BIT is the
//            `used` map of region growing: a pixel that belongs to a region holds -angle, so "unused and defined" is one integer
//            compare (bits >= 0), releasing a pixel clears the bit again, and the angle survives for the rectangle counter
//            (|T|; NOTDEF becomes 1024, above every angle).
//   Cs  8 B  {cosf, sinf} of the angle (decision D5); read with T by region growing for the accepted pixels' direction sums.
//            Written for DEFINED pixels only (round 4): an undefined pixel's entry is whatever the workspace held and is never used.
//   S   4 B  gx^2 + gy^2, defined pixels only: region2rect gathers its weights from it.
// plus, per SEGMENT (256 pixels of one row = one wave of k_lsd_grad), the list of its defined pixels in raster order, |g|^2 << 8 | column
// (k_lsd_hist turns it into bin << 8 | column): what the counting sort reads instead of a dense plane.  The lists share their memory with
// the core's overflow region lists (offReg): they are dead when the core starts.
// 16 bytes per pixel written by k_lsd_grad in round 3 (24 before), 4 B + ~24 B per defined pixel now.  (A tiled form of T / Cs -- 8 x 4 and 4 x 4 pixels per 128-byte line -- was built
// and measured first: L1 misses of the sequential core -12 %, its time unchanged, the rectangle counter +40 % for the address
// arithmetic; profiles/README.md, round 3.)
// (A form with Cs and S of a pixel in ONE 16-byte record was measured in round 4: k_lsd_grad 11.2 -> 13.0 ms, the core unchanged -- it is the bytes that count, not the bursts.)
constexpr int CS_SHIFT = 3, S_SHIFT = 2;      // log2 of the byte stride of the Cs / S entries
constexpr unsigned USED_BIT = 0x80000000u;
constexpr int N_BINS = 1024;
#ifndef SSLAM_TILE_PX
#define SSLAM_TILE_PX 8192
#endif
constexpr int TILE_PX = SSLAM_TILE_PX;          // raster tile of the counting sort (rounded down to whole rows: LsdPlan::tileRows)
constexpr int MAX_SEG = 8192;           // segments per frame (LSD output capacity)
constexpr int NUM_BANDS = 9, BAND_W = 7, LSP_H = 63;

struct LsdPlan {
    int w, h;                 // source image
    int sw, sh, spitch;       // scaled image (0.8x)
    int npx;                  // sw*sh
    int nTiles, tileRows;     // counting-sort tiles: tileRows whole rows each
    int nXB;                  // segments per row = ceil(sw / 256)
    int sMin;                 // smallest gx^2 + gy^2 of a defined pixel (k_grad_smin)
    size_t frameBytes;        // per-frame workspace
    int tW, cW;               // row pitch (elements) of the T and Cs planes (= sw)
    size_t offBlur, offT, offS, offCs, offCand, offFlag, offNfa, offOrder, offTileHist, offReg, offSeg, offMisc, offDxy, offKl, offSortIdx, offSegCnt, offComp, offSorted, offLbdDir;
    int blurTaps[7];          // sigma 0.75, 7 taps (q8)
    int blur5Taps[5];         // sigma 1, 5 taps (q8)
    int tabX, tabY;           // offsets into the resize table (int: ofs, c1)
    double rho, prec, p, logNT;
    int minRegSize;
    // stated decisions with a selectable alternative (sslam_lines_set_*)
    int nfaVariant;           // D11: first term of nfa()'s log1term -- 1 (default): (double(n) + 1); 0: log_gamma(n + 1)
    int lbdBitOrder;          // D12: LBD byte packing -- 1 (default): comparison i -> bit 7 - i (0x80 >> i); 0: comparison i -> bit i
    int lsdResize;            // D7: the 0.8x rescale -- 0: INTER_LINEAR_EXACT (q8 coefficients, one rounding); 1: INTER_LINEAR (11-bit coefficients, the two-stage 8u rounding)
};

struct Misc {                 // per-frame scalars
    int maxS;                 // max gx^2+gy^2 over defined pixels
    int nDefined;
    int nSeg;
    int nCand;                // rectangles handed from the sequential core to the NFA stage
    int nKl;
    int overflow;
    int claim;                // frame 0 only: frames handed out so far to the persistent workgroups of the sequential core (k_lsd_regions)
    int pad_;
    long long cyc[8];         // master-wave cycle breakdown (debug): grow, rect, refine, nfa count, nfa math, seed scan
};

// Streaming hand-over of candidate rectangles from the cluster form's main wave to the NFA stage (the default for calls of up to 64 frames; lsd_cluster.h writes, lsd_nfa.h's
// k_nfa_stream reads -- two translation units, hence here).  Lives in the zeroed head of a frame's cluster slot, in a cache line of its own.
struct NfaStreamCtl {
    int candReady;            // rectangles published so far (their records are complete in the slot's staging array)
    int candFinal;            // 0 while the main wave runs, then 1 + the frame's number of rectangles (everything is published)
    int claim;                // rectangles handed out to consumer waves so far (CAS; a claim takes 1 .. NFA_STREAM_BLOCK of them)
    int expired;              // consumer waves that stopped waiting (the launch behind the core takes what they left)
};
constexpr int NFA_STREAM_BLOCK = 8;        // rectangles per claim at most: one wave evaluates 8 x 5 candidates in one pass of its 64 lanes
constexpr int NFA_STREAM_CTL_OFF = 256;    // byte offset of NfaStreamCtl in the slot's zeroed head (512 bytes): a 128-byte line of its own -- the consumers poll it, and the line of ClCtl's cursors is where the helpers claim

// element index of pixel (x, y) in the T / Cs planes
__device__ __forceinline__ int tix(int x, int y, int tW) { return y * tW + x; }
__device__ __forceinline__ int cix(int x, int y, int cW) { return y * cW + x; }

// ------------------------------------------------------------------ the sequential core
struct RectD { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; };

#ifndef SSLAM_LSD_QCAP
#define SSLAM_LSD_QCAP 768
#endif
constexpr int QCAP = SSLAM_LSD_QCAP;  // region points kept in LDS; longer regions continue in global memory
constexpr int MAXC = 5;             // rectangle candidates evaluated per NFA job

__device__ __forceinline__ double angle_diff_signed(double a, double b) {
    double diff = a - b;
    while (diff <= -kPI) diff += M_2PI_;
    while (diff > kPI) diff -= M_2PI_;
    return diff;
}
__device__ __forceinline__ bool is_aligned_val(float aDeg, double theta, double prec) {
    // isAligned: |theta - a|, folded once around the circle (fabs == the reference's conditional negations; +-0 compare alike)
    double n_theta = fabs(theta - (double)aDeg * DEG2RAD);
    const double wrapped = fabs(n_theta - M_2PI_);
    n_theta = n_theta > M_3_2_PI_ ? wrapped : n_theta;
    return aDeg != NOTDEF_F && n_theta <= prec;
}
