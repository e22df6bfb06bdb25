// LSD / LBD: constants, per-frame workspace plan, rectangle record, angle helpers.
// Part of lines.hip (included there, inside its anonymous namespace: one translation unit, so device helpers are shared
// without relocatable device code).  Not a standalone header.
#pragma once

constexpr double kPI = 3.14159265358979323846;
constexpr double DEG2RAD = kPI / 180;
constexpr double M_3_2_PI_ = (3 * kPI) / 2, M_2PI_ = 2 * kPI;
constexpr float NOTDEF_F = -1024.0f;
constexpr float USED_F = -2048.0f;       // written over pix[].x while a pixel belongs to a region (the `used` map)
constexpr int N_BINS = 1024;
constexpr int TILE_PX = 8192;           // raster tile of the counting sort
constexpr int MAX_SEG = 8192;           // segments per frame (LSD output capacity)
constexpr int NUM_BANDS = 9, BAND_W = 7, LSP_H = 63;

struct LsdPlan {
    int w, h;                 // source image
    int sw, sh, spitch;       // scaled image (0.8x)
    int npx;                  // sw*sh
    int nTiles;
    size_t frameBytes;        // per-frame workspace
    size_t offBlur, offAng, offS, offPix, offCand, offFlag, offNfa, offOrder, offTileHist, offReg, offSeg, offMisc, offDxy, offKl, offSortIdx;
    int blurTaps[7];          // sigma 0.75, 7 taps (q8)
    int blur5Taps[5];         // sigma 1, 5 taps (q8)
    int tabX, tabY;           // offsets into the resize table (int: ofs, c1)
    double rho, prec, p, logNT;
    int minRegSize;
};

struct Misc {                 // per-frame scalars
    int maxS;                 // max gx^2+gy^2 over defined pixels
    int nDefined;
    int nSeg;
    int nCand;                // rectangles handed from the sequential core to the NFA stage
    int nKl;
    int overflow;
    long long cyc[8];         // master-wave cycle breakdown (debug): grow, rect, refine, nfa count, nfa math, seed scan
};

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
