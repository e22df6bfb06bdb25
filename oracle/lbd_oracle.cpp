// ORACLE — TEST INFRASTRUCTURE ONLY (see cvleaf.h header).  PARITY UNPINNED.
//
// CPU restatement of cv::line_descriptor::BinaryDescriptor::compute (LBD, Zhang &
// Koch 2013 as implemented in opencv_contrib line_descriptor/src/binary_descriptor.cpp;
// SURVEY.md Appendix A.9) as called at reference src/ExtractLineSegment.cpp:53, and
// of LineSegment::ExtractLineSegment itself (src/ExtractLineSegment.cpp:18-69,
// sort comparator include/auxiliar.h:69-74).  Not in /root/reference (un-vendored).
// Defaults: numOfOctave_ 1, widthOfBand_ 7, NUM_OF_BANDS 9, 5x5 sigma-1 blur before
// Sobel.  Decisions: D3 top-N uses a stable sort; D4 no FMA; D5 cosf/sinf.
#include "oracle.h"
#include "cvleaf.h"
#include "lines_types.h"

namespace orc {
int g_lbdBitOrder = 1;      // decision D12 (see the pack step below; 1 = the default since round 5); orc_set_lbd_bit_order(), library: sslam_lines_set_lbd_bit_order()

static const int NUM_OF_BANDS = 9, WIDTH_OF_BAND = 7;

static const int kCombinations[32][2] = {
    {0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {0, 6}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6}, {2, 3}, {2, 4}, {2, 5}, {2, 6}, {2, 7},
    {2, 8}, {3, 4}, {3, 5}, {3, 6}, {3, 7}, {3, 8}, {4, 5}, {4, 6}, {4, 7}, {4, 8}, {5, 6}, {5, 7}, {5, 8}, {6, 7}, {6, 8}, {7, 8}};

void lbd_compute(const Img8& image, const std::vector<KeyLine>& keylines, std::vector<uint8_t>& desc, std::vector<float>* float_desc) {
    const int n = (int)keylines.size();
    desc.assign((size_t)n * 32, 0);
    if (float_desc) float_desc->assign((size_t)n * 72, 0.f);
    if (n == 0) return;
    // BinaryDescriptor::computeGaussianPyramid: 5x5 sigma 1 blur of the base image, then Sobel 3x3 -> s16
    Img8 blurred = gaussian_blur_8u(image, 5, 1.0, g_gaussVariant);      // (variant 1: taps 14 63 103 63 14 instead of 14 62 104 62 14)
    std::vector<int16_t> dxImg, dyImg;
    sobel3_s16(blurred, dxImg, dyImg);
    // Gaussian weights (BinaryDescriptor ctor; integer divisions are the library's)
    double gaussCoefL[WIDTH_OF_BAND * 3], gaussCoefG[NUM_OF_BANDS * WIDTH_OF_BAND];
    {
        double u = (WIDTH_OF_BAND * 3 - 1) / 2;
        double sigma = (WIDTH_OF_BAND * 2 + 1) / 2;
        double invsigma2 = -1 / (2 * sigma * sigma);
        for (int i = 0; i < WIDTH_OF_BAND * 3; ++i) { double dis = i - u; gaussCoefL[i] = std::exp(dis * dis * invsigma2); }
        u = (NUM_OF_BANDS * WIDTH_OF_BAND - 1) / 2;
        sigma = u;
        invsigma2 = -1 / (2 * sigma * sigma);
        for (int i = 0; i < NUM_OF_BANDS * WIDTH_OF_BAND; ++i) { double dis = i - u; gaussCoefG[i] = std::exp(dis * dis * invsigma2); }
    }
    const short heightOfLSP = (short)(WIDTH_OF_BAND * NUM_OF_BANDS);
    const short halfHeight = (short)((heightOfLSP - 1) / 2);
    const short realWidth = (short)image.w, imageWidth = (short)(realWidth - 1), imageHeight = (short)(image.h - 1);
    for (int li = 0; li < n; ++li) {
        const KeyLine& kl = keylines[li];
        float pgdLBandSum[NUM_OF_BANDS] = {0}, ngdLBandSum[NUM_OF_BANDS] = {0}, pgdL2BandSum[NUM_OF_BANDS] = {0}, ngdL2BandSum[NUM_OF_BANDS] = {0};
        float pgdOBandSum[NUM_OF_BANDS] = {0}, ngdOBandSum[NUM_OF_BANDS] = {0}, pgdO2BandSum[NUM_OF_BANDS] = {0}, ngdO2BandSum[NUM_OF_BANDS] = {0};
        const short lengthOfLSP = (short)kl.numOfPixels;
        const short halfWidth = (short)((lengthOfLSP - 1) / 2);
        const float lineMiddlePointX = (float)(0.5 * (kl.sPointInOctaveX + kl.ePointInOctaveX));
        const float lineMiddlePointY = (float)(0.5 * (kl.sPointInOctaveY + kl.ePointInOctaveY));
        float dL[2], dO[2];
        dL[0] = cr_cosf(kl.angle); dL[1] = cr_sinf(kl.angle);      // D5
        dO[0] = -dL[1]; dO[1] = dL[0];
        float sCorX0 = -dL[0] * halfWidth + dL[1] * halfHeight + lineMiddlePointX;
        float sCorY0 = -dL[1] * halfWidth - dL[0] * halfHeight + lineMiddlePointY;
        for (short hID = 0; hID < heightOfLSP; ++hID) {
            float sCorX = sCorX0, sCorY = sCorY0;
            float pgdLRowSum = 0, ngdLRowSum = 0, pgdORowSum = 0, ngdORowSum = 0;
            for (short wID = 0; wID < lengthOfLSP; ++wID) {
                short tempCor = (short)std::round(sCorX);
                short xCor = (tempCor < 0) ? 0 : (tempCor > imageWidth) ? imageWidth : tempCor;
                tempCor = (short)std::round(sCorY);
                short yCor = (tempCor < 0) ? 0 : (tempCor > imageHeight) ? imageHeight : tempCor;
                short dx = dxImg[(size_t)yCor * realWidth + xCor], dy = dyImg[(size_t)yCor * realWidth + xCor];
                float gDL = dx * dL[0] + dy * dL[1];
                float gDO = dx * dO[0] + dy * dO[1];
                if (gDL > 0) pgdLRowSum += gDL; else ngdLRowSum -= gDL;
                if (gDO > 0) pgdORowSum += gDO; else ngdORowSum -= gDO;
                sCorX += dL[0]; sCorY += dL[1];
            }
            sCorX0 -= dL[1]; sCorY0 += dL[0];
            float coefInGaussion = (float)gaussCoefG[hID];
            pgdLRowSum = coefInGaussion * pgdLRowSum; ngdLRowSum = coefInGaussion * ngdLRowSum;
            float pgdL2RowSum = pgdLRowSum * pgdLRowSum, ngdL2RowSum = ngdLRowSum * ngdLRowSum;
            pgdORowSum = coefInGaussion * pgdORowSum; ngdORowSum = coefInGaussion * ngdORowSum;
            float pgdO2RowSum = pgdORowSum * pgdORowSum, ngdO2RowSum = ngdORowSum * ngdORowSum;
            auto accumulate = [&](short bandID, float c) {
                pgdLBandSum[bandID] += c * pgdLRowSum; ngdLBandSum[bandID] += c * ngdLRowSum;
                pgdL2BandSum[bandID] += c * c * pgdL2RowSum; ngdL2BandSum[bandID] += c * c * ngdL2RowSum;
                pgdOBandSum[bandID] += c * pgdORowSum; ngdOBandSum[bandID] += c * ngdORowSum;
                pgdO2BandSum[bandID] += c * c * pgdO2RowSum; ngdO2BandSum[bandID] += c * c * ngdO2RowSum;
            };
            short bandID = (short)(hID / WIDTH_OF_BAND);
            accumulate(bandID, (float)gaussCoefL[hID % WIDTH_OF_BAND + WIDTH_OF_BAND]);
            bandID--;
            if (bandID >= 0) accumulate(bandID, (float)gaussCoefL[hID % WIDTH_OF_BAND + 2 * WIDTH_OF_BAND]);
            bandID = (short)(bandID + 2);
            if (bandID < NUM_OF_BANDS) accumulate(bandID, (float)gaussCoefL[hID % WIDTH_OF_BAND]);
        }
        float desVec[NUM_OF_BANDS * 8];
        const float invN2 = (float)(1.0 / (WIDTH_OF_BAND * 2.0)), invN3 = (float)(1.0 / (WIDTH_OF_BAND * 3.0));
        for (short bandID = 0; bandID < NUM_OF_BANDS; ++bandID) {
            float invN = (bandID == 0 || bandID == NUM_OF_BANDS - 1) ? invN2 : invN3;
            short desID = (short)(bandID * 8);
            float temp = pgdLBandSum[bandID] * invN;
            desVec[desID] = temp; desVec[desID + 4] = std::sqrt(pgdL2BandSum[bandID] * invN - temp * temp);
            temp = ngdLBandSum[bandID] * invN;
            desVec[desID + 1] = temp; desVec[desID + 5] = std::sqrt(ngdL2BandSum[bandID] * invN - temp * temp);
            temp = pgdOBandSum[bandID] * invN;
            desVec[desID + 2] = temp; desVec[desID + 6] = std::sqrt(pgdO2BandSum[bandID] * invN - temp * temp);
            temp = ngdOBandSum[bandID] * invN;
            desVec[desID + 3] = temp; desVec[desID + 7] = std::sqrt(ngdO2BandSum[bandID] * invN - temp * temp);
        }
        float tempM = 0, tempS = 0;
        for (int b = 0; b < NUM_OF_BANDS; ++b) {
            const float* d = desVec + b * 8;
            tempM += d[0] * d[0]; tempM += d[1] * d[1]; tempM += d[2] * d[2]; tempM += d[3] * d[3];
            tempS += d[4] * d[4]; tempS += d[5] * d[5]; tempS += d[6] * d[6]; tempS += d[7] * d[7];
        }
        tempM = 1 / std::sqrt(tempM); tempS = 1 / std::sqrt(tempS);
        for (int b = 0; b < NUM_OF_BANDS; ++b) {
            float* d = desVec + b * 8;
            d[0] *= tempM; d[1] *= tempM; d[2] *= tempM; d[3] *= tempM;
            d[4] *= tempS; d[5] *= tempS; d[6] *= tempS; d[7] *= tempS;
        }
        for (int i = 0; i < NUM_OF_BANDS * 8; ++i) if (desVec[i] > 0.4f) desVec[i] = 0.4f;
        float temp = 0;
        for (int i = 0; i < NUM_OF_BANDS * 8; ++i) temp += desVec[i] * desVec[i];
        temp = 1 / std::sqrt(temp);
        for (int i = 0; i < NUM_OF_BANDS * 8; ++i) desVec[i] = desVec[i] * temp;
        if (float_desc) std::memcpy(float_desc->data() + (size_t)li * 72, desVec, sizeof(desVec));
        uint8_t* row = &desc[(size_t)li * 32];
        for (int comb = 0; comb < 32; ++comb) {
            const float* f1 = &desVec[8 * kCombinations[comb][0]];
            const float* f2 = &desVec[8 * kCombinations[comb][1]];
            uint8_t result = 0;
            // decision D12 -- 1 (DEFAULT since round 5): BinaryDescriptor::binaryConversion as two independent recollections have it, `result += (uchar)(0x80 >> i)`
            // (MSB first; UPSTREAM-RECALL); 0: bit i of the byte = comparison i (LSB first, the default of rounds 1-4).  Hamming distances -- every matcher result --
            // are the same under both.
            for (int i = 0; i < 8; ++i) if (f1[i] > f2[i]) result += (uint8_t)(g_lbdBitOrder == 1 ? (0x80 >> i) : (1 << i));
            row[comb] = result;
        }
    }
}

}  // namespace orc

using namespace orc;

extern "C" {

// LineSegment::ExtractLineSegment (src/ExtractLineSegment.cpp:18-69) with the cap as a parameter
// (the reference hard-codes 40, :42).  Returns the number of lines; keylines are 68-byte KeyLine
// records, ldesc n x 32, linefn n x 3 doubles.  raw_segments (optional) receives every LSD segment.
int orc_lines_extract(const uint8_t* gray, int w, int h, int stride, int max_lines, void* kl_out, uint8_t* ldesc_out,
                      double* linefn_out, int cap, float* raw_segments, int raw_cap, int* raw_n, float* float_desc_out) {
    Img8 im(w, h);
    for (int y = 0; y < h; ++y) std::memcpy(im.row(y), gray + (size_t)y * stride, w);
    std::vector<KeyLine> kls; std::vector<Seg4f> raw;
    lsd_detect_keylines(im, kls, &raw);
    if (raw_n) *raw_n = (int)raw.size();
    if (raw_segments && !raw.empty()) std::memcpy(raw_segments, raw.data(), sizeof(Seg4f) * std::min<size_t>(raw.size(), raw_cap));
    if ((int)kls.size() > max_lines) {
        std::stable_sort(kls.begin(), kls.end(), [](const KeyLine& a, const KeyLine& b) { return a.response > b.response; });   // D3
        kls.resize(max_lines);
        for (int i = 0; i < max_lines; ++i) kls[i].class_id = i;
    }
    std::vector<uint8_t> desc; std::vector<float> fdesc;
    lbd_compute(im, kls, desc, float_desc_out ? &fdesc : nullptr);
    int n = std::min((int)kls.size(), cap);
    if (n > 0) {      // (memcpy from an empty vector's null data() is undefined even for 0 bytes: tests/test_sanitize_cpu.py)
        std::memcpy(kl_out, kls.data(), sizeof(KeyLine) * n);
        std::memcpy(ldesc_out, desc.data(), (size_t)n * 32);
        if (float_desc_out) std::memcpy(float_desc_out, fdesc.data(), sizeof(float) * 72 * n);
    }
    for (int i = 0; i < n; ++i) {          // :56-68, Eigen Vector3d cross product in double
        double sx = kls[i].startPointX, sy = kls[i].startPointY, ex = kls[i].endPointX, ey = kls[i].endPointY;
        double l0 = sy * 1.0 - 1.0 * ey, l1 = 1.0 * ex - sx * 1.0, l2 = sx * ey - sy * ex;
        double nrm = std::sqrt(l0 * l0 + l1 * l1);
        linefn_out[i * 3] = l0 / nrm; linefn_out[i * 3 + 1] = l1 / nrm; linefn_out[i * 3 + 2] = l2 / nrm;
    }
    return (int)kls.size();
}

int orc_set_lbd_bit_order(int v) { const int old = g_lbdBitOrder; g_lbdBitOrder = v == 1 ? 1 : 0; return old; }

int orc_lsd_scaled(const uint8_t* gray, int w, int h, int stride, uint8_t* out, int* ow, int* oh) {
    Img8 im(w, h);
    for (int y = 0; y < h; ++y) std::memcpy(im.row(y), gray + (size_t)y * stride, w);
    Img8 s; lsd_debug_scaled(im, s);
    *ow = s.w; *oh = s.h;
    if (out) std::memcpy(out, s.d.data(), s.d.size());
    return 0;
}

}  // extern "C"
