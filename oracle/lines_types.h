// ORACLE — TEST INFRASTRUCTURE ONLY.  Shared PODs of the line front-end restatement.
#pragma once
#include <cstdint>
#include <vector>
namespace orc {
struct Seg4f { float x1, y1, x2, y2; };
// cv::line_descriptor::KeyLine field order (opencv2/line_descriptor/descriptor.hpp), 68 bytes
struct KeyLine {
    float angle; int32_t class_id, octave; float pt_x, pt_y, response, size;
    float startPointX, startPointY, endPointX, endPointY;
    float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
    float lineLength; int32_t numOfPixels;
};
static_assert(sizeof(KeyLine) == 68, "KeyLine layout");
struct Img8;
extern int g_gaussVariant;      // orb_oracle.cpp: which 8-bit GaussianBlur the path uses (decision D6 / its OpenCV-3.4.0 alternative)
void lsd_detect_keylines(const Img8& image, std::vector<KeyLine>& keylines, std::vector<Seg4f>* raw);
void lsd_debug_scaled(const Img8& image, Img8& scaled_out);
void lbd_compute(const Img8& image, const std::vector<KeyLine>& keylines, std::vector<uint8_t>& desc, std::vector<float>* float_desc);
}  // namespace orc
