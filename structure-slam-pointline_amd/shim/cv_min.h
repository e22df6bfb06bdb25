// Minimal stand-ins for the OpenCV / Eigen types on the drop-in boundary, used ONLY when the real
// headers are absent (this build image has no OpenCV/Eigen).  Field order and sizes equal the real
// types (cv::KeyPoint 28 B, cv::line_descriptor::KeyLine 68 B, Eigen::Vector3d 24 B), so code
// compiled against the real headers links against the same shim bodies unchanged.
#pragma once
#if defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>) && __has_include(<opencv2/line_descriptor/descriptor.hpp>)
#define SSLAM_HAVE_OPENCV 1
#endif
#if __has_include(<eigen3/Eigen/Core>)
#define SSLAM_HAVE_EIGEN 1
#endif
#endif

#ifdef SSLAM_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <opencv2/line_descriptor/descriptor.hpp>
#else
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <memory>
#include <vector>
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
namespace cv {
struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float a, float b) : x(a), y(b) {} };
struct KeyPoint {            // opencv2/core/types.hpp
    Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
};
class Mat {                  // the subset of cv::Mat the front-end boundary touches: single channel, 8-bit (images, descriptors) or 32-bit float (poses, points)
public:
    int rows = 0, cols = 0; size_t step = 0; uint8_t* data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void* ext, size_t st = 0) : rows(r), cols(c), step(st ? st : (size_t)c * esz(type)), data((uint8_t*)ext), type_(type) {}
    void create(int r, int c, int type) {
        if (r == rows && c == cols && type == type_ && own_) return;
        own_.reset(new uint8_t[(size_t)r * c * esz(type) + 1]); data = own_.get(); rows = r; cols = c; type_ = type; step = (size_t)c * esz(type);
    }
    void release() { own_.reset(); data = nullptr; rows = cols = 0; step = 0; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    bool isContinuous() const { return step == (size_t)cols * esz(type_); }
    uint8_t* ptr(int r = 0) { return data + (size_t)r * step; }
    const uint8_t* ptr(int r = 0) const { return data + (size_t)r * step; }
    template <class T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
    template <class T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
    template <class T> T& at(int r, int c = 0) { return ptr<T>(r)[c]; }
    template <class T> const T& at(int r, int c = 0) const { return ptr<T>(r)[c]; }
    Mat row(int r) const { Mat m(1, cols, type_, (void*)ptr(r), step); m.keep_ = own_; return m; }
private:
    static size_t esz(int type) { return type == CV_32F ? 4 : 1; }
    int type_ = CV_8U;
    std::shared_ptr<uint8_t[]> own_, keep_;
};
typedef const Mat& InputArray;
typedef Mat& OutputArray;
namespace line_descriptor {
struct KeyLine {             // opencv2/line_descriptor/descriptor.hpp
    float angle; int class_id; int octave; Point2f pt; float response; float size;
    float startPointX, startPointY, endPointX, endPointY;
    float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
    float lineLength; int numOfPixels;
};
}  // namespace line_descriptor
}  // namespace cv
#endif

#ifdef SSLAM_HAVE_EIGEN
#include <eigen3/Eigen/Core>
namespace sslam_shim { typedef Eigen::Vector3d Vector3d; }
#else
namespace sslam_shim { struct Vector3d { double v[3]; double operator()(int i) const { return v[i]; } double& operator()(int i) { return v[i]; } }; }
#endif

static_assert(sizeof(cv::KeyPoint) == 28, "cv::KeyPoint layout");
static_assert(sizeof(cv::line_descriptor::KeyLine) == 68, "KeyLine layout");
static_assert(sizeof(sslam_shim::Vector3d) == 24, "Vector3d layout");
