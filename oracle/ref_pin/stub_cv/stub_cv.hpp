// ORACLE — TEST INFRASTRUCTURE ONLY (oracle/ref_pin).  Never included by the product.
//
// A minimal stand-in for the slice of the OpenCV 3.4 C++ API that the REFERENCE's in-repo front-end
// sources use, so that /root/reference/src/ORBextractor.cc (and line-range slices of src/ORBmatcher.cc,
// src/Frame.cc) compile UNMODIFIED, where they lie, in a container without OpenCV.  What this pins:
// every line of the reference's own code (constructor tables, pyramid driver, per-cell FAST driver,
// DivideNode / DistributeOctTree, IC_Angle, computeOrbDescriptor, operator(), SearchForInitialization,
// DescriptorDistance, ComputeThreeMaxima, GetFeaturesInArea, lineDescriptorMAD).  What it does NOT pin:
// the OpenCV leaves -- cv::FAST, cv::resize, cv::copyMakeBorder, cv::GaussianBlur, cv::fastAtan2 forward
// to oracle/cvleaf.h (UPSTREAM-RECALL restatements), cv::BFMatcher::knnMatch to a restatement below.
//
// Container semantics follow OpenCV where the reference depends on them:
//  * cv::Mat is a ref-counted header over shared storage; rowRange / colRange / operator()(Rect) are views;
//    clone() owns; create() is a no-op when size and type already match (resize / copyMakeBorder write
//    into the ROI of the padded pyramid buffer IN PLACE, src/ORBextractor.cc:1115-1122);
//  * `m = Mat::zeros(r,c,t)` assigns a MatExpr: create() + fill IN PLACE, it does not rebind the header
//    (src/ORBextractor.cc:1037 zeroes the rows of the caller's descriptor matrix through a row view);
//  * copyMakeBorder without BORDER_ISOLATED reads real pixels outside a ROI, with it (or on a whole
//    matrix) it reflects inside the source;
//  * cvRound = round-half-to-even (SSE2 cvtss2si / cvtsd2si), cvFloor / cvCeil as in fast_math.hpp.
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <cmath>
#include <cassert>
#include <cstdlib>
#include <memory>
#include <sstream>      // (the real opencv2/core/core.hpp pulls these in; the vendored DBoW2 relies on it)
#include <iostream>
#include <string>
#include <vector>
#include <string>
#include <algorithm>
#include "../../cvleaf.h"

typedef unsigned char uchar;
typedef unsigned short ushort;

#define CV_PI 3.1415926535897932384626433832795
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) (((depth) & 7) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32SC1 CV_MAKETYPE(CV_32S, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)
#define CV_MAT_DEPTH(t) ((t) & 7)
#define CV_MAT_CN(t) ((((t) >> CV_CN_SHIFT) & 511) + 1)

static inline int cvRound(double v) { return (int)std::lrint(v); }
static inline int cvRound(float v) { return (int)std::lrintf(v); }
static inline int cvRound(int v) { return v; }
static inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
static inline int cvFloor(float v) { int i = (int)v; return i - (i > v); }
static inline int cvFloor(int v) { return v; }
static inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }
static inline int cvCeil(float v) { int i = (int)v; return i + (i < v); }
static inline int cvCeil(int v) { return v; }

namespace cv {

template <class T> static inline T saturate_cast(double v) { return (T)v; }
template <> inline int saturate_cast<int>(double v) { return cvRound(v); }
template <> inline uchar saturate_cast<uchar>(double v) { int i = cvRound(v); return (uchar)(i < 0 ? 0 : i > 255 ? 255 : i); }

template <class T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
    template <class U> Point_(const Point_<U>& p) : x((T)p.x), y((T)p.y) {}
};
template <class T> static inline Point_<T>& operator*=(Point_<T>& a, float b) { a.x = (T)(a.x * b); a.y = (T)(a.y * b); return a; }
template <class T> static inline Point_<T>& operator*=(Point_<T>& a, double b) { a.x = (T)(a.x * b); a.y = (T)(a.y * b); return a; }
template <class T> static inline Point_<T>& operator*=(Point_<T>& a, int b) { a.x = (T)(a.x * b); a.y = (T)(a.y * b); return a; }
template <class T> static inline Point_<T> operator-(const Point_<T>& a, const Point_<T>& b) { return Point_<T>(a.x - b.x, a.y - b.y); }
template <class T> static inline Point_<T> operator+(const Point_<T>& a, const Point_<T>& b) { return Point_<T>(a.x + b.x, a.y + b.y); }
typedef Point_<int> Point2i;
typedef Point_<int> Point;
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;

template <class T> struct Size_ {
    T width, height;
    Size_() : width(0), height(0) {}
    Size_(T w, T h) : width(w), height(h) {}
};
typedef Size_<int> Size;

template <class T> struct Rect_ {
    T x, y, width, height;
    Rect_() : x(0), y(0), width(0), height(0) {}
    Rect_(T x_, T y_, T w, T h) : x(x_), y(y_), width(w), height(h) {}
};
typedef Rect_<int> Rect;

struct Range {
    int start, end;
    Range() : start(0), end(0) {}
    Range(int s, int e) : start(s), end(e) {}
    static Range all() { return Range(INT32_MIN, INT32_MAX); }
};

struct Scalar { double val[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; } };

class KeyPoint {
public:
    KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(Point2f p, float s, float a = -1, float r = 0, int o = 0, int c = -1) : pt(p), size(s), angle(a), response(r), octave(o), class_id(c) {}
    KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1) : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
    Point2f pt; float size; float angle; float response; int octave; int class_id;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint is 28 bytes");

struct DMatch {
    DMatch() : queryIdx(-1), trainIdx(-1), imgIdx(-1), distance(3.402823466e+38f) {}
    DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(-1), distance(d) {}
    int queryIdx, trainIdx, imgIdx; float distance;
    bool operator<(const DMatch& m) const { return distance < m.distance; }
};

struct MatStep {
    size_t p0;
    MatStep() : p0(0) {}
    MatStep(size_t s) : p0(s) {}
    operator size_t() const { return p0; }
    MatStep& operator=(size_t s) { p0 = s; return *this; }
};

class Mat;
struct MatExpr {                      // only the initialiser expressions the reference uses
    int rows, cols, type; double fill;
    operator Mat() const;
};

class Mat {
public:
    int flags = 0, rows = 0, cols = 0;
    uchar* data = nullptr;
    MatStep step;
    // whole allocation this header views (for locateROI / isSubmatrix)
    std::shared_ptr<std::vector<uchar>> owner;
    uchar* datastart = nullptr;
    int wholeRows = 0, wholeCols = 0;

    Mat() {}
    Mat(int r, int c, int t) { create(r, c, t); }
    Mat(Size s, int t) { create(s.height, s.width, t); }
    Mat(int r, int c, int t, const Scalar& s) { create(r, c, t); setTo(s); }
    Mat(int r, int c, int t, void* d, size_t st = 0) {           // user data, not owned
        flags = t; rows = r; cols = c; data = datastart = (uchar*)d; step = st ? st : (size_t)c * esz(t); wholeRows = r; wholeCols = c;
    }
    Mat(const Mat&) = default;
    Mat& operator=(const Mat&) = default;
    Mat& operator=(const MatExpr& e) { create(e.rows, e.cols, e.type); fillBytes(e.fill); return *this; }
    Mat(const MatExpr& e) { create(e.rows, e.cols, e.type); fillBytes(e.fill); }

    static size_t esz(int t) { static const int d[8] = {1, 1, 2, 2, 4, 4, 8, 0}; return (size_t)d[CV_MAT_DEPTH(t)] * CV_MAT_CN(t); }
    static MatExpr zeros(int r, int c, int t) { return MatExpr{r, c, t, 0.0}; }
    static MatExpr zeros(Size s, int t) { return MatExpr{s.height, s.width, t, 0.0}; }

    void create(int r, int c, int t) {
        if (data && rows == r && cols == c && type() == t) return;      // OpenCV: nothing happens when the header already fits
        owner = std::make_shared<std::vector<uchar>>((size_t)r * c * esz(t) + 64);
        flags = t; rows = r; cols = c; step = (size_t)c * esz(t);
        data = datastart = owner->data(); wholeRows = r; wholeCols = c;
    }
    void create(Size s, int t) { create(s.height, s.width, t); }
    void release() { owner.reset(); data = datastart = nullptr; rows = cols = 0; wholeRows = wholeCols = 0; }
    int type() const { return flags & 0xFFF; }
    int depth() const { return CV_MAT_DEPTH(flags); }
    int channels() const { return CV_MAT_CN(flags); }
    size_t elemSize() const { return esz(type()); }
    size_t elemSize1() const { return esz(type()) / channels(); }
    size_t step1() const { return (size_t)step / elemSize1(); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    size_t total() const { return (size_t)rows * cols; }
    Size size() const { return Size(cols, rows); }
    bool isContinuous() const { return rows <= 1 || (size_t)step == (size_t)cols * elemSize(); }
    bool isSubmatrix() const { return rows != wholeRows || cols != wholeCols; }
    void locateROI(Size& whole, Point& ofs) const {
        size_t delta = (size_t)(data - datastart), wstep = (size_t)step;
        ofs.y = wstep ? (int)(delta / wstep) : 0; ofs.x = (int)((delta - (size_t)ofs.y * wstep) / elemSize());
        whole = Size(wholeCols, wholeRows);
    }

    uchar* ptr(int r = 0) { return data + (size_t)r * step; }
    const uchar* ptr(int r = 0) const { return data + (size_t)r * step; }
    template <class T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
    template <class T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
    template <class T> T& at(int r, int c) { return ((T*)(data + (size_t)r * step))[c]; }
    template <class T> const T& at(int r, int c) const { return ((const T*)(data + (size_t)r * step))[c]; }
    template <class T> T& at(int i) { return rows == 1 ? ((T*)data)[i] : *(T*)(data + (size_t)i * step); }
    template <class T> const T& at(int i) const { return rows == 1 ? ((const T*)data)[i] : *(const T*)(data + (size_t)i * step); }

    Mat rowRange(int s, int e) const { Mat m(*this); m.data = data + (size_t)s * step; m.rows = e - s; return m; }
    Mat colRange(int s, int e) const { Mat m(*this); m.data = data + (size_t)s * elemSize(); m.cols = e - s; return m; }
    Mat rowRange(const Range& r) const { return rowRange(r.start, r.end); }
    Mat colRange(const Range& r) const { return colRange(r.start, r.end); }
    Mat row(int y) const { return rowRange(y, y + 1); }
    Mat col(int x) const { return colRange(x, x + 1); }
    Mat operator()(const Rect& r) const { return rowRange(r.y, r.y + r.height).colRange(r.x, r.x + r.width); }
    Mat operator()(Range rr, Range cr) const {
        Mat m = (rr.start == INT32_MIN) ? *this : rowRange(rr);
        return (cr.start == INT32_MIN) ? m : m.colRange(cr);
    }
    Mat clone() const { Mat m; copyTo(m); return m; }
    Mat t() const;
    double dot(const Mat& m) const;      // (leaf) CV_32F: products and sum in double, element order
    void copyTo(Mat& m) const {
        if (empty()) { m.release(); return; }
        m.create(rows, cols, type());
        for (int y = 0; y < rows; ++y) std::memmove(m.ptr(y), ptr(y), (size_t)cols * elemSize());
    }
    Mat& setTo(const Scalar& s) { fillBytes(s.val[0]); return *this; }
    Mat& operator=(const Scalar& s) { return setTo(s); }

private:
    void fillBytes(double v) {
        assert(v == 0.0 || depth() == CV_8U);
        for (int y = 0; y < rows; ++y) std::memset(ptr(y), (int)v, (size_t)cols * elemSize());
    }
};
inline MatExpr::operator Mat() const { Mat m; m = *this; return m; }
inline Mat mat_t(const Mat& a);
inline Mat Mat::t() const { return mat_t(*this); }

// --- the little matrix algebra the tracking matchers use on CV_32F poses (src/ORBmatcher.cc:1341-1349, 1364; src/LSDmatcher.cpp:25-45):
// A * B, A + B, -A, A.t().  In OpenCV these are MatExpr nodes folded into cv::gemm -- an un-vendored leaf; the stand-in evaluates in
// float32, row by row, ((a0 b0 + a1 b1) + a2 b2) (+ c): the operation order of shim/FrontendMatchers.h's stand-in path (RxPlusT).
inline Mat mat_t(const Mat& a) { Mat r(a.cols, a.rows, CV_32F); for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) r.at<float>(j, i) = a.at<float>(i, j); return r; }
inline Mat operator-(const Mat& a) { Mat r(a.rows, a.cols, CV_32F); for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) r.at<float>(i, j) = -a.at<float>(i, j); return r; }
inline Mat operator*(const Mat& a, const Mat& b) {
    assert(a.type() == CV_32F && b.type() == CV_32F && a.cols == b.rows);
    Mat r(a.rows, b.cols, CV_32F);
    for (int i = 0; i < a.rows; ++i) for (int j = 0; j < b.cols; ++j) {
        float acc = a.at<float>(i, 0) * b.at<float>(0, j);
        for (int k = 1; k < a.cols; ++k) acc = acc + a.at<float>(i, k) * b.at<float>(k, j);
        r.at<float>(i, j) = acc;
    }
    return r;
}
inline Mat operator+(const Mat& a, const Mat& b) {
    assert(a.type() == CV_32F && b.type() == CV_32F && a.rows == b.rows && a.cols == b.cols);
    Mat r(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) r.at<float>(i, j) = a.at<float>(i, j) + b.at<float>(i, j);
    return r;
}
inline Mat operator-(const Mat& a, const Mat& b) {
    assert(a.type() == CV_32F && b.type() == CV_32F && a.rows == b.rows && a.cols == b.cols);
    Mat r(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) r.at<float>(i, j) = a.at<float>(i, j) - b.at<float>(i, j);
    return r;
}
// A / s on CV_32F (src/ORBmatcher.cc:304-305: the Sim3 decomposition): a MatExpr scaled by alpha = 1 / s in double, every element (float)(a * alpha) (leaf, UPSTREAM-RECALL:
// modules/core/src/matop.cpp MatOp_AddEx::divide -> convertTo with a double alpha)
inline Mat operator/(const Mat& a, double s) {
    assert(a.type() == CV_32F);
    const double alpha = 1.0 / s;
    Mat r(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) r.at<float>(i, j) = (float)((double)a.at<float>(i, j) * alpha);
    return r;
}
// s * A on CV_32F (src/ORBmatcher.cc:1123-1124: sR12 = s12 * R12, sR21 = (1.0 / s12) * R12.t()): a MatExpr with alpha = s in double, every element (float)(a * alpha) (leaf, as A / s above)
inline Mat operator*(double s, const Mat& a) {
    assert(a.type() == CV_32F);
    Mat r(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) r.at<float>(i, j) = (float)((double)a.at<float>(i, j) * s);
    return r;
}
// Mat::dot and cv::norm(m) (NORM_L2) on CV_32F vectors as ORBmatcher::Fuse uses them (src/ORBmatcher.cc:878, 888; un-vendored leaves, UPSTREAM-RECALL:
// modules/core/src/matmul.cpp dotProd_32f / stat.cpp normL2_32f): every product and the running sum in double, element order
inline double Mat::dot(const Mat& m) const {
    assert(type() == CV_32F && m.type() == CV_32F && rows * cols == m.rows * m.cols);
    double r = 0;
    for (int i = 0; i < rows; ++i) for (int j = 0; j < cols; ++j) { const int k = i * cols + j; r += (double)at<float>(i, j) * (double)m.at<float>(k / m.cols, k % m.cols); }
    return r;
}
inline double norm(const Mat& a) {
    assert(a.type() == CV_32F);
    double s = 0;
    for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) s += (double)a.at<float>(i, j) * (double)a.at<float>(i, j);
    return std::sqrt(s);
}
// cv::norm(a, b, NORM_HAMMING) (src/MapLine.cpp:283): popcount of the xor, as a double (leaf)
inline double norm(const Mat& a, const Mat& b, int normType);
template <class T> class Mat_;
template <class T> struct MatCommaInitializer_ {
    Mat m; int idx;
    template <class U> MatCommaInitializer_& operator,(U v) { m.at<T>(idx / m.cols, idx % m.cols) = (T)v; ++idx; return *this; }
    operator Mat() const { return m; }
};
template <class T> class Mat_ : public Mat {
public:
    Mat_(int r, int c) : Mat(r, c, sizeof(T) == 4 ? CV_32F : CV_64F) {}
    template <class U> MatCommaInitializer_<T> operator<<(U v) { MatCommaInitializer_<T> ci{*this, 0}; return (ci, v); }
};

// --- proxies ---------------------------------------------------------------------------------------------------------------------
class _InputArray {
public:
    _InputArray() : m(nullptr) {}
    _InputArray(const Mat& mm) : m(const_cast<Mat*>(&mm)) {}
    Mat getMat(int = -1) const { return m ? *m : Mat(); }
    bool empty() const { return !m || m->empty(); }
protected:
    Mat* m;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray() {}
    _OutputArray(Mat& mm) { m = &mm; }
    void create(int r, int c, int t) const { assert(m); m->create(r, c, t); }
    void create(Size s, int t) const { assert(m); m->create(s, t); }
    void release() const { if (m) m->release(); }
    bool needed() const { return m != nullptr; }
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
typedef const _OutputArray& InputOutputArray;
static inline const _OutputArray& noArray() { static _OutputArray a; return a; }

enum { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT = 2, BORDER_WRAP = 3, BORDER_REFLECT_101 = 4, BORDER_REFLECT101 = 4,
       BORDER_DEFAULT = 4, BORDER_ISOLATED = 16 };
enum { INTER_NEAREST = 0, INTER_LINEAR = 1, INTER_CUBIC = 2, INTER_AREA = 3 };
enum { NORM_L2 = 4, NORM_HAMMING = 6 };

// --- cv::FileStorage / cv::FileNode: the vendored DBoW2 (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1456-1640) has YAML save / load members that must
// compile because they are virtual; the reference never calls them (the vocabulary comes from ORBvoc.txt through loadFromTextFile).  Inert stand-ins:
// a storage never opens, so save(filename) / load(filename) throw as they do on a missing file.
struct FileNode {
    FileNode operator[](const char*) const { return FileNode(); }
    FileNode operator[](const std::string&) const { return FileNode(); }
    FileNode operator[](int) const { return FileNode(); }
    size_t size() const { return 0; }
    operator int() const { return 0; }
    operator float() const { return 0.f; }
    operator double() const { return 0.0; }
    operator std::string() const { return std::string(); }
};
struct FileStorage {
    enum { READ = 0, WRITE = 1 };
    FileStorage() {}
    FileStorage(const std::string&, int) {}
    bool isOpened() const { return false; }
    FileNode operator[](const char*) const { return FileNode(); }
    FileNode operator[](const std::string&) const { return FileNode(); }
};
template <class T> static inline FileStorage& operator<<(FileStorage& fs, const T&) { return fs; }

inline double norm(const Mat& a, const Mat& b, int normType) {
    assert(normType == NORM_HAMMING && a.type() == CV_8UC1 && a.rows == b.rows && a.cols == b.cols);
    long d = 0;
    for (int y = 0; y < a.rows; ++y) for (int x = 0; x < a.cols; ++x) d += __builtin_popcount((unsigned)(a.ptr(y)[x] ^ b.ptr(y)[x]));
    return (double)d;
}

// --- leaves (UPSTREAM-RECALL, oracle/cvleaf.h) -------------------------------------------------------------------------------------
static inline float fastAtan2(float y, float x) { return orc::fast_atan2(y, x); }

namespace stubdetail {
static inline orc::Img8 toImg(const Mat& m) {
    assert(m.type() == CV_8UC1);
    orc::Img8 o(m.cols, m.rows);
    for (int y = 0; y < m.rows; ++y) std::memcpy(o.row(y), m.ptr(y), (size_t)m.cols);
    return o;
}
static inline void fromImg(const orc::Img8& s, Mat& m) {
    m.create(s.h, s.w, CV_8UC1);
    for (int y = 0; y < s.h; ++y) std::memcpy(m.ptr(y), s.row(y), (size_t)s.w);
}
}  // namespace stubdetail

// cv::FAST(image, keypoints, threshold, nonmaxSuppression): TYPE_9_16; KeyPoint(x, y, 7.f, -1, score) in raster order
static inline void FAST(InputArray image, std::vector<KeyPoint>& kps, int threshold, bool nonmaxSuppression = true) {
    Mat m = image.getMat();
    assert(nonmaxSuppression && "the reference always passes nonmaxSuppression = true");
    orc::Img8 im = stubdetail::toImg(m);
    std::vector<orc::FastKp> out;
    orc::fast9_view(im, 0, 0, im.w, im.h, threshold, out);
    kps.clear();
    for (const orc::FastKp& k : out) kps.push_back(KeyPoint((float)k.x, (float)k.y, 7.f, -1, (float)k.score));
}

static inline void resize(InputArray src, OutputArray dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR) {
    Mat s = src.getMat();
    assert(interpolation == INTER_LINEAR && fx == 0 && fy == 0 && dsize.width > 0 && dsize.height > 0);
    orc::Img8 r = orc::resize_linear_8u(stubdetail::toImg(s), dsize.width, dsize.height);
    dst.create(dsize.height, dsize.width, s.type());          // no-op when dst is already a fitting ROI: the write below lands in place
    Mat d = dst.getMat();
    for (int y = 0; y < r.h; ++y) std::memcpy(d.ptr(y), r.row(y), (size_t)r.w);
}

// cv::copyMakeBorder (copy.cpp): a ROI source is widened into its parent as far as pixels exist unless BORDER_ISOLATED is set
static inline void copyMakeBorder(InputArray src_, OutputArray dst_, int top, int bottom, int left, int right, int borderType, const Scalar& = Scalar()) {
    Mat src = src_.getMat();
    assert(src.type() == CV_8UC1 && (borderType & ~BORDER_ISOLATED) == BORDER_REFLECT_101);
    if (src.isSubmatrix() && !(borderType & BORDER_ISOLATED)) {
        Size whole; Point ofs; src.locateROI(whole, ofs);
        int dt = std::min(ofs.y, top), db = std::min(whole.height - src.rows - ofs.y, bottom);
        int dl = std::min(ofs.x, left), dr = std::min(whole.width - src.cols - ofs.x, right);
        src.data -= (size_t)dt * src.step + (size_t)dl; src.rows += dt + db; src.cols += dl + dr;
        top -= dt; bottom -= db; left -= dl; right -= dr;
    }
    const int sw = src.cols, sh = src.rows, dw = sw + left + right, dh = sh + top + bottom;
    dst_.create(dh, dw, src.type());
    Mat dst = dst_.getMat();
    std::vector<uchar> rowbuf((size_t)dw);
    for (int y = 0; y < sh; ++y) {                               // interior rows first (in place when dst's centre IS src)
        const uchar* s = src.ptr(y);
        for (int x = 0; x < dw; ++x) rowbuf[x] = s[orc::reflect101(x - left, sw)];
        std::memcpy(dst.ptr(y + top), rowbuf.data(), (size_t)dw);
    }
    for (int y = 0; y < top; ++y) std::memcpy(dst.ptr(y), dst.ptr(top + orc::reflect101(y - top, sh)), (size_t)dw);
    for (int y = 0; y < bottom; ++y) std::memcpy(dst.ptr(top + sh + y), dst.ptr(top + orc::reflect101(sh + y, sh)), (size_t)dw);
}

// SSLAM_STUB_GAUSS_340=1 switches this leaf to decision D6's alternative (OpenCV 3.4.0's rounded taps, oracle/cvleaf.h gauss_taps_340)
static inline void GaussianBlur(InputArray src_, OutputArray dst_, Size ksize, double sigmaX, double sigmaY = 0, int borderType = BORDER_DEFAULT) {
    Mat src = src_.getMat();
    assert(ksize.width == ksize.height && (sigmaY == 0 || sigmaY == sigmaX) && (borderType & ~BORDER_ISOLATED) == BORDER_REFLECT_101);
    assert(!src.isSubmatrix() || (borderType & BORDER_ISOLATED));   // a ROI would read its parent's pixels: the reference blurs a clone
    static const bool v340 = std::getenv("SSLAM_STUB_GAUSS_340") != nullptr;
    orc::Img8 r = orc::gaussian_blur_8u(stubdetail::toImg(src), ksize.width, sigmaX, v340 ? 1 : 0);
    dst_.create(src.rows, src.cols, src.type());
    Mat d = dst_.getMat();
    for (int y = 0; y < r.h; ++y) std::memcpy(d.ptr(y), r.row(y), (size_t)r.w);
}

struct KeyPointsFilter {       // only reachable from the dead ComputeKeyPointsOld (src/ORBextractor.cc:855-1032, call commented out at :1057)
    static void retainBest(std::vector<KeyPoint>& k, int n) {
        if (n >= 0 && (int)k.size() > n) {
            std::stable_sort(k.begin(), k.end(), [](const KeyPoint& a, const KeyPoint& b) { return a.response > b.response; });
            k.resize(n);
        }
    }
};

}  // namespace cv
