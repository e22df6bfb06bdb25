// Host side of the drop-in: the reference's class surfaces implemented over the C ABI.
// Plain C++ (g++), links libsslam_frontend.so; no HIP types.
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include "../../include/sslam_frontend.h"
#include "ORBextractor.h"
#include "ExtractLineSegment.h"
#include "FrontendMatchers.h"

namespace {
struct Global {
    std::mutex mu, linesMu;                  // linesMu: held for the duration of a line extraction and by SetMaxLines
    sslam_ctx* ctx = nullptr;
    sslam_lines* lines = nullptr;
    int maxLines = 40;                       // reference cap, src/ExtractLineSegment.cpp:42
    sslam_ctx* get() {
        std::lock_guard<std::mutex> lk(mu);
        if (!ctx) {
            const char* d = std::getenv("SSLAM_DEVICE");
            int rc = sslam_ctx_create(d ? std::atoi(d) : 0, &ctx);
            if (rc != SSLAM_OK) { std::fprintf(stderr, "sslam front-end: %s (%s)\n", sslam_status_str(rc), sslam_last_error()); throw std::runtime_error("sslam front-end: no usable GPU (no CPU fallback)"); }
        }
        return ctx;
    }
    sslam_lines* getLines() {
        sslam_ctx* c = get();
        std::lock_guard<std::mutex> lk(mu);
        if (!lines) {
            if (sslam_lines_create(c, maxLines, &lines) != SSLAM_OK) throw std::runtime_error(sslam_last_error());
            // the other leaves of the line path whose upstream form is a stated decision (include/sslam_frontend.h): one environment variable each, so
            // that the day `make -C oracle/ref_pin pin` first runs against a real OpenCV no recompilation stands between a maintainer and the matching variant
            struct { const char* env; int (*set)(sslam_lines*, int); } knobs[] = {
                {"SSLAM_LSD_NFA_VARIANT", sslam_lines_set_nfa_variant}, {"SSLAM_LBD_BIT_ORDER", sslam_lines_set_lbd_bit_order},
                {"SSLAM_LSD_RESIZE_VARIANT", sslam_lines_set_resize_variant}, {"SSLAM_LSD_SEED_ORDER", sslam_lines_set_seed_order}};
            for (auto& k : knobs) if (const char* e = std::getenv(k.env)) if (k.set(lines, std::atoi(e)) != SSLAM_OK) throw std::runtime_error(sslam_last_error());
            if (const char* e = std::getenv("SSLAM_ORB_BLUR_VARIANT")) if (sslam_lines_set_blur_variant(lines, std::atoi(e)) != SSLAM_OK) throw std::runtime_error(sslam_last_error());      // one knob for both extractors: the OpenCV release behind the reference build
        }
        return lines;
    }
} G;
void check(int rc) { if (rc != SSLAM_OK) throw std::runtime_error(std::string(sslam_status_str(rc)) + ": " + sslam_last_error()); }
}  // namespace

namespace StructureSLAM
{
ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST), mHandle(nullptr)
{
    check(sslam_orb_create(G.get(), nfeatures, _scaleFactor, nlevels, iniThFAST, minThFAST, &mHandle));
    // the class keeps the reference's constructor signature, so the one leaf whose arithmetic depends on the OpenCV release the system was built against
    // is chosen by the environment: SSLAM_ORB_BLUR_VARIANT=1 = OpenCV 3.4.0's 8-bit GaussianBlur (include/sslam_frontend.h, sslam_orb_set_blur_variant)
    if (const char* e = std::getenv("SSLAM_ORB_BLUR_VARIANT")) check(sslam_orb_set_blur_variant(mHandle, std::atoi(e)));
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    check(sslam_orb_get_scales(mHandle, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(), nullptr));
    mvImagePyramid.resize(nlevels);
}
ORBextractor::~ORBextractor() { sslam_orb_destroy(mHandle); }

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray, std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors)
{
#ifdef SSLAM_HAVE_OPENCV
    if (_image.empty()) return;
    cv::Mat image = _image.getMat();
#else
    const cv::Mat& image = _image;
    if (image.empty()) return;                       // outputs untouched, src/ORBextractor.cc:1046-1047
#endif
    const int cap = sslam_orb_max_keypoints(mHandle);
    mStageKeys.resize(cap);
    std::vector<uint8_t> desc((size_t)cap * 32);
    int n = 0;
    check(sslam_orb_extract(mHandle, image.data, image.cols, image.rows, (size_t)image.step, (sslam_keypoint*)mStageKeys.data(), desc.data(), cap, &n));
    _keypoints.assign(mStageKeys.begin(), mStageKeys.begin() + n);
    if (n == 0) { _descriptors.release(); return; }  // :1064-1065
    _descriptors.create(n, 32, CV_8U);               // :1068
#ifdef SSLAM_HAVE_OPENCV
    cv::Mat d = _descriptors.getMat();
#else
    cv::Mat& d = _descriptors;
#endif
    for (int i = 0; i < n; ++i) std::memcpy(d.ptr(i), &desc[(size_t)i * 32], 32);
}

LineSegment::LineSegment() {}
void LineSegment::SetMaxLines(int n) {
    std::lock_guard<std::mutex> lk2(G.linesMu);
    std::lock_guard<std::mutex> lk(G.mu);
    if (n != G.maxLines && G.lines) { sslam_lines_destroy(G.lines); G.lines = nullptr; }
    G.maxLines = n;
}
void LineSegment::ExtractLineSegment(const cv::Mat &img, std::vector<cv::line_descriptor::KeyLine> &keylines, cv::Mat &ldesc,
                                     std::vector<sslam_shim::Vector3d> &keylineFunctions, int, int)
{
    // the handle is shared by every (never constructed) LineSegment; an extraction holds the lock that SetMaxLines needs to replace it
    G.getLines();
    std::lock_guard<std::mutex> lk(G.linesMu);
    sslam_lines* L = G.getLines();
    const int cap = G.maxLines;
    keylines.resize(cap);
    std::vector<uint8_t> d((size_t)cap * 32);
    std::vector<double> fn((size_t)cap * 3);
    int n = 0;
    int rc = sslam_lines_extract(L, img.data, img.cols, img.rows, (size_t)img.step, (sslam_keyline*)keylines.data(), d.data(), fn.data(), cap, &n);
    if (rc == SSLAM_ERR_UNSUPPORTED) {               // an image LSD cannot take (smaller than 10x10, > 8192 rectangles): the reference's OpenCV would throw; 0 lines, said aloud
        std::fprintf(stderr, "sslam front-end: ExtractLineSegment: %s -- returning no lines\n", sslam_last_error());
        n = 0;
    } else if (rc != SSLAM_OK) {                     // a GPU fault must not degrade tracking silently: same behaviour as the ORB path
        keylines.clear();
        check(rc);
    }
    keylines.resize(n);
    if (n > 0) { ldesc.create(n, 32, CV_8U); for (int i = 0; i < n; ++i) std::memcpy(ldesc.ptr(i), &d[(size_t)i * 32], 32); }
    for (int i = 0; i < n; ++i) { sslam_shim::Vector3d v; v(0) = fn[i * 3]; v(1) = fn[i * 3 + 1]; v(2) = fn[i * 3 + 2]; keylineFunctions.push_back(v); }
}
}  // namespace StructureSLAM

namespace sslam_shim
{
static std::vector<uint8_t> rows32(const cv::Mat& m) {
    std::vector<uint8_t> out((size_t)m.rows * 32);
    for (int i = 0; i < m.rows; ++i) std::memcpy(&out[(size_t)i * 32], m.ptr(i), 32);
    return out;
}
int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) {
    const uint32_t* pa = (const uint32_t*)a.ptr(0); const uint32_t* pb = (const uint32_t*)b.ptr(0);
    int dist = 0;
    for (int i = 0; i < 8; ++i) dist += __builtin_popcount(pa[i] ^ pb[i]);
    return dist;
}
int SearchForInitialization(const std::vector<cv::KeyPoint> &k1, const cv::Mat &d1, const std::vector<cv::KeyPoint> &k2, const cv::Mat &d2,
                            const float bounds[4], std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12,
                            int windowSize, float nnratio, bool checkOri) {
    vnMatches12.assign(k1.size(), -1);
    if (k1.empty()) return 0;
    std::vector<uint8_t> a = rows32(d1), b = rows32(d2);
    int n = 0;
    check(sslam_orb_search_for_initialization(G.get(), (const sslam_keypoint*)k1.data(), a.data(), (int)k1.size(), (const sslam_keypoint*)k2.data(),
                                              b.data(), (int)k2.size(), (float*)vbPrevMatched.data(), vnMatches12.data(), windowSize, nnratio,
                                              checkOri ? 1 : 0, bounds, &n));
    return n;
}
static_assert(sizeof(ProjQuery) == sizeof(sslam_proj_query), "ProjQuery layout");
// Device-resident copies of the frames the tracker keeps matching against: a Frame's mvKeysUn / mDescriptors /
// mvuRight never change after its constructor (src/Frame.cc:69-131), and Frame::mnId (include/Frame.h:128) names it -- copies made by
// `mLastFrame = Frame(mCurrentFrame)` share the id and the features.  The last few frames stay on the device, so the second and later
// matcher calls on a frame (Tracking.cc:1227 and its retry :1243, :1736) skip the upload of ~60 KB of features.
namespace {
// The key is (mnId, kind, n) AND a fingerprint of the features themselves: Tracking::Reset() sets Frame::nNextId back to 0
// (src/Tracking.cc:2150), so ids repeat after every reset -- a post-reset frame with the id and keypoint count of a cached pre-reset frame
// must not be matched against the old image's features.  InvalidateResidentFrames() drops everything (call it from Tracking::Reset; the
// fingerprint makes forgetting that harmless).
struct ResidentFrame { long id = -1; int kind = 0, n = 0; unsigned long long fp = 0; sslam_frame* h = nullptr; unsigned long stamp = 0; };
ResidentFrame g_resident[4];
unsigned long g_residentClock = 0;
std::mutex g_residentMu;
unsigned long long fnv1a(unsigned long long h, const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
// every keypoint, every descriptor row, mvuRight and the image bounds: 60 KB of hashing per lookup is cheap next to the upload it saves, and a frame that
// differs from the cached one ANYWHERE (stereo data, one row in the middle) must miss (round 4 hashed five sampled rows only)
unsigned long long frame_fingerprint(const std::vector<cv::KeyPoint>& k, const cv::Mat& desc, const float* uright, const float bounds[4]) {
    unsigned long long h = 1469598103934665603ull;
    const int n = (int)k.size();
    h = fnv1a(h, &n, sizeof(n));
    if (n) h = fnv1a(h, k.data(), sizeof(cv::KeyPoint) * (size_t)n);
    for (int i = 0; i < n && i < desc.rows; ++i) h = fnv1a(h, desc.ptr(i), 32);
    const int hasU = uright ? 1 : 0;
    h = fnv1a(h, &hasU, sizeof(hasU));
    if (uright && n) h = fnv1a(h, uright, sizeof(float) * (size_t)n);
    h = fnv1a(h, bounds, 4 * sizeof(float));
    return h;
}
// g_residentMu must be held by the caller, from the lookup to the end of the search that uses the handle (another thread may evict the slot)
sslam_frame* resident_frame_locked(long id, int kind, unsigned long long fp, const std::vector<cv::KeyPoint>& k, const cv::Mat& desc, const float* uright, const float bounds[4]) {
    const int n = (int)k.size();
    ResidentFrame* victim = &g_resident[0];
    for (auto& r : g_resident) {
        if (r.h && r.id == id && r.kind == kind && r.n == n && r.fp == fp) { r.stamp = ++g_residentClock; return r.h; }
        if (r.stamp < victim->stamp) victim = &r;
    }
    if (victim->h) { sslam_frame_destroy(victim->h); victim->h = nullptr; }
    const std::vector<uint8_t> a = rows32(desc);          // built on the miss path only
    sslam_frame* h = nullptr;
    check(sslam_frame_upload(G.get(), kind, k.data(), a.data(), n, uright, bounds, &h));
    victim->id = id; victim->kind = kind; victim->n = n; victim->fp = fp; victim->h = h; victim->stamp = ++g_residentClock;
    return h;
}
}  // namespace
void InvalidateResidentFrames() {
    std::lock_guard<std::mutex> lk(g_residentMu);
    for (auto& r : g_resident) { if (r.h) sslam_frame_destroy(r.h); r = ResidentFrame(); }
}
int SearchByProjection(int mode, const std::vector<cv::KeyPoint> &k, const cv::Mat &desc, const float bounds[4], const std::vector<float> *uRight,
                       const std::vector<unsigned char> &occupied, const std::vector<ProjQuery> &queries, const cv::Mat &qd, float nnratio,
                       int thDist, bool checkOri, std::vector<int> &assigned, long frameId) {
    assigned.assign(k.size(), -1);
    if (k.empty() || queries.empty()) return 0;
    std::vector<uint8_t> b = rows32(qd);
    int n = 0;
    if (frameId >= 0) {
        const unsigned long long fp = frame_fingerprint(k, desc, uRight ? uRight->data() : nullptr, bounds);
        std::lock_guard<std::mutex> lk(g_residentMu);      // lookup, upload and search under one lock: the handle cannot be evicted in between
        sslam_frame* fr = resident_frame_locked(frameId, 0, fp, k, desc, uRight ? uRight->data() : nullptr, bounds);
        check(sslam_search_by_projection_frame(G.get(), fr, mode, occupied.empty() ? nullptr : occupied.data(), (const sslam_proj_query*)queries.data(), b.data(),
                                               (int)queries.size(), nnratio, thDist, checkOri ? 1 : 0, assigned.data(), &n));
        return n;
    }
    std::vector<uint8_t> a = rows32(desc);
    check(sslam_search_by_projection(G.get(), 0, mode, k.data(), a.data(), (int)k.size(), bounds, uRight ? uRight->data() : nullptr,
                                     occupied.empty() ? nullptr : occupied.data(), (const sslam_proj_query*)queries.data(), b.data(),
                                     (int)queries.size(), nnratio, thDist, checkOri ? 1 : 0, assigned.data(), &n));
    return n;
}
int SearchLinesByProjection(const std::vector<cv::line_descriptor::KeyLine> &kl, const cv::Mat &ldesc, const std::vector<unsigned char> &occupied,
                            const std::vector<ProjQuery> &queries, const cv::Mat &qd, float nnratio, int thDist, std::vector<int> &assigned, int mode) {
    assigned.assign(kl.size(), -1);
    if (kl.empty() || queries.empty()) return 0;
    std::vector<uint8_t> a = rows32(ldesc), b = rows32(qd);
    const float bounds[4] = {0.f, 1.f, 0.f, 1.f};
    int n = 0;
    check(sslam_search_by_projection(G.get(), 1, mode, kl.data(), a.data(), (int)kl.size(), bounds, nullptr, occupied.empty() ? nullptr : occupied.data(),
                                     (const sslam_proj_query*)queries.data(), b.data(), (int)queries.size(), nnratio, thDist, 0, assigned.data(), &n));
    return n;
}
void KnnMatch2(const cv::Mat &q, const cv::Mat &t, std::vector<int> &idx, std::vector<int> &dist) {
    idx.assign((size_t)q.rows * 2, -1); dist.assign((size_t)q.rows * 2, -1);
    if (q.rows == 0) return;
    std::vector<uint8_t> a = rows32(q), b = rows32(t);
    check(sslam_hamming_knn2(G.get(), a.data(), q.rows, b.data(), t.rows, idx.data(), dist.data()));
}
int DistinctiveIndex(const std::vector<cv::Mat> &vDescriptors) {
    const int n = (int)vDescriptors.size();
    if (n == 0) return 0;                                   // the reference returns before reaching the loop
    std::vector<uint8_t> d((size_t)n * 32);
    for (int i = 0; i < n; ++i) memcpy(&d[(size_t)i * 32], vDescriptors[i].ptr(0), 32);
    const int32_t ptr[2] = {0, n};
    int32_t best = 0;
    check(sslam_distinctive_descriptors(G.get(), d.data(), ptr, 1, &best));
    return best;
}
ORBVocabulary::ORBVocabulary() : mHandle(nullptr) {}
ORBVocabulary::~ORBVocabulary() { if (mHandle) sslam_vocab_destroy((sslam_vocab*)mHandle); }
bool ORBVocabulary::loadFromTextFile(const std::string &filename) {
    sslam_vocab* v = nullptr;
    if (sslam_vocab_load_text(G.get(), filename.c_str(), &v) != SSLAM_OK) return false;      // the reference returns false on a bad file (TemplatedVocabulary.h:1359-1363)
    if (mHandle) sslam_vocab_destroy((sslam_vocab*)mHandle);
    mHandle = v;
    return true;
}
bool ORBVocabulary::empty() const { return mHandle == nullptr || size() == 0; }
unsigned int ORBVocabulary::size() const {
    int nwords = 0;
    if (mHandle) check(sslam_vocab_info((const sslam_vocab*)mHandle, nullptr, nullptr, nullptr, nullptr, nullptr, &nwords));
    return (unsigned int)nwords;
}
void ORBVocabulary::transform(const std::vector<cv::Mat> &features, BowVector &v, FeatureVector &fv, int levelsup) const {
    v.clear(); fv.clear();
    const int n = (int)features.size();
    if (empty() || n == 0) return;                          // TemplatedVocabulary.h:1131-1137
    std::vector<uint8_t> d((size_t)n * 32);
    for (int i = 0; i < n; ++i) memcpy(&d[(size_t)i * 32], features[i].ptr(0), 32);
    std::vector<int32_t> bw(n), fn(n), fp(n + 1), ff(n); std::vector<double> bv(n);
    int nb = 0, nf = 0;
    check(sslam_compute_bow(G.get(), (const sslam_vocab*)mHandle, d.data(), n, levelsup, bw.data(), bv.data(), &nb, fn.data(), fp.data(), ff.data(), &nf));
    for (int i = 0; i < nb; ++i) v.insert(v.end(), std::make_pair((unsigned int)bw[i], bv[i]));
    for (int j = 0; j < nf; ++j) fv.insert(fv.end(), std::make_pair((unsigned int)fn[j], std::vector<unsigned int>(ff.begin() + fp[j], ff.begin() + fp[j + 1])));
}
int LineMatch(const cv::Mat &l1, const cv::Mat &l2, double gateScale, bool ratioMode, std::vector<std::pair<int,int> > &matches) {
    matches.clear();
    if (l1.rows == 0 || l2.rows < 2) return 0;       // UB in the reference (src/LSDmatcher.cpp:167); defined as no matches
    std::vector<uint8_t> a = rows32(l1), b = rows32(l2);
    std::vector<int> pairs((size_t)l1.rows * 2);
    int n = 0;
    check(sslam_line_match(G.get(), a.data(), l1.rows, b.data(), l2.rows, gateScale, ratioMode ? 1 : 0, pairs.data(), l1.rows, &n, nullptr, nullptr));
    for (int i = 0; i < n; ++i) matches.push_back(std::make_pair(pairs[i * 2], pairs[i * 2 + 1]));
    return n;
}
}  // namespace sslam_shim
