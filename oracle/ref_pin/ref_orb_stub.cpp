// ORACLE — TEST INFRASTRUCTURE ONLY.  Driver of `make -C oracle/ref_pin pin-stub`: calls the REFERENCE's StructureSLAM::ORBextractor --
// /root/reference/src/ORBextractor.cc compiled unmodified against oracle/ref_pin/stub_cv -- the way Frame::ExtractORB does
// (src/Frame.cc:155-161) and dumps keypoints (28 B) and descriptor rows (32 B) for compare_stub.py.
//   ref_orb_stub <image.pgm> <outprefix> <nfeatures> [scaleFactor=1.2 nlevels=8 iniThFAST=20 minThFAST=7]
// -DPIN_CR_TRIG: see below (decision D5).  -DPIN_BUMP_ALLOC replaces the global allocator with a monotonic arena, so that the heap addresses of the quadtree's list nodes
// increase with creation order: src/ORBextractor.cc:684 sorts (size, ExtractorNode*) pairs, and decision D1 of the oracle
// ("equal sizes: later-created node first") is exactly what that sort yields under such an allocator.  Without the flag the
// nodes come from glibc malloc (recycled addresses) -- the difference between the two builds is D1's error bar.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <string>
#include <vector>
#include "ORBextractor.h"

#ifdef PIN_CR_TRIG
// Decision D5 of the oracle: cosf / sinf := the correctly rounded float of the real function.  src/ORBextractor.cc:113 calls cos(float) / sin(float), i.e. libm's cosf / sinf,
// which are within an ulp but not correctly rounded (glibc here: 1.3 % of random angles differ in the last bit, and which of its ifunc variants runs depends on the CPU).
// Defining the two symbols in the executable makes the reference's own code take the correctly rounded values; the build without this flag (ref_orb_stub_libm) is D5's error bar.
#include <cmath>
extern "C" float cosf(float x) noexcept { return (float)cos((double)x); }
extern "C" float sinf(float x) noexcept { return (float)sin((double)x); }
extern "C" void sincosf(float x, float* s, float* c) noexcept { *s = (float)sin((double)x); *c = (float)cos((double)x); }      // (GCC merges the pair of calls at src/ORBextractor.cc:113 into this one)
#endif
#ifdef PIN_BUMP_ALLOC
#include <sys/mman.h>
static char* g_arena = nullptr; static size_t g_off = 0; static const size_t kArena = (size_t)24 << 30;   // lazily committed
static void* bump(size_t n) {
    if (!g_arena) { g_arena = (char*)mmap(nullptr, kArena, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0); if (g_arena == MAP_FAILED) abort(); }
    size_t o = (g_off + 15) & ~(size_t)15; if (o + n > kArena) abort(); g_off = o + n; return g_arena + o;
}
void* operator new(size_t n) { return bump(n); }
void* operator new[](size_t n) { return bump(n); }
void operator delete(void*) noexcept {}
void operator delete[](void*) noexcept {}
void operator delete(void*, size_t) noexcept {}
void operator delete[](void*, size_t) noexcept {}
#endif

static bool readPgm(const char* path, cv::Mat& img) {
    std::ifstream f(path, std::ios::binary);
    std::string magic; int w = 0, h = 0, mx = 0;
    f >> magic >> w >> h >> mx; f.get();
    if (magic != "P5" || mx != 255 || w <= 0 || h <= 0) return false;
    img.create(h, w, CV_8UC1);
    f.read((char*)img.data, (std::streamsize)w * h);
    return (bool)f;
}
template <class T> static void dump(const std::string& p, const T* d, size_t n) { std::ofstream f(p, std::ios::binary); f.write((const char*)d, sizeof(T) * n); }

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    cv::Mat img;
    if (!readPgm(argv[1], img)) { std::fprintf(stderr, "ref_orb_stub: cannot read %s\n", argv[1]); return 3; }
    const std::string out = argv[2];
    const int nfeat = std::atoi(argv[3]);
    const float sf = argc > 4 ? (float)std::atof(argv[4]) : 1.2f;
    const int nlev = argc > 5 ? std::atoi(argv[5]) : 8, ini = argc > 6 ? std::atoi(argv[6]) : 20, mn = argc > 7 ? std::atoi(argv[7]) : 7;
    StructureSLAM::ORBextractor ext(nfeat, sf, nlev, ini, mn);              // Examples/ICL.yaml:41-54, src/Tracking.cc:118-120
    std::vector<cv::KeyPoint> kps; cv::Mat desc;
    ext(img, cv::Mat(), kps, desc);
    dump(out + "_kp.bin", kps.data(), kps.size());
    std::vector<unsigned char> d((size_t)desc.rows * 32);
    for (int i = 0; i < desc.rows; ++i) std::memcpy(&d[(size_t)i * 32], desc.ptr(i), 32);
    dump(out + "_desc.bin", d.data(), d.size());
    // the constructor's tables (a1): scale factors, sigma^2, and their inverses through the public getters
    std::vector<float> t;
    for (float v : ext.GetScaleFactors()) t.push_back(v);
    for (float v : ext.GetInverseScaleFactors()) t.push_back(v);
    for (float v : ext.GetScaleSigmaSquares()) t.push_back(v);
    for (float v : ext.GetInverseScaleSigmaSquares()) t.push_back(v);
    dump(out + "_tables.bin", t.data(), t.size());
    std::printf("ref_orb_stub %s nfeatures=%d: %zu keypoints\n", argv[1], nfeat, kps.size());
    return 0;
}
