// Driver of the pinning recipe (oracle/ref_pin/Makefile): calls the REFERENCE's ORBextractor and LineSegment::ExtractLineSegment -- compiled
// from /root/reference/src unmodified -- the way Frame::ExtractORB / ExtractLSD do (src/Frame.cc:150-161) and dumps their outputs in the
// C ABI's layouts (28-byte keypoints, 32-byte descriptor rows, 68-byte keylines, 3 doubles per line) for compare.py.
//   ref_dump <image.pgm> <outprefix>         (binary P5, 8 bit; parameters from the file name: NAME_nfeatures.pgm)
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>
#include "ORBextractor.h"
#include "ExtractLineSegment.h"

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
    if (argc < 3) return 2;
    cv::Mat img;
    if (!readPgm(argv[1], img)) { std::fprintf(stderr, "ref_dump: cannot read %s\n", argv[1]); return 3; }
    std::string stem = argv[1]; stem = stem.substr(0, stem.size() - 4);
    const int nfeat = std::atoi(stem.substr(stem.rfind('_') + 1).c_str());
    const std::string out = argv[2];
    StructureSLAM::ORBextractor ext(nfeat, 1.2f, 8, 20, 7);                 // Examples/ICL.yaml:41-54, src/Tracking.cc:118-120
    std::vector<cv::KeyPoint> kps; cv::Mat desc;
    ext(img, cv::Mat(), kps, desc);
    struct Kp { float x, y, size, angle, response; int octave, class_id; };
    std::vector<Kp> k(kps.size());
    for (size_t i = 0; i < kps.size(); ++i) k[i] = {kps[i].pt.x, kps[i].pt.y, kps[i].size, kps[i].angle, kps[i].response, kps[i].octave, kps[i].class_id};
    dump(out + "_kp.bin", k.data(), k.size());
    std::vector<unsigned char> d((size_t)desc.rows * 32);
    for (int i = 0; i < desc.rows; ++i) memcpy(&d[(size_t)i * 32], desc.ptr(i), 32);
    dump(out + "_desc.bin", d.data(), d.size());
    StructureSLAM::LineSegment seg;
    std::vector<cv::line_descriptor::KeyLine> kl; cv::Mat ld; std::vector<Eigen::Vector3d> fn;
    seg.ExtractLineSegment(img, kl, ld, fn);                               // hard cap of 40 lines, src/ExtractLineSegment.cpp:42
    struct Kl { float angle; int class_id, octave; float ptx, pty, response, size, sx, sy, ex, ey, sox, soy, eox, eoy, len; int npx; };
    std::vector<Kl> l(kl.size());
    for (size_t i = 0; i < kl.size(); ++i) {
        const cv::line_descriptor::KeyLine& q = kl[i];
        l[i] = {q.angle, q.class_id, q.octave, q.pt.x, q.pt.y, q.response, q.size, q.startPointX, q.startPointY, q.endPointX, q.endPointY,
                q.sPointInOctaveX, q.sPointInOctaveY, q.ePointInOctaveX, q.ePointInOctaveY, q.lineLength, q.numOfPixels};
    }
    dump(out + "_kl.bin", l.data(), l.size());
    std::vector<unsigned char> ldd((size_t)ld.rows * 32);
    for (int i = 0; i < ld.rows; ++i) memcpy(&ldd[(size_t)i * 32], ld.ptr(i), 32);
    dump(out + "_ldesc.bin", ldd.data(), ldd.size());
    std::vector<double> f3; for (auto& v : fn) { f3.push_back(v(0)); f3.push_back(v(1)); f3.push_back(v(2)); }
    dump(out + "_linefn.bin", f3.data(), f3.size());
    std::printf("ref_dump %s: %zu keypoints, %zu lines\n", argv[1], kps.size(), kl.size());
    return 0;
}
