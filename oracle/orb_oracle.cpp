// ORACLE — TEST INFRASTRUCTURE ONLY (see cvleaf.h header).
//
// CPU restatement of StructureSLAM::ORBextractor (reference src/ORBextractor.cc,
// include/ORBextractor.h) with the OpenCV leaves from cvleaf.h.
// Each function cites the reference lines it follows.
//
// PINNED (round 4) for everything that is the reference's own code: `make -C oracle/ref_pin pin-stub`
// compiles /root/reference/src/ORBextractor.cc unmodified against a stub cv:: layer whose leaves are
// cvleaf.h, and its keypoints + descriptors equal this file's byte for byte on 16 fixtures (the reference's
// real frame, synthetic frames up to 1280x960, non-default scale / level / threshold settings):
// oracle/ref_pin/pin_report_stub.json, tests/test_pin_cpu.py.  The OpenCV leaves themselves stay UNPINNED.
//
// Determinism decision D1: the reference breaks ties in the
// careful-split ordering by heap pointer value (src/ORBextractor.cc:684); here a
// node's creation sequence number stands in for its address (later created ==
// larger), so equal-size nodes are split latest-created first.
#include "oracle.h"
#include "cvleaf.h"
#include "lines_types.h"
#include <list>
#include <utility>

namespace orc {

int g_gaussVariant = 0;      // decision D6 (0) or its OpenCV-3.4.0 alternative (1) for EVERY 8-bit GaussianBlur of the path (ORB 7x7 sigma 2, LSD 7x7 sigma 0.75, LBD 5x5 sigma 1): cvleaf.h, orc_set_gauss_variant

static const int kPattern[1024] = {
#include "orb_pattern.inc"
};

static const int PATCH_SIZE = 31, HALF_PATCH_SIZE = 15, EDGE_THRESHOLD = 19;   // :72-74

struct OrbParams {
    int nfeatures; float scaleFactorF; double scaleFactor; int nlevels, iniTh, minTh;
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> perLevel, umax;
};

// ORBextractor::ORBextractor, src/ORBextractor.cc:410-470.  NB the member
// scaleFactor is a double holding a float value (include/ORBextractor.h:97).
static OrbParams make_params(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
    OrbParams P;
    P.nfeatures = nfeatures; P.scaleFactorF = scaleFactor; P.scaleFactor = (double)scaleFactor;
    P.nlevels = nlevels; P.iniTh = iniTh; P.minTh = minTh;
    P.scale.resize(nlevels); P.sigma2.resize(nlevels);
    P.scale[0] = 1.0f; P.sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; ++i) {
        P.scale[i] = (float)(P.scale[i - 1] * P.scaleFactor);
        P.sigma2[i] = P.scale[i] * P.scale[i];
    }
    P.invScale.resize(nlevels); P.invSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; ++i) { P.invScale[i] = 1.0f / P.scale[i]; P.invSigma2[i] = 1.0f / P.sigma2[i]; }
    P.perLevel.resize(nlevels);
    float factor = (float)(1.0f / P.scaleFactor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) {
        P.perLevel[l] = cv_roundf(nDesired);
        sum += P.perLevel[l];
        nDesired *= factor;
    }
    P.perLevel[nlevels - 1] = std::max(nfeatures - sum, 0);
    // umax, :454-469
    P.umax.assign(HALF_PATCH_SIZE + 1, 0);
    int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) P.umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (P.umax[v0] == P.umax[v0 + 1]) ++v0;
        P.umax[v] = v0;
        ++v0;
    }
    return P;
}

// A padded pyramid level: `pad` is the whole buffer, the level proper is the ROI
// at (19,19) of size w x h  (ComputePyramid, :1107-1132).
struct Level { int w, h; Img8 pad; Img8 roi; };

static std::vector<Level> compute_pyramid(const OrbParams& P, const Img8& image) {
    std::vector<Level> L(P.nlevels);
    for (int l = 0; l < P.nlevels; ++l) {
        float scale = P.invScale[l];
        int w = cv_roundf((float)image.w * scale), h = cv_roundf((float)image.h * scale);
        L[l].w = w; L[l].h = h;
        if (l != 0) L[l].roi = resize_linear_8u(L[l - 1].roi, w, h);   // resize of the *previous level*, :1120
        else L[l].roi = image;
        L[l].pad = copy_make_border101(L[l].roi, EDGE_THRESHOLD);    // :1122,1127
    }
    return L;
}

struct KP { float x, y, size, angle, response; int octave, class_id; };

// ExtractorNode, include/ORBextractor.h:34-46; DivideNode src/ORBextractor.cc:481-537
struct Node {
    std::vector<KP> keys;
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    std::list<Node>::iterator lit;
    bool noMore = false;
    long seq = 0;                 // D1: creation order stands in for the heap address
};

static void divide_node(const Node& p, Node& n1, Node& n2, Node& n3, Node& n4) {
    const int halfX = (int)std::ceil((float)(p.URx - p.ULx) / 2);
    const int halfY = (int)std::ceil((float)(p.BRy - p.ULy) / 2);
    n1.ULx = p.ULx; n1.ULy = p.ULy; n1.URx = p.ULx + halfX; n1.URy = p.ULy;
    n1.BLx = p.ULx; n1.BLy = p.ULy + halfY; n1.BRx = p.ULx + halfX; n1.BRy = p.ULy + halfY;
    n2.ULx = n1.URx; n2.ULy = n1.URy; n2.URx = p.URx; n2.URy = p.URy;
    n2.BLx = n1.BRx; n2.BLy = n1.BRy; n2.BRx = p.URx; n2.BRy = p.ULy + halfY;
    n3.ULx = n1.BLx; n3.ULy = n1.BLy; n3.URx = n1.BRx; n3.URy = n1.BRy;
    n3.BLx = p.BLx; n3.BLy = p.BLy; n3.BRx = n1.BRx; n3.BRy = p.BLy;
    n4.ULx = n3.URx; n4.ULy = n3.URy; n4.URx = n2.BRx; n4.URy = n2.BRy;
    n4.BLx = n3.BRx; n4.BLy = n3.BRy; n4.BRx = p.BRx; n4.BRy = p.BRy;
    for (const KP& kp : p.keys) {
        if (kp.x < n1.URx) { if (kp.y < n1.BRy) n1.keys.push_back(kp); else n3.keys.push_back(kp); }
        else if (kp.y < n1.BRy) n2.keys.push_back(kp);
        else n4.keys.push_back(kp);
    }
    if (n1.keys.size() == 1) n1.noMore = true;
    if (n2.keys.size() == 1) n2.noMore = true;
    if (n3.keys.size() == 1) n3.noMore = true;
    if (n4.keys.size() == 1) n4.noMore = true;
}

// DistributeOctTree, src/ORBextractor.cc:539-763
static std::vector<KP> distribute_octtree(const std::vector<KP>& in, int minX, int maxX, int minY, int maxY, int N) {
    std::vector<KP> result;
    const int nIni = (int)std::round((float)(maxX - minX) / (maxY - minY));
    if (nIni <= 0) return result;           // reference: division by zero (portrait aspect < 0.5); defined here as "no keypoints"
    const float hX = (float)(maxX - minX) / nIni;
    std::list<Node> nodes;
    std::vector<Node*> ini(nIni);
    long seq = 0;
    for (int i = 0; i < nIni; ++i) {
        Node ni;
        ni.ULx = (int)(hX * (float)i); ni.ULy = 0;
        ni.URx = (int)(hX * (float)(i + 1)); ni.URy = 0;
        ni.BLx = ni.ULx; ni.BLy = maxY - minY;
        ni.BRx = ni.URx; ni.BRy = maxY - minY;
        ni.seq = seq++;
        nodes.push_back(ni);
        ini[i] = &nodes.back();
    }
    for (const KP& kp : in) {
        int r = (int)(kp.x / hX);
        if (r >= nIni) r = nIni - 1;        // cannot happen for x < width; guard only
        ini[r]->keys.push_back(kp);
    }
    for (auto it = nodes.begin(); it != nodes.end();) {
        if (it->keys.size() == 1) { it->noMore = true; ++it; }
        else if (it->keys.empty()) it = nodes.erase(it);
        else ++it;
    }
    bool finish = false;
    typedef std::pair<int, Node*> SP;
    std::vector<SP> sizeAndNode;
    auto push_child = [&](Node& c, int* nToExpand) {
        if (c.keys.empty()) return;
        c.seq = seq++;
        nodes.push_front(c);
        if (c.keys.size() > 1) {
            if (nToExpand) ++*nToExpand;
            sizeAndNode.push_back(SP((int)c.keys.size(), &nodes.front()));
            nodes.front().lit = nodes.begin();
        }
    };
    auto by_size_then_seq = [](const SP& a, const SP& b) {
        if (a.first != b.first) return a.first < b.first;
        return a.second->seq < b.second->seq;            // D1
    };
    while (!finish) {
        int prevSize = (int)nodes.size();
        auto it = nodes.begin();
        int nToExpand = 0;
        sizeAndNode.clear();
        while (it != nodes.end()) {
            if (it->noMore) { ++it; continue; }
            Node n1, n2, n3, n4;
            divide_node(*it, n1, n2, n3, n4);
            push_child(n1, &nToExpand); push_child(n2, &nToExpand);
            push_child(n3, &nToExpand); push_child(n4, &nToExpand);
            it = nodes.erase(it);
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
        else if ((int)nodes.size() + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = (int)nodes.size();
                std::vector<SP> prev = sizeAndNode;
                sizeAndNode.clear();
                std::sort(prev.begin(), prev.end(), by_size_then_seq);
                for (int j = (int)prev.size() - 1; j >= 0; --j) {
                    Node n1, n2, n3, n4;
                    divide_node(*prev[j].second, n1, n2, n3, n4);
                    push_child(n1, nullptr); push_child(n2, nullptr);
                    push_child(n3, nullptr); push_child(n4, nullptr);
                    nodes.erase(prev[j].second->lit);
                    if ((int)nodes.size() >= N) break;
                }
                if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
            }
        }
    }
    for (Node& n : nodes) {                               // :742-760 best response, first wins ties
        const KP* best = &n.keys[0];
        float mx = best->response;
        for (size_t k = 1; k < n.keys.size(); ++k)
            if (n.keys[k].response > mx) { best = &n.keys[k]; mx = n.keys[k].response; }
        result.push_back(*best);
    }
    return result;
}

// IC_Angle, src/ORBextractor.cc:77-104 (on the *unblurred* padded level)
static float ic_angle(const Img8& pad, float ptx, float pty, const std::vector<int>& umax) {
    int m_01 = 0, m_10 = 0;
    const int step = pad.w;
    const uint8_t* center = pad.d.data() + (size_t)(cv_roundf(pty) + EDGE_THRESHOLD) * step + (cv_roundf(ptx) + EDGE_THRESHOLD);
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int vp = center[u + v * step], vm = center[u - v * step];
            v_sum += (vp - vm);
            m_10 += u * (vp + vm);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10);
}

// computeOrbDescriptor, src/ORBextractor.cc:107-147 (on the blurred level, size w x h)
static void orb_descriptor(const KP& kp, const Img8& img, uint8_t* desc) {
    const float factorPI = (float)(M_PI / 180.f);
    float angle = kp.angle * factorPI;
    float a = cr_cosf(angle), b = cr_sinf(angle);      // D5
    const int step = img.w;
    const uint8_t* center = img.d.data() + (size_t)cv_roundf(kp.y) * step + cv_roundf(kp.x);
    const int* pat = kPattern;
    auto tap = [&](int idx) -> int {
        float px = (float)pat[idx * 2], py = (float)pat[idx * 2 + 1];
        int yy = cv_roundf(px * b + py * a), xx = cv_roundf(px * a - py * b);   // D4: no FMA
        return center[yy * step + xx];
    };
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int k = 0; k < 8; ++k) {
            int t0 = tap(2 * k), t1 = tap(2 * k + 1);
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

struct OrbDebug {      // optional per-stage taps for stage-by-stage parity tests
    std::vector<Img8> levels;                      // unpadded pyramid levels
    std::vector<std::vector<FastKp>> candidates;   // per level, coords relative to minBorder (16)
};

// ORBextractor::operator() :1043-1105 with ComputeKeyPointsOctTree :765-853
static void orb_extract(const OrbParams& P, const Img8& image, std::vector<KP>& kps,
                        std::vector<uint8_t>& desc, OrbDebug* dbg) {
    kps.clear(); desc.clear();
    if (image.w == 0 || image.h == 0) return;
    std::vector<Level> pyr = compute_pyramid(P, image);
    std::vector<std::vector<KP>> all(P.nlevels);
    if (dbg) { dbg->levels.clear(); dbg->candidates.assign(P.nlevels, {}); for (auto& l : pyr) dbg->levels.push_back(l.roi); }
    const float W = 30;
    for (int level = 0; level < P.nlevels; ++level) {
        const Img8& im = pyr[level].roi;     // mvImagePyramid[level] is the ROI; FAST views index it directly
        const int minBX = EDGE_THRESHOLD - 3, minBY = minBX;
        const int maxBX = im.w - EDGE_THRESHOLD + 3, maxBY = im.h - EDGE_THRESHOLD + 3;
        std::vector<KP> toDistribute;
        const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        if (nCols > 0 && nRows > 0) {
            const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
            std::vector<FastKp> cell;
            for (int i = 0; i < nRows; ++i) {
                const float iniY = (float)(minBY + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBY - 3) continue;
                if (maxY > maxBY) maxY = (float)maxBY;
                for (int j = 0; j < nCols; ++j) {
                    const float iniX = (float)(minBX + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBX - 6) continue;
                    if (maxX > maxBX) maxX = (float)maxBX;
                    fast9_view(im, (int)iniX, (int)iniY, (int)maxX, (int)maxY, P.iniTh, cell);
                    if (cell.empty()) fast9_view(im, (int)iniX, (int)iniY, (int)maxX, (int)maxY, P.minTh, cell);
                    for (const FastKp& c : cell) {
                        KP k; k.x = (float)c.x + j * wCell; k.y = (float)c.y + i * hCell;
                        k.size = 7.f; k.angle = -1.f; k.response = (float)c.score; k.octave = 0; k.class_id = -1;
                        toDistribute.push_back(k);
                        if (dbg) dbg->candidates[level].push_back({(int)k.x, (int)k.y, c.score});
                    }
                }
            }
        }
        std::vector<KP>& keypoints = all[level];
        keypoints = distribute_octtree(toDistribute, minBX, maxBX, minBY, maxBY, P.perLevel[level]);
        const int scaledPatchSize = (int)(PATCH_SIZE * P.scale[level]);
        for (KP& k : keypoints) { k.x += minBX; k.y += minBY; k.octave = level; k.size = (float)scaledPatchSize; }
    }
    for (int level = 0; level < P.nlevels; ++level)
        for (KP& k : all[level]) k.angle = ic_angle(pyr[level].pad, k.x, k.y, P.umax);
    int n = 0;
    for (auto& v : all) n += (int)v.size();
    desc.assign((size_t)n * 32, 0);
    int offset = 0;
    for (int level = 0; level < P.nlevels; ++level) {
        std::vector<KP>& keypoints = all[level];
        if (keypoints.empty()) continue;
        Img8 working = gaussian_blur_8u(pyr[level].roi, 7, 2.0, g_gaussVariant);     // clone + GaussianBlur(7x7, 2, 2, REFLECT_101), :1085-1086
        for (size_t i = 0; i < keypoints.size(); ++i) orb_descriptor(keypoints[i], working, &desc[(size_t)(offset + i) * 32]);
        offset += (int)keypoints.size();
        if (level != 0) {
            float scale = P.scale[level];
            for (KP& k : keypoints) { k.x *= scale; k.y *= scale; }
        }
        kps.insert(kps.end(), keypoints.begin(), keypoints.end());
    }
}

}  // namespace orc

// ---------------------------------------------------------------- C API (ctypes)
using namespace orc;

extern "C" {

int orc_orb_params(int nfeatures, float scaleFactor, int nlevels, float* scale_out, int* per_level_out, int* umax_out) {
    OrbParams P = make_params(nfeatures, scaleFactor, nlevels, 20, 7);
    for (int i = 0; i < nlevels; ++i) { scale_out[i] = P.scale[i]; per_level_out[i] = P.perLevel[i]; }
    for (int i = 0; i < 16; ++i) umax_out[i] = P.umax[i];
    return 0;
}

// keypoints as 28-byte records {x,y,size,angle,response (f32), octave, class_id (i32)}
int orc_orb_extract(const uint8_t* gray, int w, int h, int stride, int nfeatures, float scaleFactor, int nlevels,
                    int iniTh, int minTh, void* kp_out, uint8_t* desc_out, int cap) {
    OrbParams P = make_params(nfeatures, scaleFactor, nlevels, iniTh, minTh);
    Img8 im(w, h);
    for (int y = 0; y < h; ++y) std::memcpy(im.row(y), gray + (size_t)y * stride, w);
    std::vector<KP> kps; std::vector<uint8_t> desc;
    orb_extract(P, im, kps, desc, nullptr);
    int n = std::min((int)kps.size(), cap);
    if (n > 0) {      // (memcpy from an empty vector's null data() is undefined even for 0 bytes: tests/test_sanitize_cpu.py)
        std::memcpy(kp_out, kps.data(), (size_t)n * sizeof(KP));
        std::memcpy(desc_out, desc.data(), (size_t)n * 32);
    }
    return (int)kps.size();
}

// stage taps: pyramid level bytes (unpadded, contiguous) and FAST candidates per level
int orc_orb_pyramid_level(const uint8_t* gray, int w, int h, int stride, float scaleFactor, int nlevels, int level,
                          uint8_t* out, int* lw, int* lh) {
    OrbParams P = make_params(1000, scaleFactor, nlevels, 20, 7);
    Img8 im(w, h);
    for (int y = 0; y < h; ++y) std::memcpy(im.row(y), gray + (size_t)y * stride, w);
    std::vector<Level> pyr = compute_pyramid(P, im);
    *lw = pyr[level].w; *lh = pyr[level].h;
    if (out) std::memcpy(out, pyr[level].roi.d.data(), pyr[level].roi.d.size());
    return 0;
}

int orc_orb_candidates(const uint8_t* gray, int w, int h, int stride, int nfeatures, float scaleFactor, int nlevels,
                       int iniTh, int minTh, int level, int32_t* xys_out, int cap) {
    OrbParams P = make_params(nfeatures, scaleFactor, nlevels, iniTh, minTh);
    Img8 im(w, h);
    for (int y = 0; y < h; ++y) std::memcpy(im.row(y), gray + (size_t)y * stride, w);
    std::vector<KP> kps; std::vector<uint8_t> desc; OrbDebug dbg;
    orb_extract(P, im, kps, desc, &dbg);
    const auto& c = dbg.candidates[level];
    int n = std::min((int)c.size(), cap);
    for (int i = 0; i < n; ++i) { xys_out[i * 3] = c[i].x; xys_out[i * 3 + 1] = c[i].y; xys_out[i * 3 + 2] = c[i].score; }
    return (int)c.size();
}

int orc_set_gauss_variant(int v) { const int old = g_gaussVariant; g_gaussVariant = v == 1 ? 1 : 0; return old; }
int orc_blur7(const uint8_t* src, int w, int h, uint8_t* dst) {
    Img8 im(w, h); std::memcpy(im.d.data(), src, (size_t)w * h);
    Img8 o = gaussian_blur_8u(im, 7, 2.0, g_gaussVariant);
    std::memcpy(dst, o.d.data(), (size_t)w * h);
    return 0;
}

int orc_gauss_taps(int n, double sigma, int* out) {
    std::vector<int> t = g_gaussVariant == 1 ? gauss_taps_340(n, sigma) : gauss_taps_q8(n, sigma);
    for (int i = 0; i < n; ++i) out[i] = t[i];
    return 0;
}

float orc_fast_atan2(float y, float x) { return fast_atan2(y, x); }
int orc_reflect101(int i, int n) { return reflect101(i, n); }

// cv::FAST on the whole image as one view: score_map_out[w*h] = cornerScore of every pixel that passes the 9-of-16 segment test at
// `threshold` (0 elsewhere), before non-maximum suppression; returns the number of keypoints after NMS (xys_out: x, y, score).
int orc_fast_image(const uint8_t* gray, int w, int h, int stride, int threshold, uint8_t* score_map_out, int32_t* xys_out, int cap) {
    Img8 im(w, h);
    for (int y = 0; y < h; ++y) std::memcpy(im.row(y), gray + (size_t)y * stride, w);
    std::vector<FastKp> kps; std::vector<uint8_t> sc;
    fast9_view(im, 0, 0, w, h, threshold, kps, &sc);
    if (score_map_out) { if (sc.empty()) std::memset(score_map_out, 0, (size_t)w * h); else std::memcpy(score_map_out, sc.data(), (size_t)w * h); }
    const int n = std::min((int)kps.size(), cap);
    for (int i = 0; i < n && xys_out; ++i) { xys_out[i * 3] = kps[i].x; xys_out[i * 3 + 1] = kps[i].y; xys_out[i * 3 + 2] = kps[i].score; }
    return (int)kps.size();
}

int orc_fast_score(const uint8_t* patch7x7) {   // score of the centre of a 7x7 patch (threshold-independent form): cornerScore with t=0
    return fast_corner_score16(patch7x7 + 3 * 7 + 3, 7, 0);
}

}  // extern "C"
