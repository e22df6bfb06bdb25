// Exercises the drop-in C++ surface exactly as Frame.cc / Tracking.cc do and dumps the results for
// tests/test_shim_gpu.py to compare against the oracle.
//   shim_test <gray.raw> <w> <h> <prev.raw> <outprefix> <nfeatures> <maxlines>
#include <cstdio>
#include <fstream>
#include <iostream>
#include "ORBextractor.h"
#include "ExtractLineSegment.h"
#include "FrontendMatchers.h"
#include "ORBmatcher.h"
#include "LSDmatcher.h"

// Minimal stand-ins for the reference's Frame / KeyFrame / MapPoint / MapLine with exactly the members the matcher bodies read
// (include/Frame.h:150-200, include/MapPoint.h, include/MapLine.h, include/KeyFrame.h): the class drop-ins are member templates, so the
// reference's own types bind the same way at its call sites.
struct TMapPoint {
    bool mbTrackInView = true, bad = false; int mnTrackScaleLevel = 0, nObs = 1; float mTrackViewCos = 1.f, mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = -1;
    cv::Mat desc, worldPos;                    // worldPos: 3 x 1 CV_32F (MapPoint::GetWorldPos)
    cv::Mat GetWorldPos() const { return worldPos; }
    bool isBad() const { return bad; }
    cv::Mat GetDescriptor() const { return desc; }
    int Observations() const { return nObs; }
};
struct TVector6d { double v[6]; double operator()(int i) const { return v[i]; } };      // Eigen Vector6d (MapLine::GetWorldPos)
struct TMapLine {
    bool mbTrackInView = true, bad = false; int mnTrackScaleLevel = 0, nObs = 1; float mTrackViewCos = 1.f, mTrackProjX1 = 0, mTrackProjY1 = 0, mTrackProjX2 = 0, mTrackProjY2 = 0;
    cv::Mat desc; TVector6d worldPos{};
    TVector6d GetWorldPos() const { return worldPos; }
    bool isBad() const { return bad; }
    cv::Mat GetDescriptor() const { return desc; }
    int Observations() const { return nObs; }
};
struct TFrame {
    std::vector<cv::KeyPoint> mvKeysUn; cv::Mat mDescriptors; std::vector<float> mvuRight, mvScaleFactors; std::vector<TMapPoint*> mvpMapPoints;
    std::vector<cv::line_descriptor::KeyLine> mvKeylinesUn; cv::Mat mLdesc; std::vector<TMapLine*> mvpMapLines; int NL = 0;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
    // what the two per-frame tracking calls read on top of that (include/Frame.h:96-200)
    cv::Mat mTcw; float fx = 0, fy = 0, cx = 0, cy = 0, mb = 0, mbf = 0; int N = 0;
    long unsigned int mnId = 0;                  // Frame::mnId: makes the shim keep this frame's features on the device between calls
    std::vector<cv::KeyPoint> mvKeys; std::vector<bool> mvbOutlier, mvbLineOutlier;
};
struct TKeyFrame {
    cv::Mat mLineDescriptors; std::vector<TMapLine*> lines;
    std::vector<TMapLine*> GetMapLineMatches() const { return lines; }
    TMapLine* GetMapLine(size_t i) const { return lines[i]; }
};

static std::vector<uint8_t> readAll(const char* p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }
template <class T> static void dump(const std::string& p, const T* d, size_t n) { std::ofstream f(p, std::ios::binary); f.write((const char*)d, sizeof(T) * n); }

int main(int argc, char** argv) {
    if (argc < 8) return 2;
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
    std::vector<uint8_t> cur = readAll(argv[1]), prev = readAll(argv[4]);
    const std::string out = argv[5];
    cv::Mat imCur(h, w, CV_8UC1, cur.data()), imPrev(h, w, CV_8UC1, prev.data());
    StructureSLAM::ORBextractor* ext = new StructureSLAM::ORBextractor(std::atoi(argv[6]), 1.2f, 8, 20, 7);   // Tracking.cc:118
    StructureSLAM::LineSegment::SetMaxLines(std::atoi(argv[7]));
    StructureSLAM::LineSegment* seg = nullptr;            // deliberately NOT constructed: Frame::mpLineSegment is never initialised in the reference
    alignas(8) char fake[sizeof(StructureSLAM::LineSegment)];
    seg = reinterpret_cast<StructureSLAM::LineSegment*>(fake);

    std::vector<cv::KeyPoint> k1, k2; cv::Mat d1, d2;
    (*ext)(imPrev, cv::Mat(), k1, d1);                    // Frame::ExtractORB, src/Frame.cc:158
    (*ext)(imCur, cv::Mat(), k2, d2);
    std::vector<cv::line_descriptor::KeyLine> l1, l2; cv::Mat ld1, ld2; std::vector<sslam_shim::Vector3d> f1, f2;
    seg->ExtractLineSegment(imPrev, l1, ld1, f1);         // Frame::ExtractLSD, src/Frame.cc:152
    seg->ExtractLineSegment(imCur, l2, ld2, f2);
    std::vector<cv::Point2f> prevMatched(k1.size());
    for (size_t i = 0; i < k1.size(); ++i) prevMatched[i] = k1[i].pt;      // Tracking.cc:340-342
    std::vector<int> m12;
    const float bounds[4] = {0.f, (float)w, 0.f, (float)h};
    int nm = sslam_shim::SearchForInitialization(k1, d1, k2, d2, bounds, prevMatched, m12, 100, 0.9f, true);   // Tracking.cc:365-366
    std::vector<std::pair<int,int> > lm;
    int nlm = sslam_shim::LineMatch(ld1, ld2, 0.5, false, lm);             // Tracking.cc:367-368
    std::vector<int> kidx, kdist;
    sslam_shim::KnnMatch2(ld1, ld2, kidx, kdist);
    // ---- the matcher CLASS drop-ins (shim/ORBmatcher.h, shim/LSDmatcher.h) with the call syntax of Tracking.cc / LocalMapping.cc
    {
        TFrame F1, F2;
        F1.mvKeysUn = k1; F1.mDescriptors = d1; F2.mvKeysUn = k2; F2.mDescriptors = d2;
        F1.mnMaxX = F2.mnMaxX = (float)w; F1.mnMaxY = F2.mnMaxY = (float)h;
        F1.mLdesc = ld1; F2.mLdesc = ld2; F1.mvKeylinesUn = l1; F2.mvKeylinesUn = l2; F1.NL = (int)l1.size(); F2.NL = (int)l2.size();
        for (int i = 0; i < 8; ++i) F2.mvScaleFactors.push_back(ext->GetScaleFactors()[i]);
        std::vector<cv::Point2f> pm2(k1.size()); for (size_t i = 0; i < k1.size(); ++i) pm2[i] = k1[i].pt;
        std::vector<int> m12c;
        StructureSLAM::ORBmatcher matcher(0.9, true);                          // Tracking.cc:365
        const int nmc = matcher.SearchForInitialization(F1, F2, pm2, m12c, 100);
        std::vector<std::pair<int,int> > lmc;
        StructureSLAM::LSDmatcher lmatcher;                                     // Tracking.cc:367
        const int nlmc = lmatcher.SerachForInitialize(F1, F2, lmc);
        if (nmc != nm || m12c != m12 || nlmc != nlm || lmc != lm) { std::fprintf(stderr, "class drop-ins differ from the free functions\n"); return 5; }
        // Tracking::SearchLocalPoints (src/Tracking.cc:1729-1736): map points = the previous frame's keypoints "projected" to where they were
        std::vector<TMapPoint> mps(k1.size());
        for (size_t i = 0; i < k1.size(); ++i) {
            TMapPoint& m = mps[i];
            m.desc = d1.row((int)i); m.mTrackProjX = k1[i].pt.x + 2.5f; m.mTrackProjY = k1[i].pt.y - 1.5f; m.mnTrackScaleLevel = k1[i].octave;
            m.mTrackViewCos = (i % 2) ? 0.9995f : 0.9f; m.mbTrackInView = (i % 3) != 0; m.bad = (i % 7) == 0;
        }
        std::vector<TMapPoint*> vp; for (auto& m : mps) vp.push_back(&m);
        TMapPoint held; held.nObs = 1; TMapPoint ghost; ghost.nObs = 0;
        F2.mvpMapPoints.assign(k2.size(), nullptr); F2.mvuRight.assign(k2.size(), -1.f);
        for (size_t i = 0; i < k2.size(); ++i) if (i % 5 == 0) F2.mvpMapPoints[i] = (i % 10 == 0) ? &held : &ghost;      // observed points block a keypoint, unobserved ones do not
        StructureSLAM::ORBmatcher local(0.8);                                  // Tracking.cc:1729
        const int nloc = local.SearchByProjection(F2, vp, 3);
        std::vector<int> own(k2.size(), -1);
        for (size_t i = 0; i < k2.size(); ++i) { TMapPoint* q = F2.mvpMapPoints[i]; own[i] = q == nullptr ? -1 : q == &held ? -2 : q == &ghost ? -3 : (int)(q - mps.data()); }
        own.push_back(nloc);
        dump(out + "_cls_local.bin", own.data(), own.size());
        // LSDmatcher on keyframes: LocalMapping.cc:920 (triangulation, gate 0.1) and Tracking.cc:1024 (keyframe -> frame, ratio gate)
        std::vector<TMapLine> mls(l1.size() + l2.size());
        TKeyFrame KF1, KF2; KF1.mLineDescriptors = ld1; KF2.mLineDescriptors = ld2;
        for (size_t i = 0; i < l1.size(); ++i) KF1.lines.push_back(i % 4 == 0 ? &mls[i] : nullptr);
        for (size_t i = 0; i < l2.size(); ++i) KF2.lines.push_back(i % 6 == 1 ? &mls[l1.size() + i] : nullptr);
        std::vector<std::pair<size_t, size_t> > tri;
        lmatcher.SearchForTriangulation(&KF1, &KF2, tri);
        std::vector<int> trif; for (auto& pr : tri) { trif.push_back((int)pr.first); trif.push_back((int)pr.second); }
        dump(out + "_cls_tri.bin", trif.data(), trif.size());
        std::vector<TMapLine*> k2f;
        const int nk2f = lmatcher.SearchByProjection(&KF1, F2, k2f);
        std::vector<int> k2fi; for (TMapLine* q : k2f) k2fi.push_back(q ? (int)(q - mls.data()) : -1);
        k2fi.push_back(nk2f);
        dump(out + "_cls_kf2f.bin", k2fi.data(), k2fi.size());
        std::vector<TMapLine*> kk;
        lmatcher.SearchByDescriptor(&KF1, &KF2, kk);
        std::vector<int> kki; for (TMapLine* q : kk) kki.push_back(q ? (int)(q - mls.data()) : -1);
        dump(out + "_cls_kf2kf.bin", kki.data(), kki.size());
    }
    // ---- the per-frame tracking calls (Tracking::TrackWithMotionModel, src/Tracking.cc:1227-1243): ORBmatcher / LSDmatcher ::SearchByProjection(
    // CurrentFrame, LastFrame, th, bMono) through stand-in frames with poses, intrinsics and map points / lines at known world positions.
    // Poses, world positions and flags are dumped; tests/test_shim_gpu.py forms the same queries in float32 and asks the oracle.
    {
        TFrame Last, Cur;
        Last.mnId = 41; Cur.mnId = 42;          // pass 1 below finds Cur's features resident (uploaded by pass 0)
        Last.mvKeysUn = k1; Last.mvKeys = k1; Last.N = (int)k1.size(); Last.NL = (int)l1.size(); Last.mvKeylinesUn = l1;
        Cur.mvKeysUn = k2; Cur.mvKeys = k2; Cur.mDescriptors = d2; Cur.N = (int)k2.size(); Cur.mvKeylinesUn = l2; Cur.mLdesc = ld2; Cur.NL = (int)l2.size();
        Cur.mnMaxX = (float)w; Cur.mnMaxY = (float)h;
        for (int i = 0; i < 8; ++i) Cur.mvScaleFactors.push_back(ext->GetScaleFactors()[i]);
        Cur.fx = 520.9f; Cur.fy = 521.0f; Cur.cx = 325.1f; Cur.cy = 249.7f; Cur.mbf = 40.f;
        Last.mTcw = cv::Mat(4, 4, CV_32F); Cur.mTcw = cv::Mat(4, 4, CV_32F);
        const float cz = 0.99995f, sz = 0.0099998f;                     // ~0.573 degrees about the optical axis + a small translation
        const float TL[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        const float TC[16] = {cz, -sz, 0, 0.012f, sz, cz, 0, -0.007f, 0, 0, 1, -0.05f, 0, 0, 0, 1};
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { Last.mTcw.at<float>(r, c) = TL[4 * r + c]; Cur.mTcw.at<float>(r, c) = TC[4 * r + c]; }
        std::vector<TMapPoint> mps(k1.size()); std::vector<float> wp(3 * k1.size());
        Last.mvpMapPoints.assign(k1.size(), nullptr); Last.mvbOutlier.assign(k1.size(), false);
        for (size_t i = 0; i < k1.size(); ++i) {
            const float z = 2.0f + 0.35f * (float)(i % 7);
            wp[3 * i] = (k1[i].pt.x - Cur.cx) / Cur.fx * z; wp[3 * i + 1] = (k1[i].pt.y - Cur.cy) / Cur.fy * z; wp[3 * i + 2] = (i % 61 == 0) ? -z : z;      // a few behind the camera
            mps[i].worldPos = cv::Mat(3, 1, CV_32F, &wp[3 * i]); mps[i].desc = d1.row((int)i); mps[i].nObs = (i % 11 == 0) ? 0 : 2;
            if (i % 4 != 0) Last.mvpMapPoints[i] = &mps[i];
            Last.mvbOutlier[i] = (i % 9 == 0);
        }
        std::vector<TMapLine> mls(l1.size()); std::vector<double> wl(6 * l1.size());
        Last.mvpMapLines.assign(l1.size(), nullptr); Last.mvbLineOutlier.assign(l1.size(), false);
        for (size_t i = 0; i < l1.size(); ++i) {
            const double z = 2.5 + 0.4 * (double)(i % 5);
            const double e[4] = {l1[i].startPointX, l1[i].startPointY, l1[i].endPointX, l1[i].endPointY};
            for (int k = 0; k < 2; ++k) { mls[i].worldPos.v[3 * k] = (e[2 * k] - Cur.cx) / Cur.fx * z; mls[i].worldPos.v[3 * k + 1] = (e[2 * k + 1] - Cur.cy) / Cur.fy * z; mls[i].worldPos.v[3 * k + 2] = z; }
            for (int k = 0; k < 6; ++k) wl[6 * i + k] = mls[i].worldPos.v[k];
            mls[i].desc = ld1.row((int)i); mls[i].bad = (i % 13 == 5); mls[i].nObs = (i % 6 == 0) ? 0 : 1;
            if (i % 3 != 1) Last.mvpMapLines[i] = &mls[i];
            Last.mvbLineOutlier[i] = (i % 8 == 7);
        }
        TMapPoint held; held.nObs = 1; TMapPoint ghost; ghost.nObs = 0; TMapLine lheld; lheld.nObs = 1; TMapLine lghost; lghost.nObs = 0;
        std::vector<int> res;
        for (int pass = 0; pass < 2; ++pass) {          // pass 0: monocular (levels [oct-1, oct+1]); pass 1: "stereo", moving forward (levels >= oct)
            const bool bMono = pass == 0;
            Cur.mb = 0.01f;                              // tlc.z = 0.05 > mb: bForward when !bMono
            Cur.mvpMapPoints.assign(k2.size(), nullptr); Cur.mvuRight.assign(k2.size(), -1.f);
            for (size_t i = 0; i < k2.size(); ++i) if (i % 5 == 0) Cur.mvpMapPoints[i] = (i % 10 == 0) ? &held : &ghost;
            Cur.mvpMapLines.assign(l2.size(), nullptr);
            for (size_t i = 0; i < l2.size(); ++i) if (i % 4 == 0) Cur.mvpMapLines[i] = (i % 8 == 0) ? &lheld : &lghost;
            StructureSLAM::ORBmatcher matcher(0.9, true);                      // Tracking.cc:1216
            const int np = matcher.SearchByProjection(Cur, Last, 15.f, bMono); // Tracking.cc:1227 (th = 15 for monocular)
            StructureSLAM::LSDmatcher lmatcher;
            const int nl = lmatcher.SearchByProjection(Cur, Last, 15.f, bMono);
            for (size_t i = 0; i < k2.size(); ++i) { TMapPoint* q = Cur.mvpMapPoints[i]; res.push_back(q == nullptr ? -1 : q == &held ? -2 : q == &ghost ? -3 : (int)(q - mps.data())); }
            res.push_back(np);
            for (size_t i = 0; i < l2.size(); ++i) { TMapLine* q = Cur.mvpMapLines[i]; res.push_back(q == nullptr ? -1 : q == &lheld ? -2 : q == &lghost ? -3 : (int)(q - mls.data())); }
            res.push_back(nl);
        }
        dump(out + "_trk.bin", res.data(), res.size());
        dump(out + "_trk_wp.bin", wp.data(), wp.size());
        dump(out + "_trk_wl.bin", wl.data(), wl.size());
        std::vector<float> cam = {Cur.fx, Cur.fy, Cur.cx, Cur.cy, Cur.mbf, Cur.mb};
        cam.insert(cam.end(), TL, TL + 16); cam.insert(cam.end(), TC, TC + 16);
        dump(out + "_trk_cam.bin", cam.data(), cam.size());
    }
    // empty image: outputs untouched
    std::vector<cv::KeyPoint> ke(3); cv::Mat de; (*ext)(cv::Mat(), cv::Mat(), ke, de);
    const int emptyOk = ke.size() == 3 ? 1 : 0;

    dump(out + "_kp.bin", k2.data(), k2.size());
    std::vector<uint8_t> dd((size_t)d2.rows * 32); for (int i = 0; i < d2.rows; ++i) memcpy(&dd[(size_t)i * 32], d2.ptr(i), 32);
    dump(out + "_desc.bin", dd.data(), dd.size());
    dump(out + "_kl.bin", l2.data(), l2.size());
    std::vector<uint8_t> ll((size_t)ld2.rows * 32); for (int i = 0; i < ld2.rows; ++i) memcpy(&ll[(size_t)i * 32], ld2.ptr(i), 32);
    dump(out + "_ldesc.bin", ll.data(), ll.size());
    dump(out + "_fn.bin", f2.data(), f2.size());
    dump(out + "_m12.bin", m12.data(), m12.size());
    std::vector<int> lmf; for (auto& p : lm) { lmf.push_back(p.first); lmf.push_back(p.second); }
    dump(out + "_lm.bin", lmf.data(), lmf.size());
    // MapPoint::ComputeDistinctiveDescriptors over the first rows of the ORB descriptors (src/MapPoint.cc:276-306)
    std::vector<cv::Mat> obs; for (int i = 0; i < std::min(d2.rows, 37); ++i) obs.push_back(d2.row(i));
    const int bestIdx = sslam_shim::DistinctiveIndex(obs);
    // Frame::ComputeBoW (src/Frame.cc:474-481) with a vocabulary loaded the way System.cc:64-73 does
    int nWords = -1;
    if (argc > 8) {
        sslam_shim::ORBVocabulary voc;
        if (!voc.loadFromTextFile(argv[8])) return 3;
        nWords = (int)voc.size();
        std::vector<cv::Mat> vCurrentDesc; for (int j = 0; j < d2.rows; ++j) vCurrentDesc.push_back(d2.row(j));      // Converter::toDescriptorVector
        sslam_shim::BowVector bowVec; sslam_shim::FeatureVector featVec;
        voc.transform(vCurrentDesc, bowVec, featVec, 2);
        std::vector<double> bow; for (auto& kv : bowVec) { bow.push_back((double)kv.first); bow.push_back(kv.second); }
        std::vector<int> fv; for (auto& kv : featVec) { fv.push_back((int)kv.first); fv.push_back((int)kv.second.size()); for (unsigned f : kv.second) fv.push_back((int)f); }
        dump(out + "_bow.bin", bow.data(), bow.size());
        dump(out + "_fv.bin", fv.data(), fv.size());
        sslam_shim::ORBVocabulary bad;
        if (bad.loadFromTextFile(std::string(argv[8]) + ".missing") || !bad.empty()) return 4;
    }
    {   // Frame ids repeat after Tracking::Reset() (src/Tracking.cc:2150 sets Frame::nNextId = 0): a second frame with the id AND the keypoint count of
        // a frame whose features are resident must be matched against ITS features, not the cached ones (the cache keys on a fingerprint as well)
        const int n = (int)std::min(k1.size(), k2.size());
        if (n > 50) {
            std::vector<cv::KeyPoint> ka(k2.begin(), k2.begin() + n), kb(k1.begin(), k1.begin() + n);
            cv::Mat da(n, 32, CV_8U), db(n, 32, CV_8U);
            for (int i = 0; i < n; ++i) { memcpy(da.ptr(i), d2.ptr(i), 32); memcpy(db.ptr(i), d1.ptr(i), 32); }
            std::vector<sslam_shim::ProjQuery> q; cv::Mat qd(std::min(n, 200), 32, CV_8U);
            for (int i = 0; i < qd.rows; ++i) { sslam_shim::ProjQuery e = {ka[i].pt.x + 1.f, ka[i].pt.y - 1.f, 0.f, 0.f, 12.f, 0, 7, ka[i].angle, -1.f, 1, 1}; q.push_back(e); memcpy(qd.ptr(i), da.ptr(i), 32); }
            const float bounds[4] = {0.f, (float)w, 0.f, (float)h};
            std::vector<unsigned char> occ; std::vector<int> ra, rb, rb0, rb2;
            sslam_shim::SearchByProjection(0, ka, da, bounds, nullptr, occ, q, qd, 0.9f, 100, true, ra, 4711);       // uploads "frame 4711"
            sslam_shim::SearchByProjection(0, kb, db, bounds, nullptr, occ, q, qd, 0.9f, 100, true, rb, 4711);       // another image under the same id and count
            sslam_shim::SearchByProjection(0, kb, db, bounds, nullptr, occ, q, qd, 0.9f, 100, true, rb0, -1);        // the same call without the cache
            sslam_shim::InvalidateResidentFrames();
            sslam_shim::SearchByProjection(0, kb, db, bounds, nullptr, occ, q, qd, 0.9f, 100, true, rb2, 4711);
            if (rb != rb0 || rb2 != rb0 || ra == rb0) { std::fprintf(stderr, "shim_test: a reused frame id was served from the resident-frame cache\n"); return 5; }
        }
    }
    int meta[8] = {(int)k2.size(), (int)l2.size(), nm, nlm, emptyOk, ext->GetLevels(), bestIdx, nWords};
    dump(out + "_meta.bin", meta, 8);
    std::printf("shim_test: %zu keypoints, %zu lines, %d ORB matches, %d line matches\n", k2.size(), l2.size(), nm, nlm);
    delete ext;
    return 0;
}
