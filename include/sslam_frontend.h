/* sslam_frontend.h — C ABI of the MI355X-native point+line feature front-end.
 *
 * This is the drop-in boundary beneath the reference's C++ call surface
 * (Structure-SLAM-PointLine).  Every entry point names the reference interface it
 * replaces (paths relative to the reference tree).  Plain pointers and sizes only;
 * no C++/torch/HIP types in any signature (streams travel as void*).
 *
 * Conventions
 *   - every function returns an int status: 0 = ok, <0 = error (sslam_status_str()).
 *     Nothing throws across the boundary.
 *   - "host" entry points take host pointers and are synchronous (H2D, kernels, D2H);
 *     "_dev" entry points take DEVICE pointers, enqueue on the given HIP stream
 *     (void* = hipStream_t, NULL = the context's own stream) and return without
 *     synchronising.
 *   - The caller owns every output buffer (reference: _keypoints/_descriptors are
 *     caller containers, src/ORBextractor.cc:1064-1073).
 *   - The library fails loudly (SSLAM_ERR_NO_DEVICE) when no HIP device is usable:
 *     there is no CPU fallback.
 */
#ifndef SSLAM_FRONTEND_H
#define SSLAM_FRONTEND_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSLAM_OK 0
#define SSLAM_ERR_INVALID (-1)     /* bad argument */
#define SSLAM_ERR_NO_DEVICE (-2)   /* no usable HIP device / HIP runtime error */
#define SSLAM_ERR_CAPACITY (-3)    /* caller buffer too small */
#define SSLAM_ERR_HIP (-4)         /* a HIP call failed; see sslam_last_error() */
#define SSLAM_ERR_UNSUPPORTED (-5)

/* cv::KeyPoint, 28 bytes (opencv2/core/types.hpp): pt.x, pt.y, size, angle,
 * response, octave, class_id.  Layout consumed verbatim by include/Frame.h:155-160. */
typedef struct sslam_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} sslam_keypoint;

/* cv::line_descriptor::KeyLine, 68 bytes (opencv2/line_descriptor/descriptor.hpp):
 * consumed by include/Frame.h:183-189. */
typedef struct sslam_keyline {
    float angle;
    int32_t class_id, octave;
    float pt_x, pt_y, response, size;
    float startPointX, startPointY, endPointX, endPointY;
    float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
    float lineLength;
    int32_t numOfPixels;
} sslam_keyline;

typedef struct sslam_ctx sslam_ctx;     /* device context: HIP device, stream, scratch */
typedef struct sslam_orb sslam_orb;     /* = StructureSLAM::ORBextractor state */
typedef struct sslam_lines sslam_lines; /* = LSDDetector + BinaryDescriptor state */

const char* sslam_status_str(int status);
const char* sslam_last_error(void);      /* thread-local text of the last failure */
int sslam_abi_version(void);

/* ---- context ------------------------------------------------------------- */
int sslam_ctx_create(int device, sslam_ctx** out);
int sslam_ctx_destroy(sslam_ctx* ctx);
int sslam_ctx_synchronize(sslam_ctx* ctx);
void* sslam_ctx_stream(sslam_ctx* ctx);  /* hipStream_t of the context */

/* Per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline
 * leg; no reference counterpart).  drain() synchronises, returns the number of distinct
 * kernels and fills name / total ms / launch count per kernel. */
int sslam_profile_enable(sslam_ctx* ctx, int on);
int sslam_profile_drain(sslam_ctx* ctx, const char** names_out, double* ms_out, int* launches_out, int cap);

/* ---- ORB  (replaces StructureSLAM::ORBextractor) --------------------------- */
/* ORBextractor::ORBextractor(int nfeatures,float scaleFactor,int nlevels,int iniThFAST,
 * int minThFAST), src/ORBextractor.cc:410-470. */
int sslam_orb_create(sslam_ctx* ctx, int nfeatures, float scaleFactor, int nlevels,
                     int iniThFAST, int minThFAST, sslam_orb** out);
int sslam_orb_destroy(sslam_orb* orb);
/* Which 8-bit cv::GaussianBlur(7x7, sigma 2) the descriptors are computed on (src/ORBextractor.cc:1085-1086; the leaf is OpenCV's, and its
 * arithmetic changed inside the 3.4 series): 0 (default) = the bit-exact fixed-point filter of OpenCV >= 3.4.1, 8.8 taps 18 34 48 56 48 34 18
 * that sum to 256 (DESIGN.md decision D6); 1 = OpenCV 3.4.0 -- the version the reference's README names -- whose filter engine rounds every
 * tap of the float kernel: 18 34 49 55 49 34 18 (sum 257, result saturated).  Keypoints are the same either way; 3.7 % of the descriptor bits
 * differ between the two (oracle/ref_pin/pin_report_stub.json).  Takes effect with the next extraction. */
int sslam_orb_set_blur_variant(sslam_orb* orb, int variant);

/* GetLevels/GetScaleFactors/GetInverseScaleFactors/GetScaleSigmaSquares/
 * GetInverseScaleSigmaSquares, include/ORBextractor.h:63-83.  Each array has
 * nlevels entries; NULL pointers are skipped. */
int sslam_orb_get_scales(const sslam_orb* orb, float* scale, float* inv_scale,
                         float* sigma2, float* inv_sigma2, int32_t* features_per_level);

/* Upper bound on keypoints one frame can return (nfeatures + per-level overshoot,
 * SURVEY.md D.2 step 5); size kp/desc buffers with it. */
int sslam_orb_max_keypoints(const sslam_orb* orb);

/* ORBextractor::operator()(image, mask, keypoints, descriptors),
 * src/ORBextractor.cc:1043-1105 as called from Frame::ExtractORB (src/Frame.cc:155-161).
 * gray: host CV_8UC1, stride in bytes.  kp_out[cap], desc_out[cap*32] host.
 * Empty image (w==0||h==0) -> *n_out = 0, outputs untouched (:1046-1047). */
int sslam_orb_extract(sslam_orb* orb, const uint8_t* gray, int w, int h, size_t stride,
                      sslam_keypoint* kp_out, uint8_t* desc_out, int cap, int* n_out);

/* Batch-of-frames mode (north_star): nframes independent images of identical size
 * resident in HBM.  d_images + i*image_stride is frame i (row pitch `pitch`).
 * d_kp[nframes*cap], d_desc[nframes*cap*32], d_counts[nframes] are device buffers.
 * Level 0 of the pyramid is READ IN PLACE from d_images when base, pitch and image_stride are multiples of 4 and the frames are whole blocks
 * (pitch == w, or image_stride >= pitch * h): the kernels load aligned dwords and may touch the padding bytes [w, pitch) of a row, so with
 * padded rows every frame -- the last one included -- must be readable for pitch * h bytes.  Any other layout is copied first (one more pass).
 * The images must stay valid until the call's work on `stream` has finished (as for any asynchronous launch). */
int sslam_orb_extract_batch_dev(sslam_orb* orb, const uint8_t* d_images, int w, int h,
                                size_t pitch, size_t image_stride, int nframes,
                                sslam_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts,
                                int cap, void* stream);

/* Batch status.  The batch kernels clamp a frame's rows to `cap` (d_counts[i] <= cap) and cannot return a status from the device, so a
 * batch caller asks afterwards: frames of the LAST batch on this handle whose keypoint total exceeded `cap` -- the condition for which the
 * single-frame call returns SSLAM_ERR_CAPACITY.  The host form synchronises the stream and returns SSLAM_ERR_CAPACITY when there is one
 * (first_frame_out = its index, -1 if none); the _dev form enqueues the check and writes d_status4 = {frames over capacity, first such frame
 * (INT_MAX if none), 0, INT_MAX}.  sslam_frontend_batch performs both checks per chunk itself. */
int sslam_orb_batch_status(sslam_orb* orb, int cap, void* stream, int* truncated_frames_out, int* first_frame_out);
int sslam_orb_batch_status_dev(sslam_orb* orb, int cap, int32_t* d_status4, void* stream);

/* Stage taps for stage-by-stage parity tests (not used by the drop-in shim):
 * copy pyramid level `level` of frame `frame` of the LAST batch to host (unpadded,
 * contiguous w*h), and the FAST candidate list (x,y,score triplets relative to
 * minBorder, reference src/ORBextractor.cc:820-825) of that level.
 * Level 0 (and the blurred patches of level-0 keypoints) of a batch that was read in place come from the CALLER's image buffer of that
 * batch: it must still be alive when a tap is called (sslam_orb_extract keeps its own copy; a freed device buffer is the caller's error). */
int sslam_orb_debug_level(sslam_orb* orb, int frame, int level, uint8_t* out, int* w, int* h);
int sslam_orb_debug_candidates(sslam_orb* orb, int frame, int level, int32_t* xys_out, int cap, int* n_out);
/* Stage tap of GaussianBlur(7x7, sigma 2) on the level clones (src/ORBextractor.cc:1085-1086): the blurred levels are never stored
 * (the blur is fused into the descriptor kernel), so the tap re-runs that kernel's two blur passes over the selection of frame
 * `frame` of the LAST batch and returns, per keypoint in output order, its KeyPoint record and the blurred 37x37 window
 * (|dx|,|dy| <= 18, row-major, in level coordinates) around it: kp_out[cap], patches_out[cap*37*37]. */
int sslam_orb_debug_blur_patches(sslam_orb* orb, int frame, sslam_keypoint* kp_out, uint8_t* patches_out, int cap, int* n_out);

/* ---- Hamming matching (replaces DescriptorDistance / BFMatcher / Search*) -- */
/* cv::BFMatcher(NORM_HAMMING,false).knnMatch(q,t,matches,2) as used at
 * src/LSDmatcher.cpp:150-155,261-266,293-298,336-341,387-392 and
 * ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1650-1666).
 * idx/dist are nq x 2 (best, second); missing neighbours (nt<2) are idx=-1,dist=-1.
 * Ties resolve to the lower train index. */
int sslam_hamming_knn2(sslam_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt,
                       int32_t* idx_out, int32_t* dist_out);
int sslam_hamming_knn2_dev(sslam_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt,
                           int32_t* d_idx, int32_t* d_dist, void* stream);
/* Batch form: frame f = rows [f*cap, f*cap+nq[f]) of d_q against rows [f*cap, f*cap+nt[f]) of d_t;
 * d_idx/d_dist are nframes*cap x 2.  For cap <= 4096 the train rows are first expanded to matrix-core operands in a buffer the CONTEXT keeps
 * (256 B per train row, nframes * cap rows, never shrunk: 3 GB for 12 288 frames of 1 000 rows).  Calls on different streams of one context are
 * ordered on that buffer by an event (a later call's expansion waits for the earlier call's search), so they do not overlap; use one context per
 * stream for concurrent searches. */
int sslam_hamming_knn2_batch_dev(sslam_ctx* ctx, const uint8_t* d_q, const int32_t* d_nq, const uint8_t* d_t,
                                 const int32_t* d_nt, int cap, int nframes, int32_t* d_idx, int32_t* d_dist, void* stream);
/* Dense nq x nt distance matrix (uint16), the input of the windowed Search* variants. */
int sslam_hamming_matrix(sslam_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* D);

/* ORBmatcher::SearchForInitialization(F1,F2,vbPrevMatched,vnMatches12,windowSize),
 * src/ORBmatcher.cc:408-523, with Frame::GetFeaturesInArea (src/Frame.cc:368-421) and
 * the 64x48 grid (src/Frame.cc:133-148,462-472) evaluated on the device.
 * kp1/desc1 (n1) = F1.mvKeysUn/mDescriptors, kp2/desc2 (n2) = F2; prev_matched[n1*2]
 * in/out (vbPrevMatched); matches12[n1] out; bounds = mnMinX,mnMaxX,mnMinY,mnMaxY. */
int sslam_orb_search_for_initialization(sslam_ctx* ctx,
        const sslam_keypoint* kp1, const uint8_t* desc1, int n1,
        const sslam_keypoint* kp2, const uint8_t* desc2, int n2,
        float* prev_matched, int32_t* matches12, int window_size,
        float nnratio, int check_orientation, const float bounds[4], int* nmatches_out);
/* Batch/device form: frame pair p uses kp1 + p*cap etc.; counts n1[p], n2[p] on device. */
int sslam_orb_search_for_initialization_batch_dev(sslam_ctx* ctx,
        const sslam_keypoint* d_kp1, const uint8_t* d_desc1, const int32_t* d_n1,
        const sslam_keypoint* d_kp2, const uint8_t* d_desc2, const int32_t* d_n2,
        int cap, int npairs, float* d_prev_matched, int32_t* d_matches12, int32_t* d_nmatches,
        int window_size, float nnratio, int check_orientation, const float bounds[4], void* stream);

/* The projection-window matcher family.  The tracker keeps the projection / visibility logic (poses, map
 * points: host state) and hands over one query per map point or map line; the device replays the
 * window search, the best/second-best Hamming selection, the level-consistent ratio test, the
 * "keypoint already taken" rule and the rotation-histogram pruning of:
 *   kind 0, mode 0  ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)   src/ORBmatcher.cc:45-129
 *   kind 0, mode 1  ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)        src/ORBmatcher.cc:1331-1473
 *   kind 1, mode 0  LSDmatcher::SearchByProjection(Frame&, const vector<MapLine*>&, th)    src/LSDmatcher.cpp:185-255
 *                   LSDmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)        src/LSDmatcher.cpp:22-141
 * Mode 1 is the "best candidate only, a feature that already holds a map point is skipped" rule; three more overloads are the same
 * rule under other arguments (each checked against its own line-by-line restatement, tests/test_match_gpu.py):
 *   kind 0, mode 1  ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)  src/ORBmatcher.cc:1475-1602 (relocalisation):
 *                   occupied[i] = CurrentFrame.mvpMapPoints[i] != NULL, obs_positive = 1, uright = NULL, th_dist = ORBdist,
 *                   min/max_level = pred-1 / pred+1, angle = pKF->mvKeysUn[i].angle
 *   kind 0, mode 1  ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)        src/ORBmatcher.cc:293-406 (loop closing):
 *                   occupied[i] = vpMatched[i] != NULL, obs_positive = 1, uright = NULL, th_dist = TH_LOW, check_orientation = 0,
 *                   min/max_level = pred-1 / pred (KeyFrame::GetFeaturesInArea, src/KeyFrame.cc:610-649, scans like Frame's)
 *   kind 1, mode 1  LSDmatcher::SearchByProjection(KeyFrame*, Scw, vpLines, vpMatched, th)         src/LSDmatcher.cpp:558-683: as the line above
 *                   with both projected endpoints (kind 1 accepts mode 1 only with check_orientation = 0)
 * feats = F.mvKeysUn (sslam_keypoint) or F.mvKeylinesUn (sslam_keyline), desc = F.mDescriptors / F.mLdesc,
 * uright = F.mvuRight or NULL (monocular), occupied[i] = F.mvpMapPoints[i] has Observations()>0 (or NULL).
 * assigned_out[i] = index of the query now owning feature i (-> F.mvpMapPoints[i] = that pMP), -1 if no query touched it (the pointer
 * keeps its value), or -2 (mode 1 with check_orientation only) if it was matched and then removed by the rotation-consistency check
 * (the reference writes NULL there, src/ORBmatcher.cc:1465, :1596). */
typedef struct sslam_proj_query {
    float u, v;            /* window centre (mTrackProjX/Y or projected u,v); lines: first projected endpoint */
    float u2, v2;          /* lines: second projected endpoint */
    float radius;          /* r * mvScaleFactors[level] resp. th * mvScaleFactors[octave] */
    int32_t min_level, max_level;   /* the GetFeaturesInArea / GetLinesInArea level arguments */
    float angle;           /* mode 1: LastFrame.mvKeysUn[i].angle (rotation histogram) */
    float ur;              /* stereo: mTrackProjXR resp. u - mbf*invzc; ignored when uright == NULL */
    int32_t valid;         /* 0: skipped (not in view, bad, outlier, behind the camera, outside the image) */
    int32_t obs_positive;  /* the map point / line has Observations() > 0 */
} sslam_proj_query;
int sslam_search_by_projection(sslam_ctx* ctx, int kind, int mode, const void* feats, const uint8_t* desc, int n,
                               const float bounds[4], const float* uright, const uint8_t* occupied,
                               const sslam_proj_query* queries, const uint8_t* qdesc, int nq,
                               float nnratio, int th_dist, int check_orientation,
                               int32_t* assigned_out, int* nmatches_out);

/* ---- device-resident frames (no single reference function: replaces the re-upload of
 * Frame::mvKeysUn / mDescriptors / mvKeylinesUn / mLdesc (include/Frame.h:155-189) around every matcher call) -------------
 * A frame handle owns device copies of one Frame's features (kind 0: sslam_keypoint rows, 1: sslam_keyline rows), their
 * 32-byte descriptors, optionally mvuRight, and the image bounds {mnMinX, mnMaxX, mnMinY, mnMaxY}. */
typedef struct sslam_frame sslam_frame;
int sslam_frame_upload(sslam_ctx* ctx, int kind, const void* feats, const uint8_t* desc, int n, const float* uright, const float bounds[4],
                       sslam_frame** out);
/* Snapshot (device to device) of what the last sslam_orb_extract / sslam_lines_extract call on this handle produced:
 * Frame::ExtractORB / ExtractLSD (src/Frame.cc:150-161) followed by the matchers without any upload. */
int sslam_frame_from_orb(sslam_orb* orb, const float bounds[4], sslam_frame** out);
int sslam_frame_from_lines(sslam_lines* lines, const float bounds[4], sslam_frame** out);
void sslam_frame_destroy(sslam_frame* frame);
int sslam_frame_count(const sslam_frame* frame);
/* sslam_search_by_projection with the frame's features taken from the handle (same kernels, same results). */
int sslam_search_by_projection_frame(sslam_ctx* ctx, const sslam_frame* frame, int mode, const uint8_t* occupied,
                                     const sslam_proj_query* queries, const uint8_t* qdesc, int nq,
                                     float nnratio, int th_dist, int check_orientation, int32_t* assigned_out, int* nmatches_out);
/* sslam_hamming_knn2 between the descriptors of two frame handles (cv::BFMatcher::knnMatch(d1, d2, m, 2),
 * src/LSDmatcher.cpp:150,261,293,336,387). */
int sslam_hamming_knn2_frames(sslam_ctx* ctx, const sslam_frame* query, const sslam_frame* train, int32_t* idx, int32_t* dist);

/* The candidate search of the Fuse family on a device-resident keyframe:
 *   ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th)            src/ORBmatcher.cc:828-960   chi2_mode 1 (stereo 7.8 / mono 5.99 gates, :913-936)
 *   ORBmatcher::Fuse(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, th, vpReplacePoint) :962-1103 chi2_mode 0
 *   LSDmatcher::Fuse(KeyFrame*, const vector<MapLine*>&, th)             src/LSDmatcher.cpp:417-548  chi2_mode 0, keyline frame
 * One sslam_proj_query per projected map point / line: u,v (and u2,v2 for lines), radius = th*mvScaleFactors[pred], ur,
 * min_level = pred-1, max_level = pred, valid.  best_idx_out[q] = the feature with the smallest descriptor distance in the
 * window (first in KeyFrame::GetFeaturesInArea / GetLinesInArea order on ties, src/KeyFrame.cc:610-683), -1 if none;
 * best_dist_out[q] = that distance (INT_MAX if none).  inv_level_sigma2 = KeyFrame::mvInvLevelSigma2 (chi2_mode 1 only).
 * The `bestDist <= TH_LOW` test and the Replace / AddObservation bookkeeping stay with the caller, in query order.
 * The same inner loop (chi2_mode 0) is the candidate search of ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1196-1227, :1276-1307,
 * gate TH_HIGH), LSDmatcher::SearchBySim3 (src/LSDmatcher.cpp:685-929) and LSDmatcher::Fuse(KeyFrame*, Scw, ...) (:931-1063).  The
 * loop-closing SearchByProjection(KeyFrame*, Scw, ...) overloads skip features that are already matched, which makes their queries
 * order dependent: they are sslam_search_by_projection mode 1 (above), not this function. */
int sslam_fuse_search(sslam_ctx* ctx, const sslam_frame* keyframe, int chi2_mode, const float* inv_level_sigma2, int nlevels,
                      const sslam_proj_query* queries, const uint8_t* qdesc, int nq, int32_t* best_idx_out, int32_t* best_dist_out);

/* ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, vector<pair<size_t,size_t>>&, bOnlyStereo),
 * src/ORBmatcher.cc:660-826 (epipolar test CheckDistEpipolarLine :140-157), on two device-resident keyframes (their
 * mvuRight are in the handles).  free1[i] / free2[i] = !GetMapPoint(i); the two FeatureVectors as CSR lists over the shared
 * vocabulary nodes in ascending node id (see sslam_orb_search_by_bow); F12 row-major (F12.at<float>(r,c) = F12[3r+c]);
 * (ex, ey) = epipole in image 2 (:667-673); scale_factors2 = pKF2->mvScaleFactors, level_sigma2_2 = pKF2->mvLevelSigma2.
 * matches12_out[i1] = matched keyframe-2 feature or -1 (the reference's vMatches12 after the rotation-histogram pruning). */
int sslam_orb_search_for_triangulation(sslam_ctx* ctx, const sslam_frame* kf1, const sslam_frame* kf2, const uint8_t* free1, const uint8_t* free2,
                                       const int32_t* node_kf1_ptr, const int32_t* node_kf2_ptr, int nnodes, const int32_t* kf1_idx, const int32_t* kf2_idx,
                                       const float F12[9], float ex, float ey, const float* scale_factors2, const float* level_sigma2_2, int nlevels,
                                       int only_stereo, int check_orientation, int32_t* matches12_out, int* nmatches_out);

/* ---- DBoW2 vocabulary descent: Frame::ComputeBoW, src/Frame.cc:474-481 ->
 * ORBVocabulary::transform(vCurrentDesc, mBowVec, mFeatVec, 4), whose per-feature work is
 * TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup), Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1216-1259.
 * The vocabulary is handed over as arrays in DBoW2's node numbering (node 0 = root): children of node i are
 * children[child_ptr[i] .. child_ptr[i+1]) in m_nodes[i].children order (a node without children is a leaf), node_desc =
 * 32-byte descriptor per node, word_id / weight per node (meaningful for leaves), levels = m_L (sslam_vocab_load_text below builds
 * exactly these arrays from an ORBvoc.txt-format file).  Outputs per feature: the word, its weight (0 = stopped
 * word) and the node passed at level m_L - levelsup; BowVector / FeatureVector are std::maps the caller fills from them
 * in feature order (v.addWeight(id, w); fv.addFeature(nid, i); then the L1 normalisation, TemplatedVocabulary.h:1147-1199). */
typedef struct sslam_vocab sslam_vocab;
int sslam_vocab_create(sslam_ctx* ctx, int nnodes, int levels, const int32_t* child_ptr, const int32_t* children, const uint8_t* node_desc,
                       const int32_t* word_id, const double* weight, sslam_vocab** out);
void sslam_vocab_destroy(sslam_vocab* vocab);
int sslam_bow_transform(sslam_ctx* ctx, const sslam_vocab* vocab, const uint8_t* desc, int n, int levelsup,
                        int32_t* word_out, double* weight_out, int32_t* node_out);
int sslam_bow_transform_frame(sslam_ctx* ctx, const sslam_vocab* vocab, const sslam_frame* frame, int levelsup,
                              int32_t* word_out, double* weight_out, int32_t* node_out);

/* ORBVocabulary::loadFromTextFile (src/System.cc:64-73 -> TemplatedVocabulary.h:1338-1423): first line "k L scoring weighting"
 * (rejected outside 0<=k<=20, 1<=L<=10, 0<=scoring<=5, 0<=weighting<=3, as the reference does), then one line per node in id order
 * starting at 1: "parent isLeaf d0 .. d31 weight"; children keep file order, word ids count the leaves in file order.  Blank lines
 * are skipped (the reference's getline loop appends a node from uninitialised values after the final newline: undefined behaviour,
 * not reproduced).  sslam_vocab_set_types overrides what sslam_vocab_create assumes (TF_IDF = 0, L1_NORM = 0, the enums of
 * Thirdparty/DBoW2/DBoW2/BowVector.h:36-53). */
int sslam_vocab_load_text(sslam_ctx* ctx, const char* path, sslam_vocab** out);
int sslam_vocab_set_types(sslam_vocab* vocab, int weighting, int scoring);
int sslam_vocab_info(const sslam_vocab* vocab, int* k, int* levels, int* scoring, int* weighting, int* nnodes, int* nwords);

/* Frame::ComputeBoW / KeyFrame::ComputeBoW (src/Frame.cc:474-481, src/KeyFrame.cc:74-84) complete:
 * TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup), TemplatedVocabulary.h:1126-1208, with
 * BowVector::addWeight / addIfNotExist / normalize (BowVector.cpp:34-85) and FeatureVector::addFeature (FeatureVector.cpp:30-44).
 * The descent runs on the device, the two std::maps are assembled on the host in feature order and returned flattened in key
 * order: BowVector = (bow_word[i], bow_value[i]) for i < *nbow; FeatureVector = node fv_node[j] owns the feature indices
 * fv_feat[fv_ptr[j] .. fv_ptr[j+1]) for j < *nfv.  Capacities: n entries each, n + 1 for fv_ptr. */
int sslam_compute_bow(sslam_ctx* ctx, const sslam_vocab* vocab, const uint8_t* desc, int n, int levelsup,
                      int32_t* bow_word, double* bow_value, int* nbow, int32_t* fv_node, int32_t* fv_ptr, int32_t* fv_feat, int* nfv);
int sslam_compute_bow_frame(sslam_ctx* ctx, const sslam_vocab* vocab, const sslam_frame* frame, int levelsup,
                            int32_t* bow_word, double* bow_value, int* nbow, int32_t* fv_node, int32_t* fv_ptr, int32_t* fv_feat, int* nfv);

/* MapPoint::ComputeDistinctiveDescriptors, src/MapPoint.cc:247-312, and MapLine::ComputeDistinctiveDescriptors,
 * src/MapLine.cpp:246-317, for nsets observation sets at once: set s owns rows ptr[s]..ptr[s+1]
 * of desc (ptr[0] = 0).  best_out[s] = index inside the set of the descriptor with the least median Hamming distance to
 * the rest (median = sorted[int(0.5*(N-1))], first row wins ties), -1 for an empty set.  Sets are limited to 1024 rows. */
int sslam_distinctive_descriptors(sslam_ctx* ctx, const uint8_t* desc, const int32_t* ptr, int nsets, int32_t* best_out);

/* ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches), src/ORBmatcher.cc:159-291.
 * The DBoW2 vocabulary transform stays on the host (the vocabulary file is not part of the reference tree); the
 * matcher takes the two FeatureVectors as CSR lists over the nodes BOTH frames contain, ascending node id:
 * node k owns keyframe features kf_idx[node_kf_ptr[k]..node_kf_ptr[k+1]) and frame features f_idx[node_f_ptr[k]..).
 * kf_valid[i] = vpMapPointsKF[i] && !isBad().  assigned_out[j] = keyframe feature matched to frame feature j, or -1. */
int sslam_orb_search_by_bow(sslam_ctx* ctx, const sslam_keypoint* kf_kp, const uint8_t* kf_desc, const uint8_t* kf_valid, int nkf,
                            const sslam_keypoint* f_kp, const uint8_t* f_desc, int nf,
                            const int32_t* node_kf_ptr, const int32_t* node_f_ptr, int nnodes,
                            const int32_t* kf_idx, const int32_t* f_idx, float nnratio, int check_orientation,
                            int32_t* assigned_out, int* nmatches_out);

/* ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12), src/ORBmatcher.cc:525-658 (loop closing,
 * src/LoopClosing.cc:271): same node walk; kfX_valid[i] = vpMapPointsX[i] && !isBad(); a candidate of the second keyframe is skipped
 * when it is already matched or has no good map point (:576-580); the gate is `bestDist1 < TH_LOW` (strict, :600).
 * matches12_out[i] = feature of the second keyframe whose map point goes into vpMatches12[i], or -1. */
int sslam_orb_search_by_bow_keyframes(sslam_ctx* ctx, const sslam_keypoint* kf1_kp, const uint8_t* kf1_desc, const uint8_t* kf1_valid, int n1,
                                      const sslam_keypoint* kf2_kp, const uint8_t* kf2_desc, const uint8_t* kf2_valid, int n2,
                                      const int32_t* node_kf1_ptr, const int32_t* node_kf2_ptr, int nnodes,
                                      const int32_t* kf1_idx, const int32_t* kf2_idx, float nnratio, int check_orientation,
                                      int32_t* matches12_out, int* nmatches_out);

/* LSDmatcher::SerachForInitialize(InitialFrame,CurrentFrame,LineMatches),
 * src/LSDmatcher.cpp:257-284 = knn2 + Frame::lineDescriptorMAD (src/Frame.cc:190-215)
 * + the `d2-d1 > 0.5*MAD12` gate.  pairs_out[cap*2] (qdx,tdx); gate_scale = 0.5
 * (0.1 reproduces SearchForTriangulation, src/LSDmatcher.cpp:396).
 * ratio_mode != 0 uses the `d1/d2 < 1/1.5` gate of SearchByProjection(KF,F)/
 * SearchByDescriptor (src/LSDmatcher.cpp:158-169,303-314) instead. */
int sslam_line_match(sslam_ctx* ctx, const uint8_t* ldesc1, int n1, const uint8_t* ldesc2, int n2,
                     double gate_scale, int ratio_mode, int32_t* pairs_out, int cap, int* npairs_out,
                     double* nn_mad_out, double* nn12_mad_out);
int sslam_line_match_batch_dev(sslam_ctx* ctx, const uint8_t* d_ldesc1, const int32_t* d_n1,
                               const uint8_t* d_ldesc2, const int32_t* d_n2, int cap, int npairs_frames,
                               double gate_scale, int ratio_mode, int32_t* d_pairs, int32_t* d_npairs,
                               void* stream);

/* ---- Lines (replaces LineSegment::ExtractLineSegment) --------------------- */
/* LineSegment::ExtractLineSegment(img,keylines,ldesc,keylineFunctions,scale,numOctaves),
 * src/ExtractLineSegment.cpp:18-69 = LSDDetector::detect + top-N by response +
 * BinaryDescriptor::compute + normalised line equations.  max_lines = 40 reproduces
 * the reference's hard cap (:42); BASELINE configs use 200 / 400. */
int sslam_lines_create(sslam_ctx* ctx, int max_lines, sslam_lines** out);
int sslam_lines_destroy(sslam_lines* ln);
int sslam_lines_extract(sslam_lines* ln, const uint8_t* gray, int w, int h, size_t stride,
                        sslam_keyline* kl_out, uint8_t* ldesc_out, double* linefn_out /*3 per line*/,
                        int cap, int* n_out);
/* The line twin of sslam_orb_set_blur_variant: which 8-bit cv::GaussianBlur LBD's 5x5 sigma-1 pre-blur is (BinaryDescriptor::compute, called at
 * src/ExtractLineSegment.cpp:53): 0 = OpenCV >= 3.4.1 (taps 14 62 104 62 14), 1 = OpenCV 3.4.0 (14 63 103 63 14).  LSD's own 7x7 sigma-0.75 pre-blur has the
 * same taps (0 4 56 136 56 4 0) under both, so the segments do not depend on it; the LBD bytes do. */
int sslam_lines_set_blur_variant(sslam_lines* ln, int variant);
/* The other stated decisions of the line path that have a selectable alternative.  The arithmetic of LSD and LBD is OpenCV's (called at
 * src/ExtractLineSegment.cpp:38-40,53); this image holds no OpenCV, so each of these is a choice between two restatements, with the size of what it moves
 * measured in oracle/ref_pin/pin_report_stub.json.  variant is 0 or 1; the next extraction takes it.
 *   sslam_lines_set_nfa_variant    D11 -- the first term of LineSegmentDetectorImpl::nfa()'s log1term.  1 (DEFAULT since round 5) = `(double(n) + 1)` without the log_gamma,
 *                                  as imgproc/src/lsd.cpp is recalled to read (two independent recollections); 0 = log_gamma(n + 1), the binomial coefficient of von Gioi's
 *                                  lsd.c (the default of rounds 1-4).  Under 1 nearly every rectangle passes at its first rect_nfa: 2.0-2.7 x the segments of 0.
 *   sslam_lines_set_lbd_bit_order  D12 -- BinaryDescriptor::binaryConversion.  1 (DEFAULT since round 5) = comparison i sets `0x80 >> i` (as recalled); 0 = comparison i
 *                                  sets bit i (rounds 1-4).  Every byte is bit-reversed; Hamming distances, hence every matcher result, are the same.
 *   sslam_lines_set_resize_variant D7 -- LSD's 0.8x rescale: 0 (default) = INTER_LINEAR_EXACT (8.8 coefficients, one rounding); 1 = INTER_LINEAR (11-bit coefficients,
 *                                  the two-stage 8u rounding of the ORB pyramid's resize).
 *   sslam_lines_set_seed_order     D2 -- the order of LSD's seeds inside one of the 1024 gradient bins: 0 (default) = raster order (a stable counting sort, on the device);
 *                                  1 = whatever `std::sort` of this library's libstdc++ leaves, as upstream's ll_angle sorts: the bins are sorted ON THE HOST (one
 *                                  download, one std::sort of every pixel and one upload per frame: ~15 ms per 640x480 frame -- a switch for comparing with a
 *                                  maintainer's CPU build, not a production path). */
int sslam_lines_set_nfa_variant(sslam_lines* ln, int variant);
int sslam_lines_set_lbd_bit_order(sslam_lines* ln, int variant);
int sslam_lines_set_resize_variant(sslam_lines* ln, int variant);
int sslam_lines_set_seed_order(sslam_lines* ln, int variant);
int sslam_lines_extract_batch_dev(sslam_lines* ln, const uint8_t* d_images, int w, int h,
                                  size_t pitch, size_t image_stride, int nframes,
                                  sslam_keyline* d_kl, uint8_t* d_ldesc, double* d_linefn,
                                  int32_t* d_counts, int cap, void* stream);
/* Batch status of the line extractor (see sslam_orb_batch_status): frames of the last batch holding more lines than `cap`
 * (SSLAM_ERR_CAPACITY) and frames whose LSD stage overflowed its 8192 candidate rectangles (SSLAM_ERR_UNSUPPORTED, as the single-frame call
 * reports it).  d_status4 = {frames over capacity, first of them, frames over the LSD limit, first of them}. */
int sslam_lines_batch_status(sslam_lines* ln, int cap, void* stream, int* truncated_frames_out, int* unsupported_frames_out, int* first_frame_out);
int sslam_lines_batch_status_dev(sslam_lines* ln, int cap, int32_t* d_status4, void* stream);
/* Scheduling hint for a caller that keeps a second stream busy while a batch of line extractions is in flight (bench.py's two-stream mode;
 * the reference has no batch mode and no counterpart).  `hip_event` (a hipEvent_t; NULL clears it) is recorded on the stream of every
 * following sslam_lines_extract_batch_dev call immediately BEFORE the sequential core (k_lsd_regions) is launched.  The core is latency-bound
 * and leaves issue slots free, the kernels in front of it are bandwidth-bound and do not: a second stream that waits for the event overlaps
 * the core instead of the prologue.  The event stays the caller's. */
int sslam_lines_set_core_event(sslam_lines* ln, void* hip_event);
/* With a core event set, a batch of at least 32 frames per compute unit (two rounds of 16 workgroups per CU; 18 per CU from 36 frames per compute unit on) runs the sequential core in its GUEST form: a persistent grid of four-wave workgroups that
 * claim frames dynamically and leave a third of every SIMD's registers to the caller's second stream.  Returns 1 when a call of `nframes` frames
 * would take that form, 0 otherwise.  A caller in the guest form gets the most from building the ORB pyramid BEFORE the event (sslam_orb_set_gate_event: FAST and
 * what follows wait for it) -- in the other form from making its whole second branch wait for the event (bench.py / pipeline.py / sslam_frontend_batch do exactly that). */
int sslam_lines_core_guest_form(sslam_lines* ln, int nframes);
/* The counterpart on the point side: `hip_event` (a hipEvent_t; NULL clears it) is WAITED for on the stream of every following
 * sslam_orb_extract_batch_dev call after the pyramid kernels (level copy, resizes: bandwidth-bound like the line branch's prologue) and before
 * the FAST / quadtree / descriptor kernels (vector-bound: the ones that should run under the sequential core).  With the core event here the
 * point stream needs no wait of its own: the pyramid is built beside the line prologue, the rest starts with the core. */
int sslam_orb_set_gate_event(sslam_orb* orb, void* hip_event);
/* Two extractors that take turns (a batch processed as two half batches, each on its own streams): the sequential core wants every wave slot
 * of the chip, so two cores must not overlap -- but one half's kernels BEHIND its core (NFA stages, LBD: VALU-bound) overlap well with the
 * other half's kernels IN FRONT of its core (blur, gradient, counting sort: bandwidth-bound).  Every following sslam_lines_extract_batch_dev
 * of `ln` makes its stream wait for `wait_event` right before the core (NULL: no wait; an event that was never recorded does not block) and
 * records `done_event` right behind it (NULL: nothing).  The events stay the caller's. */
int sslam_lines_set_core_gate(sslam_lines* ln, void* wait_event, void* done_event);
/* Host-buffer batch (SURVEY.md §8(b) `sslam_frontend_batch`): n frames of one size in host memory through Frame::ExtractORB and, when
 * `lines` is not NULL, Frame::ExtractLSD (src/Frame.cc:150-161); frame i starts at images + i*image_stride (row pitch `stride`).
 * Per-frame results in the caller's arrays: kp_out[n*cap], desc_out[n*cap*32], nkp_out[n], kl_out[n*lcap], ldesc_out[n*lcap*32],
 * linefn_out[n*lcap*3], nl_out[n]; rows past a frame's count are unspecified.  A frame with more keypoints than `cap` / more lines than
 * `lcap` makes the call return SSLAM_ERR_CAPACITY, an LSD overflow SSLAM_ERR_UNSUPPORTED (as the single-frame entry points do; the arrays
 * then hold the truncated rows).  Frames are processed in chunks of `chunk` (0 = as many as fill every wave slot of the sequential LSD core,
 * 6144, bounded by a third of the free device memory): the upload of chunk k+1 and the download of chunk k-1 overlap the kernels of chunk
 * k (two copy streams; pinned caller memory -- hipHostMalloc / hipHostRegister -- is copied directly, pageable memory through pinned staging
 * buffers filled by a few host threads, SSLAM_BATCH_THREADS); the point branch and the line branch of a chunk run on two HIP streams, the point
 * branch released when the sequential core starts (sslam_lines_set_core_event).  Staging buffers, streams and events are kept per context
 * between calls (sslam_frontend_batch_release frees them early).  This is the PCIe-inclusive form of the batch mode; callers that already
 * hold their frames in HBM use the *_batch_dev entry points directly. */
int sslam_frontend_batch(sslam_orb* orb, sslam_lines* lines, const uint8_t* images, int n, int w, int h, size_t stride, size_t image_stride, int chunk,
                         sslam_keypoint* kp_out, uint8_t* desc_out, int32_t* nkp_out, int cap,
                         sslam_keyline* kl_out, uint8_t* ldesc_out, double* linefn_out, int32_t* nl_out, int lcap);
int sslam_frontend_batch_release(sslam_ctx* ctx);
/* sslam_frontend_batch plus the match stage of BASELINE configs[2] ("extract + Hamming match vs previous frame"): the batch is a SEQUENCE,
 * frame i is matched against frame i-1 of the same call with the three matchers the resident-frame pipeline times
 *   ORBmatcher::SearchForInitialization(F1 = frame i-1, F2 = frame i, vbPrevMatched = F1's keypoint positions, window_size)
 *                                                       src/ORBmatcher.cc:408-523 as called at src/Tracking.cc:330-345,
 *   the dense Hamming 2-NN of F1's descriptors in F2's (cv::BFMatcher::knnMatch, the reduction under every Search*; optional),
 *   LSDmatcher::SerachForInitialize(F1, F2) on the LBD descriptors (src/LSDmatcher.cpp:257-284; only with `lines`),
 * on the device, behind the chunk's extraction (point matchers on the point stream, the line matcher on the line stream).  Rows of frame i
 * describe the pair (i-1, i) and are indexed by F1's features: init_matches12[i*cap + j] = keypoint of frame i matched to keypoint j of
 * frame i-1 or -1 (j < nkp_out[i-1]), init_nmatches[i]; knn_idx / knn_dist[(i*cap + j)*2 + {0,1}]; line_pairs[(i*lcap + p)*2 + {0,1}] =
 * (line of frame i-1, line of frame i) for p < line_npairs[i].  Frame 0 has no predecessor: its counts are 0 and its rows unspecified.
 * Chunking, staging, streams and error behaviour are those of sslam_frontend_batch (chunk boundaries do not show: the last frame of a
 * chunk stays on the device as the next chunk's predecessor); the match arrays follow the same pinned-or-staged rule as the others. */
typedef struct sslam_batch_match {
    int32_t window_size; float nnratio; int32_t check_orientation; float bounds[4];      /* SearchForInitialization: windowSize (100), mfNNratio (0.9), mbCheckOrientation, mnMinX/MaxX/MinY/MaxY */
    double line_gate_scale; int32_t line_ratio_mode;                                    /* sslam_line_match arguments (0.5, 0) */
    int32_t* init_matches12;      /* [n*cap] */
    int32_t* init_nmatches;       /* [n] */
    int32_t* knn_idx;             /* [n*cap*2] or NULL (then knn_dist is NULL as well) */
    int32_t* knn_dist;            /* [n*cap*2] */
    int32_t* line_pairs;          /* [n*lcap*2]; ignored without `lines` */
    int32_t* line_npairs;         /* [n] */
} sslam_batch_match;
int sslam_frontend_batch_match(sslam_orb* orb, sslam_lines* lines, const uint8_t* images, int n, int w, int h, size_t stride, size_t image_stride, int chunk,
                               sslam_keypoint* kp_out, uint8_t* desc_out, int32_t* nkp_out, int cap,
                               sslam_keyline* kl_out, uint8_t* ldesc_out, double* linefn_out, int32_t* nl_out, int lcap, const sslam_batch_match* match);


/* ---- multi-GPU batch mode (SURVEY.md §8(b) "sslam_group_create + sslam_frontend_batch_sharded", §8(e)) ----------------------
 * north_star: "a batch-of-frames mode shards independent images across the 8 GPUs of one node with RCCL over xGMI only for the final
 * keypoint/line gather".  Frames are independent units: global frame i lives on GPU i % G.  Each GPU runs the single-GPU path on its
 * shard; the ONE exchange step is a gather of compacted per-frame records to the root GPU (grouped ncclSend / ncclRecv, sizes first).
 * There is no reference counterpart (the reference front-end is one thread on one CPU, src/Frame.cc:86-87); the caller-side loop this
 * replaces is the per-frame Frame::ExtractORB / ExtractLSD pair (src/Frame.cc:150-161).  RCCL is bound at run time (dlopen of
 * librccl.so.1, or the copy a host process such as PyTorch already loaded), so the library itself has no link-time RCCL dependency.
 *
 * Record stream (what travels over xGMI and what rank 0 ingests): per frame, 16-byte aligned,
 *   sslam_record_header {n_kp, n_ln, frame, bytes}  then  n_kp x sslam_keypoint (28 B), n_kp x 32 B descriptors,
 *   n_ln x sslam_keyline (68 B), n_ln x 32 B LBD descriptors, n_ln x 3 doubles (mvKeyLineFunctions, src/ExtractLineSegment.cpp:56-68)
 * -- compacted to the counts, not padded to the capacities: about 80 KB per 640x480 frame at 1000 keypoints / 165 lines. */
typedef struct sslam_group sslam_group;
#define SSLAM_GROUP_ID_BYTES 128
typedef struct sslam_record_header { int32_t n_kp, n_ln, frame, bytes; } sslam_record_header;
typedef struct sslam_frontend_params {     /* ORBextractor ctor arguments (Examples/ICL.yaml:41-54) + the line cap; max_lines 0 = no line extraction */
    int32_t nfeatures; float scale_factor; int32_t nlevels, ini_th_fast, min_th_fast, max_lines;
} sslam_frontend_params;

/* One process drives `ngpu` devices (0 .. ngpu-1): one context, extractor pair and host thread per device, communicators from
 * ncclCommInitAll.  This is the form a C++ host like the reference's (one process) uses. */
int sslam_group_create(int ngpu, sslam_group** out);
/* One process per GPU (torch.distributed / MPI style): rank 0 obtains an id, the launcher broadcasts it, every rank joins.  `device` is the
 * HIP device of this process. */
int sslam_group_unique_id(uint8_t id_out[SSLAM_GROUP_ID_BYTES]);
int sslam_group_create_rank(int device, int rank, int nranks, const uint8_t id[SSLAM_GROUP_ID_BYTES], sslam_group** out);
int sslam_group_destroy(sslam_group* group);
int sslam_group_size(const sslam_group* group);
int sslam_group_rank(const sslam_group* group);      /* 0 in the single-process form */

/* Compacts the per-frame results of a *_batch_dev call into the record stream above: frame i of the batch becomes the record with
 * header.frame = frame0 + i*frame_step.  d_kl / d_ldesc / d_linefn / d_nl may be NULL (no lines).  d_out[out_capacity]; *d_total_bytes
 * (device) receives the stream length, or UINT64_MAX when it does not fit.  Enqueued on `stream`, no synchronisation. */
int sslam_pack_records_dev(sslam_ctx* ctx, int nframes, int frame0, int frame_step,
                           const sslam_keypoint* d_kp, const uint8_t* d_desc, const int32_t* d_nkp, int cap,
                           const sslam_keyline* d_kl, const uint8_t* d_ldesc, const double* d_linefn, const int32_t* d_nl, int lcap,
                           uint8_t* d_out, uint64_t out_capacity, uint64_t* d_total_bytes, void* stream);
/* Upper bound of the stream length for nframes frames at the given capacities (buffer sizing). */
uint64_t sslam_record_stream_capacity(int nframes, int cap, int lcap);
/* Host side of the ingest: scatters a record stream (host memory) into per-frame arrays indexed by header.frame, laid out like the
 * outputs of sslam_frontend_batch (kp_out[nframes*cap] ...; line outputs may be NULL).  Returns SSLAM_ERR_INVALID on a malformed
 * stream, SSLAM_ERR_CAPACITY when a record exceeds cap / lcap.  *nrecords_out = records seen. */
int sslam_unpack_records(const uint8_t* stream, uint64_t bytes, int nframes,
                         sslam_keypoint* kp_out, uint8_t* desc_out, int32_t* nkp_out, int cap,
                         sslam_keyline* kl_out, uint8_t* ldesc_out, double* linefn_out, int32_t* nl_out, int lcap, int* nrecords_out);

/* The exchange step, one-process-per-GPU form: every rank hands over its device-resident record stream (length in *d_send_bytes, a
 * device word as written by sslam_pack_records_dev); rank 0 receives the streams of all ranks back to back in d_recv (rank order) and
 * gets the per-rank lengths in bytes_per_rank_out[nranks] (host).  Sizes travel first (one 8-byte ncclAllGather), then one grouped
 * ncclSend / ncclRecv per peer.  Synchronous on `stream` (NULL = an internal stream); call it from a side thread to overlap it with the
 * next batch's kernels.  d_recv / recv_capacity / bytes_per_rank_out are ignored on ranks other than 0. */
int sslam_group_gather_dev(sslam_group* group, const uint8_t* d_send, const uint64_t* d_send_bytes,
                           uint8_t* d_recv, uint64_t recv_capacity, uint64_t* bytes_per_rank_out, void* stream);

/* sslam_frontend_batch over all GPUs of a single-process group: n frames in host memory, frame i -> GPU i % G, per-GPU extraction in
 * chunks, RCCL gather of the records to GPU 0, results in the caller's arrays exactly as sslam_frontend_batch returns them (same
 * error behaviour).  The per-device extractors are created from `params` on first use and kept in the group. */
int sslam_frontend_batch_sharded(sslam_group* group, const sslam_frontend_params* params,
                                 const uint8_t* images, int n, int w, int h, size_t stride, size_t image_stride,
                                 sslam_keypoint* kp_out, uint8_t* desc_out, int32_t* nkp_out, int cap,
                                 sslam_keyline* kl_out, uint8_t* ldesc_out, double* linefn_out, int32_t* nl_out, int lcap);
/* How sslam_frontend_batch_sharded deals a batch over the GPUs (host-only bookkeeping, callable without a device): global frame i lives on
 * GPU i mod ngpu, every GPU walks its frames in chunks of *chunk_slots_out slots.  sslam_shard_frame = the global frame in slot `slot` of
 * chunk `chunk` on GPU `gpu` (-1: the slot is empty -- uneven tails); sslam_shard_chunk_count = how many leading slots of that chunk hold
 * a frame.  On any error of sslam_frontend_batch_sharded the outputs hold what arrived: nkp_out[i] (and nl_out[i]) = -1 marks a frame whose
 * records never reached the root (a HIP / RCCL failure on its GPU); truncated rows (SSLAM_ERR_CAPACITY) and LSD overflow
 * (SSLAM_ERR_UNSUPPORTED) are deferred statuses as in sslam_frontend_batch -- every frame is still delivered, clamped. */
int sslam_shard_layout(int n, int ngpu, int* chunk_slots_out, int* nchunks_out);
int sslam_shard_frame(int n, int ngpu, int chunk, int gpu, int slot);
int sslam_shard_chunk_count(int n, int ngpu, int chunk, int gpu);

/* ---- Stage taps (diagnostics).  They read back intermediate results of the LAST extraction of a handle so that a maintainer who pins this
 * library against the reference built with OpenCV 3.4 can compare stage by stage (pyramid level bytes, FAST candidates, blurred patches: sslam_orb_debug_*, declared with
 * the extractor above; LSD segments before the top-N cut: below).  Synchronous; nothing on the product path calls them. */
/* Stage tap: all LSD segments (x1,y1,x2,y2 float) of frame `frame` of the last batch, before top-N. */
int sslam_lines_debug_segments(sslam_lines* ln, int frame, float* seg_out, int cap, int* n_out);
/* Stage clocks of the sequential LSD core for frame `frame` of the last call (grow, rect, refine, radius reduction, total, ...): zeros unless
 * the library was built with -DSSLAM_LSD_CYCLES (tools/lsd_cycles.py). */
int sslam_lines_debug_cycles(sslam_lines* ln, int frame, long long* out8);
#ifdef __cplusplus
}
#endif
#endif /* SSLAM_FRONTEND_H */
