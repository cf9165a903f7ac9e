/* orbx.h -- C ABI of the MI355X-native ORB front-end (liborbx.so).
 *
 * The reference (cheukwaylee/ORB_SLAM2_detailed_comments) has no FFI layer: the boundary of
 * the hot path is the C++ class surface of ORB_SLAM2::ORBextractor (include/ORBextractor.h:82-185)
 * and ORB_SLAM2::ORBmatcher (include/ORBmatcher.h:54-225).  Every entry point below names the
 * reference interface it replaces; compat/ORBextractor.h + compat/ORBmatcher.h are the drop-in
 * C++ classes a maintainer compiles against this ABI.
 *
 * Plain pointers and sizes only; no C++/torch/HIP types in any signature.  "device pointer"
 * arguments are ordinary HIP device addresses (e.g. torch.Tensor.data_ptr()).
 *
 * Threading (mirrors src/Frame.cc:158-168): distinct handles are fully concurrent (one HIP
 * stream + private workspace each); one handle is not re-entrant.
 */
#ifndef ORBX_H
#define ORBX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORBX_ABI_VERSION 1

/* Bit-compatible with cv::KeyPoint (28 bytes): what ORBextractor::operator() appends to
 * std::vector<cv::KeyPoint>& _keypoints (src/ORBextractor.cc:2016-2082). */
typedef struct orbx_keypoint {
    float x, y;       /* pt, in level-0 *padded-image* coordinates (fork semantics) */
    float size;       /* (float)(int)(31 * scale[octave]) */
    float angle;      /* degrees [0,360) */
    float response;   /* FAST score */
    int32_t octave;
    int32_t class_id; /* always -1 */
} orbx_keypoint;

typedef enum orbx_status {
    ORBX_OK = 0,
    ORBX_EMPTY_IMAGE = 1,   /* reference: silent return, outputs untouched (src/ORBextractor.cc:1966-1967) */
    ORBX_BAD_ARGUMENT = 2,
    ORBX_BAD_ASPECT = 3,    /* nIni == 0: reference divides by zero (src/ORBextractor.cc:1059-1063) */
    ORBX_CAPACITY = 4,      /* caller buffer or internal candidate capacity too small */
    ORBX_HIP_ERROR = 5,
    ORBX_NO_DEVICE = 6,
    ORBX_UNSUPPORTED = 7
} orbx_status;

enum { ORBX_PYRAMID_FORK_PADDED = 0 };           /* src/ORBextractor.cc:2165-2166 */
enum { ORBX_FP_GCC_FMA = 0, ORBX_FP_STRICT = 1 }; /* contraction of GET_VALUE, src/ORBextractor.cc:207-209 */

/* The five values ORBextractor's constructor takes (include/ORBextractor.h:104; read from YAML at
 * src/Tracking.cc:160-168) plus the parity-contract switches and device sizing hints. */
typedef struct orbx_params {
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
    int32_t pyramid_mode;        /* ORBX_PYRAMID_FORK_PADDED */
    int32_t fp_mode;             /* ORBX_FP_GCC_FMA (default parity contract) or ORBX_FP_STRICT */
    int32_t device;              /* HIP device ordinal; -1 = current device */
    int32_t max_batch;           /* frames in flight per call (>=1) */
    int32_t max_cand_per_cell;   /* candidate slots per FAST cell; 0 = exact worst case (never overflows) */
} orbx_params;

typedef struct orbx_handle orbx_handle;

/* ---- lifecycle: replaces `new ORBextractor(nFeatures, fScaleFactor, nLevels, fIniThFAST, fMinThFAST)`
 *      (src/Tracking.cc:171-182) ---------------------------------------------------------- */
orbx_status orbx_create(const orbx_params *params, orbx_handle **out);
void orbx_destroy(orbx_handle *h);
void orbx_default_params(orbx_params *p); /* TUM1.yaml values: 1000, 1.2, 8, 20, 7 */
const char *orbx_last_error(void);        /* thread-local text of the last failure */
const char *orbx_status_string(orbx_status s);
int orbx_abi_version(void);

/* ---- getters: GetLevels / GetScaleFactor(s) / GetInverseScaleFactors / GetScaleSigmaSquares /
 *      GetInverseScaleSigmaSquares (include/ORBextractor.h:120-170), mnFeaturesPerLevel, umax ---- */
int orbx_get_levels(const orbx_handle *h);
float orbx_get_scale_factor(const orbx_handle *h);
orbx_status orbx_get_scale_tables(const orbx_handle *h, float *scale, float *inv_scale,
                                  float *sigma2, float *inv_sigma2); /* nlevels floats each, may be NULL */
orbx_status orbx_get_features_per_level(const orbx_handle *h, int32_t *n_per_level);
orbx_status orbx_get_umax(const orbx_handle *h, int32_t *umax16);
/* output capacity that can never overflow: sum over levels of max(N_l + 3, 4 * nIni_l);
 * < 0: -(orbx_status) (e.g. -ORBX_BAD_ASPECT) */
int orbx_max_keypoints(orbx_handle *h, int width, int height);

/* ---- extraction: replaces ORBextractor::operator()(image, mask, keypoints, descriptors)
 *      (src/ORBextractor.cc:1961-2084; called from Frame::ExtractORB, src/Frame.cc:468-481) ---- */
/* one host frame (8-bit gray, `stride` bytes per row); kps[cap], desc[cap*32]; *n = count */
orbx_status orbx_extract(orbx_handle *h, const uint8_t *img, int width, int height, int stride,
                         orbx_keypoint *kps, uint8_t *desc, int cap, int *n);
/* nframes contiguous host frames (frame_stride bytes apart); outputs [nframes][cap] */
orbx_status orbx_extract_batch(orbx_handle *h, int nframes, const uint8_t *imgs, int width,
                               int height, int stride, int64_t frame_stride, orbx_keypoint *kps,
                               uint8_t *desc, int32_t *counts, int cap);
/* same, all pointers are DEVICE pointers; asynchronous on the handle's stream.  d_status[nframes]
 * receives a per-frame orbx_status (OK / CAPACITY).  No host synchronisation. */
orbx_status orbx_extract_batch_device(orbx_handle *h, int nframes, const uint8_t *d_imgs, int width,
                                      int height, int stride, int64_t frame_stride,
                                      orbx_keypoint *d_kps, uint8_t *d_desc, int32_t *d_counts,
                                      int32_t *d_status, int cap);

/* Pixel format of the frames given to the three extract entry points (default ORBX_FMT_GRAY8).  The colour formats
 * fuse the cv::cvtColor(CV_RGB2GRAY / CV_BGR2GRAY / CV_RGBA2GRAY / CV_BGRA2GRAY) that Tracking::GrabImage* runs before
 * the extractor (src/Tracking.cc:245-271, 302-320, 372-385; mbRGB selects the RGB variants) into level 0:
 * gray = (R*4899 + G*9617 + B*1868 + 8192) >> 14 (OpenCV 3.2, 8-bit).  `stride` stays in bytes, `width` in pixels. */
enum { ORBX_FMT_GRAY8 = 0, ORBX_FMT_RGB8 = 1, ORBX_FMT_BGR8 = 2, ORBX_FMT_RGBA8 = 3, ORBX_FMT_BGRA8 = 4 };
orbx_status orbx_set_input_format(orbx_handle *h, int pixel_format);

/* EuRoC rectification (reference Examples/Stereo/stereo_euroc.cc:126-194): cv::remap(im, imRect, M1, M2, cv::INTER_LINEAR)
 * with the CV_32F maps of cv::initUndistortRectifyMap is fused into level 0; the extract entry points then take the RAW
 * image.  map_x / map_y: width x height floats, row major (M1l / M2l); raw and rectified images have the same size, as in
 * the reference's EuRoC driver.  OpenCV 3.2 fixed-point arithmetic (5-bit fractions, 2^15 weights, border value 0): parity
 * unpinned.  NULL, NULL switches it off.  8-bit gray input only. */
orbx_status orbx_set_rectification(orbx_handle *h, const float *map_x, const float *map_y, int width, int height);

/* ---- pyramid access: replaces the public member `mvImagePyramid` (include/ORBextractor.h:185),
 *      read by Frame::ComputeStereoMatches (src/Frame.cc:910,1040,1072,1079).  Valid until the next
 *      extract on this handle ("pyramid is overwritten every frame", include/ORBextractor.h:30-35). -- */
orbx_status orbx_pyramid_level_info(orbx_handle *h, int level, int *width, int *height, int *pitch);
/* device view of (frame, level): padded image, `pitch` bytes per row */
orbx_status orbx_pyramid_level_device(orbx_handle *h, int frame, int level, const uint8_t **d_ptr);
/* copies the padded level into dst (dst_stride >= width) */
orbx_status orbx_pyramid_level_copy(orbx_handle *h, int frame, int level, uint8_t *dst, int dst_stride);

/* ---- matching: ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:2073-2093) evaluated for every
 *      (query, train) pair with the best / second-best bookkeeping of the search loops
 *      (e.g. src/ORBmatcher.cc:627-640).  Device pointers, asynchronous on the handle's stream.
 *      npairs independent problems: pair p uses query rows d_q + p*q_stride (nq[p] rows) and train
 *      rows d_t + p*t_stride (nt[p] rows); outputs [npairs][out_stride].  Contract: out_stride is the
 *      query capacity -- queries beyond it are ignored (nq[p] is clamped to out_stride on the device,
 *      nothing is written outside row p of the outputs). --------------------------------------------- */
orbx_status orbx_match_bruteforce_device(orbx_handle *h, int npairs, const uint8_t *d_q,
                                         const int32_t *d_nq, int64_t q_stride, const uint8_t *d_t,
                                         const int32_t *d_nt, int64_t t_stride, int32_t *d_best_idx,
                                         int32_t *d_best_dist, int32_t *d_second_dist, int out_stride);
/* host convenience: one problem, host buffers */
orbx_status orbx_match_bruteforce(orbx_handle *h, const uint8_t *q, int nq, const uint8_t *t, int nt,
                                  int32_t *best_idx, int32_t *best_dist, int32_t *second_dist);
/* full nq x nt distance matrix (uint16), for host-side sequential policies */
orbx_status orbx_hamming_matrix(orbx_handle *h, const uint8_t *q, int nq, const uint8_t *t, int nt,
                                uint16_t *dist);

/* ---- matcher policies on the path ------------------------- */
/* Frame::AssignFeaturesToGrid + PosInGrid (src/Frame.cc:432-460, 729-745): 64 x 48 buckets over keypoints that the
 * caller keeps alive; bounds = mnMinX, mnMaxX, mnMinY, mnMaxY.  Host-side utility (the reference's own form); every policy
 * entry point below builds and queries the grid ON THE DEVICE instead (orbx_grid_build_device / orbx_gated_candidates). */
typedef struct orbx_grid orbx_grid;
orbx_grid *orbx_grid_create(const orbx_keypoint *kps, int n, float min_x, float max_x, float min_y, float max_y);
void orbx_grid_destroy(orbx_grid *g);
/* Frame::GetFeaturesInArea (src/Frame.cc:633-717); returns the number of hits (may exceed cap), < 0 on error */
int orbx_grid_query(const orbx_grid *g, float x, float y, float r, int min_level, int max_level, int32_t *out, int cap);
/* ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:2026-2068) */
void orbx_three_maxima(const int32_t *sizes, int L, int *ind1, int *ind2, int *ind3);
/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:570-712; caller src/Tracking.cc:950-962).
 * F1 = (k1, d1, n1), F2 = (k2, d2, n2), bounds4 = {mnMinX, mnMaxX, mnMinY, mnMaxY} of F2,
 * prev_matched[2*n1] = vbPrevMatched (in/out), matches12[n1] = vnMatches12 (out).  Host buffers. */
orbx_status orbx_search_for_initialization(orbx_handle *h, const orbx_keypoint *k1, const uint8_t *d1, int n1,
                                           const orbx_keypoint *k2, const uint8_t *d2, int n2, const float *bounds4,
                                           float *prev_matched, int window, float nnratio, int check_orientation,
                                           int32_t *matches12, int *nmatches);
/* Frame::ComputeStereoMatches (src/Frame.cc:880-1176).  hl / hr are the left / right extractors whose pyramids of
 * (frame_left, frame_right) are still resident (mvImagePyramid is read at :910,1040,1072,1079); keypoints and
 * descriptors are host buffers; u_right / depth = mvuRight / mvDepth.  The reference's unchecked row index (F6)
 * is clamped. */
orbx_status orbx_stereo_match(orbx_handle *hl, orbx_handle *hr, int frame_left, int frame_right,
                              const orbx_keypoint *kl, const uint8_t *dl, int nl, const orbx_keypoint *kr,
                              const uint8_t *dr, int nr, float mb, float mbf, float *u_right, float *depth,
                              int *nmatches);

/* Batched, device-resident form: pair p = frame p of the last orbx_extract_batch_device of `hl` (left eyes) and of `hr`
 * (right eyes); d_k* / d_d* / d_n* are the device buffers those calls filled, `cap` records apart.  Outputs (device):
 * u_right / depth [npairs][cap], nmatches [npairs].  The median cut of :1160-1175 runs on the device too.  Asynchronous on
 * hl's stream.  hl == hr is allowed: the two eyes then went through ONE orbx_extract_batch_device call of 2 * npairs images
 * on that handle, left images first (frames 0 .. npairs-1), right images after them (the reference's two extractors carry the
 * same parameters in stereo, src/Tracking.cc; one batch halves the launches per stereo frame). */
orbx_status orbx_stereo_match_batch_device(orbx_handle *hl, orbx_handle *hr, int npairs, const orbx_keypoint *d_kl,
                                           const uint8_t *d_dl, const int32_t *d_nl, const orbx_keypoint *d_kr,
                                           const uint8_t *d_dr, const int32_t *d_nr, int cap, float mb, float mbf,
                                           float *d_u_right, float *d_depth, int32_t *d_nmatches);

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) (src/ORBmatcher.cc:1702-1871;
 * caller Tracking::TrackWithMotionModel, src/Tracking.cc:1430,1445).  The Frame / MapPoint fields the policy reads are
 * passed as arrays.  CurrentFrame.mvpMapPoints is all-NULL on entry (src/Tracking.cc:1420 fills it); the result is
 * matched_last[i2] = index of the last-frame feature whose MapPoint was assigned to current feature i2, or -1. */
typedef struct orbx_frame_view {
    const orbx_keypoint *keys_un;   /* mvKeysUn */
    const uint8_t *desc;            /* mDescriptors, n x 32 */
    const float *u_right;           /* mvuRight (-1 for monocular points) */
    int32_t n;
    float Tcw[16];                  /* mTcw, row major 4x4 */
    float fx, fy, cx, cy;
    float min_x, max_x, min_y, max_y; /* mnMinX .. mnMaxY */
    float mb, mbf;
} orbx_frame_view;
typedef struct orbx_last_frame_view {
    const orbx_keypoint *keys_un;   /* LastFrame.mvKeysUn (octave, angle) */
    int32_t n;
    const uint8_t *has_map_point;   /* mvpMapPoints[i] != NULL && !mvbOutlier[i] */
    const float *world_pos;         /* pMP->GetWorldPos(), 3 floats per feature */
    const uint8_t *mp_desc;         /* pMP->GetDescriptor(), 32 bytes per feature */
    const int32_t *observations;    /* pMP->Observations() */
    float Tcw[16];                  /* LastFrame.mTcw */
} orbx_last_frame_view;
orbx_status orbx_search_by_projection_frame(orbx_handle *h, const orbx_frame_view *cur, const orbx_last_frame_view *last,
                                            float th, int mono, int check_orientation, int32_t *matched_last,
                                            int *nmatches);

/* ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th) (src/ORBmatcher.cc:69-184; caller
 * Tracking::SearchLocalPoints, src/Tracking.cc:1953).  MapPoint fields (written by Frame::isInFrustum in the reference)
 * as arrays; frame_observations[idx] = Observations() of the MapPoint already attached to feature idx, -1 if none.
 * assigned[idx] = index of the MapPoint newly attached to feature idx, or -1.  Only frame->keys_un/desc/u_right/n and
 * the bounds of `frame` are read. */
typedef struct orbx_mappoint_view {
    int32_t n;
    const uint8_t *in_view;       /* mbTrackInView && !isBad() */
    const float *proj;            /* mTrackProjX, mTrackProjY, mTrackProjXR: 3 floats per point */
    const int32_t *level;         /* mnTrackScaleLevel */
    const float *view_cos;        /* mTrackViewCos */
    const uint8_t *desc;          /* GetDescriptor(), 32 bytes per point */
    const int32_t *observations;  /* Observations() */
} orbx_mappoint_view;
orbx_status orbx_search_by_projection_mappoints(orbx_handle *h, const orbx_frame_view *frame,
                                                const int32_t *frame_observations, const orbx_mappoint_view *mps,
                                                float th, float nnratio, int32_t *assigned, int *nmatches);

/* ---- BoW-guided policies.  Host code keeps the pointer chasing (KeyFrame / MapPoint /
 * DBoW2 containers) and hands the fields the policies read as arrays; the Hamming distances come from the GPU, the
 * order-dependent selection runs on the host exactly as the reference does. */
/* DBoW2::FeatureVector = std::map<NodeId, std::vector<unsigned int>> (Thirdparty/DBoW2/DBoW2/FeatureVector.h), flattened
 * in map order: node ids strictly ascending, n_nodes + 1 offsets into `index`, feature indices in each vector's order. */
typedef struct orbx_featvec_view {
    int32_t n_nodes;
    const uint32_t *node_id;
    const int32_t *begin;
    const uint32_t *index;
} orbx_featvec_view;
typedef struct orbx_keyframe_view {
    const orbx_keypoint *keys_un;    /* mvKeysUn */
    const uint8_t *desc;             /* mDescriptors, n x 32 */
    int32_t n;
    const uint8_t *has_map_point;    /* GetMapPoint(i) != NULL && !isBad() */
    const float *u_right;            /* mvuRight (SearchForTriangulation only) */
    orbx_featvec_view feat_vec;      /* mFeatVec */
    const float *scale_factors;      /* mvScaleFactors, per octave (SearchForTriangulation: second keyframe) */
    const float *level_sigma2;       /* mvLevelSigma2, per octave (SearchForTriangulation: second keyframe) */
} orbx_keyframe_view;
/* ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches) (src/ORBmatcher.cc:248-410;
 * callers Tracking::TrackReferenceKeyFrame src/Tracking.cc:1281, Relocalization :2115).  f_keys = F.mvKeys (angle only).
 * matched_kf[i] = index of the keyframe feature whose MapPoint is vpMapPointMatches[i], -1 = NULL. */
orbx_status orbx_search_by_bow_keyframe_frame(orbx_handle *h, const orbx_keyframe_view *kf, const orbx_keypoint *f_keys,
                                              const uint8_t *f_desc, int nf, const orbx_featvec_view *f_feat_vec,
                                              float nnratio, int check_orientation, int32_t *matched_kf, int *nmatches);
/* ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12) (:722-866; caller
 * LoopClosing::ComputeSim3).  matches12[i1] = index of the KF2 feature whose MapPoint is vpMatches12[i1], -1 = NULL. */
orbx_status orbx_search_by_bow_keyframes(orbx_handle *h, const orbx_keyframe_view *kf1, const orbx_keyframe_view *kf2,
                                         float nnratio, int check_orientation, int32_t *matches12, int *nmatches);
/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) (:879-1087, CheckDistEpipolarLine
 * :206-233; caller LocalMapping::CreateNewMapPoints).  F12: row-major 3x3; (ex, ey): epipole of KF1's centre in KF2
 * (:892-898, computed by the caller from its pose matrices).  vMatchedPairs = (i1, matches12[i1]) for ascending i1 with
 * matches12[i1] >= 0.  The handle's fp_mode selects the contraction of the float expressions. */
orbx_status orbx_search_for_triangulation(orbx_handle *h, const orbx_keyframe_view *kf1, const orbx_keyframe_view *kf2,
                                          const float *F12, float ex, float ey, int only_stereo, int check_orientation,
                                          int32_t *matches12, int *nmatches);
/* The loop of LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:375-430) calls SearchForTriangulation for the current keyframe
 * against each covisible neighbour and creates MapPoints for the matches before it goes on to the next neighbour.  The Hamming
 * distances depend on descriptors and feature vectors only: _create computes them for ALL neighbours in one device round trip;
 * _select runs the reference's selection for neighbour k on the host with the has_map_point flags the views carry WHEN IT IS
 * CALLED (same keyframes, same feature counts as at _create): the matches of K single calls for one ~60 us round trip. */
typedef struct orbx_triangulation_batch orbx_triangulation_batch;
orbx_status orbx_triangulation_batch_create(orbx_handle *h, const orbx_keyframe_view *kf1, int nproblems,
                                            const orbx_keyframe_view *const *kf2, orbx_triangulation_batch **out);
orbx_status orbx_triangulation_batch_select(const orbx_triangulation_batch *b, int k, const orbx_keyframe_view *kf1,
                                            const orbx_keyframe_view *kf2, const float *F12, float epipole_x, float epipole_y,
                                            int only_stereo, int check_orientation, int32_t *matches12, int *nmatches);
void orbx_triangulation_batch_destroy(orbx_triangulation_batch *b);

/* ---- projection-guided back-end policies.  The pose algebra in front of them (cv::Mat products, cv::norm,
 * MapPoint::PredictScale: OpenCV / libm code) stays in the maintainer's shim, which IS the reference's code; the entry
 * points start where the reference holds, for every MapPoint, "passed every geometric test", the projection, the
 * predicted level and the representative descriptor. */
typedef struct orbx_projected_points {
    int32_t n;
    const uint8_t *valid;     /* the point reaches `const float radius = th * ...` in the reference */
    const float *uv;          /* projection (u, v), 2 floats per point */
    const float *u_right;     /* ur = u - bf * invz (orbx_fuse only) */
    const int32_t *level;     /* nPredictedLevel */
    const uint8_t *desc;      /* pMP->GetDescriptor(), 32 bytes per point */
    const float *angle;       /* pKF->mvKeysUn[i].angle (orbx_search_by_projection_keyframe with check_orientation) */
} orbx_projected_points;
/* Every projection-guided entry point below gates its candidates on the device with 16-bit feature indices: a target with more
 * than 65535 features returns ORBX_UNSUPPORTED (ORB-SLAM2 frames carry 1000-4000). */
typedef struct orbx_target_view {   /* the KeyFrame / Frame whose features are searched */
    const orbx_keypoint *keys_un;    /* mvKeysUn */
    const uint8_t *desc;             /* mDescriptors */
    const float *u_right;            /* mvuRight (orbx_fuse only) */
    int32_t n;
    float min_x, max_x, min_y, max_y; /* mnMinX .. mnMaxY (grid of GetFeaturesInArea) */
    const float *scale_factors;      /* mvScaleFactors */
    const float *inv_level_sigma2;   /* mvInvLevelSigma2 (orbx_fuse only) */
} orbx_target_view;
/* ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, th) (src/ORBmatcher.cc:1100-1280), lines
 * :1168-1245: best_idx[i] = keyframe feature the point is fused into, -1 = none.  The caller then runs the Replace /
 * AddObservation / AddMapPoint branch (:1248-1275) over i in order; it does not feed back into the selection. */
orbx_status orbx_fuse(orbx_handle *h, const orbx_target_view *kf, const orbx_projected_points *pts, float th,
                      int32_t *best_idx, int *nfused);
/* ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, vpPoints, th, vpReplacePoint) (:1282-1430), lines :1345-1400 */
orbx_status orbx_fuse_sim3(orbx_handle *h, const orbx_target_view *kf, const orbx_projected_points *pts, float th,
                           int32_t *best_idx, int *nfused);
/* Batched forms: nproblems (target, point set) pairs through ONE upload, ONE grid-build + gate launch pair and ONE download
 * instead of a ~60 us synchronous round trip each.  The reference calls Fuse once per neighbour keyframe in a loop --
 * LocalMapping::SearchInNeighbors (src/LocalMapping.cc:750-768), LoopClosing::SearchAndFuse -- and that loop is the batch.
 * best_idx[k][i], nfused[k] are exactly what orbx_fuse / orbx_fuse_sim3 return for (kfs[k], pts[k]); all targets must share the
 * image bounds (min_x .. max_y: Frame's static mnMinX .. mnMaxY).  Earlier iterations of the reference's loop reach later ones
 * only through pMP->isBad() / IsInKeyFrame(), which the caller re-checks while it applies the results in order. */
orbx_status orbx_fuse_batch(orbx_handle *h, int nproblems, const orbx_target_view *const *kfs,
                            const orbx_projected_points *const *pts, float th, int32_t *const *best_idx, int *nfused);
orbx_status orbx_fuse_sim3_batch(orbx_handle *h, int nproblems, const orbx_target_view *const *kfs,
                                 const orbx_projected_points *const *pts, float th, int32_t *const *best_idx, int *nfused);
/* ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, vpPoints, vpMatched, int th) (:415-560), lines :498-556.
 * matched[idx] = vpMatched[idx] != NULL, updated exactly as the reference updates vpMatched (point order matters);
 * best_idx[i] = feature that received point i, -1 = none. */
orbx_status orbx_search_by_projection_sim3(orbx_handle *h, const orbx_target_view *kf, const orbx_projected_points *pts,
                                           int th, uint8_t *matched, int32_t *best_idx, int *nmatches);
/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (:1433-1690) from the two projection loops on.
 * pts1_in_2: one entry per KF1 feature (valid = has a good MapPoint, !vbAlreadyMatched1, all geometric tests), projected
 * into KF2; pts2_in_1 the converse.  matches12[i1] = KF2 feature whose MapPoint becomes vpMatches12[i1], -1 otherwise. */
orbx_status orbx_search_by_sim3(orbx_handle *h, const orbx_target_view *kf1, const orbx_target_view *kf2,
                                const orbx_projected_points *pts1_in_2, const orbx_projected_points *pts2_in_1, float th,
                                int32_t *matches12, int *nfound);
/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist) (:1873-2020; caller
 * Tracking::Relocalization).  pts = pKF's MapPoints; cur_has_map_point[i2] = CurrentFrame.mvpMapPoints[i2] != NULL
 * (in/out); matched_point[i2] = index of the point attached to feature i2, -1. */
orbx_status orbx_search_by_projection_keyframe(orbx_handle *h, const orbx_target_view *cur, const orbx_projected_points *pts,
                                               float th, int orb_dist, int check_orientation, uint8_t *cur_has_map_point,
                                               int32_t *matched_point, int *nmatches);

/* ---- DBoW2 transform: Frame::ComputeBoW (src/Frame.cc:750-765) =
 * TemplatedVocabulary<FORB>::transform(features, BowVector&, FeatureVector&, 4)
 * (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1136-1216, 1240-1285; FORB::distance FORB.cpp:81-101). */
typedef struct orbx_vocabulary_view {   /* m_nodes flattened; node 0 is the root */
    int32_t n_nodes, k, L;
    int32_t weighting;               /* DBoW2::WeightingType: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY */
    int32_t scoring;                 /* DBoW2::ScoringType: 0 L1_NORM, 1 L2_NORM, 2 CHI_SQUARE, 3 KL, 4 BHATTACHARYYA, 5 DOT_PRODUCT */
    const int32_t *child_begin;      /* n_nodes + 1 offsets into child_ids: Node::children in vector order */
    const uint32_t *child_ids;       /* every child id is greater than its parent's (true of every DBoW2-built tree) */
    const uint8_t *desc;             /* Node::descriptor, n_nodes x 32 (the root's row is not read) */
    const double *weight;            /* Node::weight */
    const uint32_t *word_id;         /* Node::word_id (meaningful for leaves) */
} orbx_vocabulary_view;
typedef struct orbx_vocabulary orbx_vocabulary;
/* copies the tree to the handle's device; the vocabulary can then be used with any handle on that device */
orbx_status orbx_vocabulary_create(orbx_handle *h, const orbx_vocabulary_view *view, orbx_vocabulary **out);
void orbx_vocabulary_destroy(orbx_vocabulary *voc);
/* per descriptor: transform(feature, word_id, weight, &nid, levelsup).  Host buffers. */
orbx_status orbx_bow_transform(orbx_handle *h, const orbx_vocabulary *voc, const uint8_t *desc, int n, int levelsup,
                               uint32_t *word_id, double *weight, uint32_t *node_id);
/* batched, device buffers (descriptors as orbx_extract_batch_device leaves them): d_leaf_node[f][i] = leaf reached by
 * descriptor i of frame f (word id / weight = tables of the view), d_node_id[f][i] = node at depth L - levelsup */
orbx_status orbx_bow_transform_device(orbx_handle *h, const orbx_vocabulary *voc, int nframes, const uint8_t *d_desc,
                                      const int32_t *d_counts, int64_t desc_frame_stride, int max_n, int levelsup,
                                      uint32_t *d_leaf_node, uint32_t *d_node_id, int out_stride);
/* the BowVector and FeatureVector transform() builds from the per-descriptor results, flattened in map order (bow_word
 * ascending with bow_value; fv_node ascending, fv_begin[n_fv_nodes + 1], fv_index = an orbx_featvec_view).  Capacities: n. */
orbx_status orbx_bow_vectors(const orbx_vocabulary *voc, const uint32_t *word_id, const double *weight,
                             const uint32_t *node_id, int n, uint32_t *bow_word, double *bow_value, int *n_bow,
                             uint32_t *fv_node, int32_t *fv_begin, uint32_t *fv_index, int *n_fv_nodes);

/* ---- Frame glue: Frame::UndistortKeyPoints / ComputeImageBounds (src/Frame.cc:770-865) =
 * cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK) on the keypoint coordinates.  camera4 = fx, fy, cx, cy of mK;
 * dist = mDistCoef (k1, k2, p1, p2[, k3], at most 14); dist[0] == 0 copies the keypoints unchanged (:772-776).  OpenCV 3.2
 * arithmetic (double precision, five fixed iterations): parity unpinned, like every OpenCV-owned stage.  The device form
 * works on the buffers orbx_extract_batch_device filled (records `cap` apart, counts per frame) so that the keypoints need
 * not leave the GPU; AssignFeaturesToGrid on those buffers is orbx_grid_build_device (below), so a gated match reads them
 * where the extraction left them. */
orbx_status orbx_undistort_keypoints_device(orbx_handle *h, int nframes, const orbx_keypoint *d_kps, const int32_t *d_counts,
                                            int cap, const float *camera4, const float *dist, int ndist,
                                            orbx_keypoint *d_kps_un);
orbx_status orbx_undistort_keypoints(orbx_handle *h, const orbx_keypoint *kps, int n, const float *camera4, const float *dist,
                                     int ndist, orbx_keypoint *kps_un);
/* Frame::AssignFeaturesToGrid + PosInGrid (src/Frame.cc:432-460, 729-745) on the device, for the keypoint buffers
 * orbx_extract_batch_device / orbx_undistort_keypoints_device filled (records `cap` apart, cap <= 65535): per frame
 * d_cell_begin[64 * 48 + 1] offsets (bucket c = column * 48 + row, the reference's mGrid[column][row]) into d_items[cap], the
 * feature indices of every bucket in ascending order (= push_back order).  bounds4 = mnMinX, mnMaxX, mnMinY, mnMaxY.
 * Asynchronous on the handle's stream: the keypoints do not leave the GPU between extraction and a gated match. */
orbx_status orbx_grid_build_device(orbx_handle *h, int nframes, const orbx_keypoint *d_kps, const int32_t *d_counts, int cap,
                                   const float *bounds4, int32_t *d_cell_begin, uint16_t *d_items);
/* The primitive behind every projection-guided policy: Frame::GetFeaturesInArea (src/Frame.cc:633-717, incl. the quirk
 * bCheckLevels = minLevel > 0 || maxLevel >= 0, :673) for nq queries against one target frame, fused with
 * ORBmatcher::DescriptorDistance.  Host buffers.  xyr = (x, y, r) per query (r < 0: no candidates), levels = (minLevel,
 * maxLevel) per query, qdesc = one descriptor per query.  begin[nq + 1] = offsets into items; items = the candidates of each
 * query in the reference's visiting order (column outer, row inner, bucket order), feature index | Hamming distance << 16.
 * *total = number of items (ORBX_CAPACITY if it exceeds items_cap; begin / total are valid then). */
orbx_status orbx_gated_candidates(orbx_handle *h, const orbx_keypoint *target_keys, const uint8_t *target_desc, int nt,
                                  const float *bounds4, const float *xyr, const int32_t *levels, const uint8_t *qdesc, int nq,
                                  uint32_t *begin, uint32_t *items, int items_cap, int *total);
/* mnMinX, mnMaxX, mnMinY, mnMaxY */
orbx_status orbx_image_bounds(orbx_handle *h, int cols, int rows, const float *camera4, const float *dist, int ndist,
                              float *bounds4);

/* Page-locked host memory for the host-buffer entry points: with it every upload / download of orbx_extract_batch is an
 * asynchronous DMA that overlaps the kernels of the neighbouring chunks (pageable memory works too, the runtime then
 * stages the copies and blocks the calling thread while it does).  A call with more frames than the handle's max_batch is
 * processed in chunks of max_batch frames, the upload of chunk c+1 overlapping the kernels of chunk c on ONE copy stream.
 * Keypoint / descriptor buffers allocated here (or any page-locked, device-mapped memory: hipHostMalloc, hipHostRegister
 * with the mapped flag) are not downloaded at all in such a call: the describe kernel writes its records straight into them
 * over the link, only the per-frame counts come back by copy; their contents are defined when the call returns, rows beyond
 * a frame's count are left as they were.  (640x480, 1000 features, chunks of 64: 134 k frames/s per 256-frame call, 155 k per
 * 1024-frame call from page-locked memory, 109-127 k from pageable memory.)  NULL on failure. */
void *orbx_host_alloc(size_t bytes);
void orbx_host_free(void *p);

/* ---- stream / timing plumbing ------------------------------------------------------------ */
void *orbx_get_stream(orbx_handle *h);            /* hipStream_t */
orbx_status orbx_set_stream(orbx_handle *h, void *hip_stream); /* NULL restores the private stream */
orbx_status orbx_synchronize(orbx_handle *h);
/* per-kernel HIP-event timing on the handle's stream.  mask bit k enables kernel id k. */
enum {
    ORBX_K_PYR_L0 = 0, ORBX_K_PYR_RESIZE = 1, ORBX_K_FAST = 2, ORBX_K_QUADTREE = 3, ORBX_K_ORIENT = 4,
    ORBX_K_BLUR = 5, ORBX_K_DESC = 6, ORBX_K_MATCH = 7, ORBX_K_MISC = 8, ORBX_K_COUNT = 9
};
orbx_status orbx_profile_enable(orbx_handle *h, uint32_t kernel_mask);
/* synchronises, then returns accumulated milliseconds and launch counts since the last reset */
orbx_status orbx_profile_read(orbx_handle *h, float *ms, int32_t *launches, int reset);
const char *orbx_kernel_name(int kernel_id);

/* ---- per-stage inspection of the last batch (parity tests only; synchronises) -------------- */
/* candidates of (frame, level) = vToDistributeKeys (src/ORBextractor.cc:1451-1548), border-relative
 * coordinates, in unspecified order; returns count via *n (may exceed cap => ORBX_CAPACITY) */
orbx_status orbx_debug_candidates(orbx_handle *h, int frame, int level, orbx_keypoint *out, int cap, int *n);
/* keypoints of (frame, level) after quadtree + orientation, level coordinates, list order */
orbx_status orbx_debug_level_keypoints(orbx_handle *h, int frame, int level, orbx_keypoint *out, int cap, int *n);
/* blurred padded level */
orbx_status orbx_debug_blur_copy(orbx_handle *h, int frame, int level, uint8_t *dst, int dst_stride);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */
