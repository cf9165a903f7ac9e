/* orb_oracle.h -- CPU restatement ("oracle") of the ORB front-end hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the HIP library under
 * orb_slam2_detailed_comments_amd/) may include, link, load or call anything declared
 * here.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * only as the checker / the timed CPU baseline.
 *
 * What it restates (all file:line are relative to the reference tree):
 *   ORBextractor ctor              src/ORBextractor.cc:776-925
 *   ComputePyramid (fork: padded)  src/ORBextractor.cc:2093-2168
 *   ComputeKeyPointsOctTree        src/ORBextractor.cc:1424-1601
 *   DivideNode / DistributeOctTree src/ORBextractor.cc:963-1035, 1050-1417
 *   IC_Angle / computeOrientation  src/ORBextractor.cc:104-161, 933-952
 *   computeOrbDescriptor           src/ORBextractor.cc:177-254
 *   operator()                     src/ORBextractor.cc:1961-2084
 *   ORBmatcher::DescriptorDistance src/ORBmatcher.cc:2073-2093
 *   ORBmatcher::ComputeThreeMaxima src/ORBmatcher.cc:2026-2068
 *   ORBmatcher::SearchForInitialization src/ORBmatcher.cc:570-712
 *   Frame grid / GetFeaturesInArea src/Frame.cc:432-460, 633-745
 *   Frame::ComputeStereoMatches    src/Frame.cc:880-1176
 *
 * PARITY PINNING: the reference has no tests, fixtures or golden vectors, and the
 * arithmetic of cv::FAST / cv::resize / cv::copyMakeBorder / cv::GaussianBlur /
 * cv::fastAtan2 / cvRound lives in OpenCV (un-vendored system dependency, required
 * version 3.0+, README-tested 2.4.11 and 3.2; CMakeLists.txt:47-53, README.md:103),
 * which is absent from this image.  Those primitives are restated here from the published
 * OpenCV 3.2 algorithms (x86-64 SSE2 baseline build, no IPP).  The reference cannot be
 * built here (needs OpenCV headers+libs) => **parity unpinned** at the OpenCV boundary;
 * the only reference-held known answers (umax table, vmax/vmin, thresholds; see
 * tests/test_oracle_kat.py) are checked.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* bit-compatible with cv::KeyPoint (28 bytes) */
typedef struct orc_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;

enum { ORC_FP_GCC_FMA = 0, ORC_FP_STRICT = 1 };

typedef struct orc_extractor orc_extractor;

/* ---- extractor ---------------------------------------------------------------- */
orc_extractor *orc_create(int nfeatures, float scale_factor, int nlevels,
                          int ini_th_fast, int min_th_fast, int fp_mode);
void orc_destroy(orc_extractor *e);
/* returns number of keypoints (>=0) or <0: -1 empty image, -3 bad aspect (nIni==0),
 * -4 capacity too small.  kps/desc may be NULL (cap ignored) to only run the stages. */
int orc_extract(orc_extractor *e, const uint8_t *img, int w, int h, int stride,
                orc_keypoint *kps, uint8_t *desc, int cap);
/* tables (ctor) */
void orc_get_tables(const orc_extractor *e, float *scale, float *inv_scale, float *sigma2,
                    float *inv_sigma2, int *features_per_level, int *umax16);
/* per-stage results of the last orc_extract */
int orc_level_dims(const orc_extractor *e, int level, int *w, int *h); /* padded dims */
const uint8_t *orc_level_image(const orc_extractor *e, int level);      /* padded, step=w */
const uint8_t *orc_level_blur(const orc_extractor *e, int level);       /* NULL if level had 0 kps */
int orc_level_candidates(const orc_extractor *e, int level, orc_keypoint *out, int cap);
int orc_level_keypoints(const orc_extractor *e, int level, orc_keypoint *out, int cap);
/* per-stage wall time (seconds) accumulated since create: pyramid, fast, quadtree, orient, blur, desc */
void orc_get_stage_times(const orc_extractor *e, double *t6);

/* ---- primitives (unit-testable) -------------------------------------------------- */
void orc_border_reflect101(const uint8_t *src, int w, int h, int sstride,
                           uint8_t *dst, int dstride, int border);
void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride,
                          uint8_t *dst, int dw, int dh, int dstride);
/* cv::FAST(img, kps, threshold, nonmax=true), TYPE_9_16; returns count (may exceed cap) */
int orc_fast9_16(const uint8_t *img, int w, int h, int stride, int threshold,
                 orc_keypoint *out, int cap);
void orc_gaussian_blur7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride);
void orc_cvt_gray(const uint8_t *src, int w, int h, int sstride, int nch, int rgb_order, uint8_t *dst, int dstride);
float orc_fast_atan2(float y, float x);
int orc_cv_round_f(float v);
/* quadtree on its own: keys in (x,y,response); returns count, writes selected input indices
 * in output (list) order */
int orc_distribute_octtree(const orc_keypoint *keys, int nkeys, int minX, int maxX, int minY,
                           int maxY, int N, int *out_idx, int cap);
float orc_ic_angle(const uint8_t *img, int stride, int x, int y);
void orc_descriptor(const uint8_t *blur, int stride, int x, int y, float angle_deg, int fp_mode,
                    uint8_t *desc32);

/* ---- matcher ------------------------------------------------------------------- */
int orc_descriptor_distance(const uint8_t *a, const uint8_t *b);
/* brute force: for every query the best (lowest index on ties) / second-best distance */
void orc_match_bruteforce(const uint8_t *q, int nq, const uint8_t *t, int nt,
                          int *best_idx, int *best_dist, int *second_dist);
void orc_three_maxima(const int *hist_sizes, int L, int *ind1, int *ind2, int *ind3);

/* Frame grid (64 x 48 buckets): AssignFeaturesToGrid / GetFeaturesInArea */
typedef struct orc_grid orc_grid;
orc_grid *orc_grid_build(const orc_keypoint *kps, int n, float minx, float maxx, float miny, float maxy);
void orc_grid_free(orc_grid *g);
int orc_grid_query(const orc_grid *g, float x, float y, float r, int minLevel, int maxLevel, int *out, int cap);
/* ORBmatcher::SearchForInitialization; prev_matched[2*n1] in/out, matches12[n1] out; returns nmatches */
int orc_search_for_initialization(const orc_keypoint *k1, const uint8_t *d1, int n1, const orc_keypoint *k2,
                                  const uint8_t *d2, int n2, float minx, float maxx, float miny, float maxy,
                                  float *prev_matched, int window, float nnratio, int check_ori, int *matches12);
/* Frame::ComputeStereoMatches; returns the number of stereo matches kept */
int orc_stereo_matches(const orc_keypoint *kL, const uint8_t *dL, int nL, const orc_keypoint *kR, const uint8_t *dR,
                       int nR, int nlevels, const float *scale, const float *inv_scale, const uint8_t *const *pyrL,
                       const uint8_t *const *pyrR, const int *pw, const int *ph, float mb, float mbf, float *uRight,
                       float *depth);
/* ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono); matched_last[nc] out; returns nmatches */
int orc_search_by_projection_ff(const orc_keypoint *kc, const uint8_t *dc, const float *u_right, int nc,
                                const float *Tcw, float fx, float fy, float cx, float cy, float minx, float maxx,
                                float miny, float maxy, float mb, float mbf, const float *scale_factors,
                                const orc_keypoint *kl, int nl, const uint8_t *has_mp, const float *xw,
                                const uint8_t *mpdesc, const int *obs, const float *Tlw, float th, int bMono,
                                int check_ori, int fp_mode, int *matched_last);
/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (local-map tracking); assigned[nf] out */
int orc_search_by_projection_mp(const orc_keypoint *kf, const uint8_t *df, const float *u_right, const int *frame_obs,
                                int nf, float minx, float maxx, float miny, float maxy, const float *scale_factors,
                                int nmp, const uint8_t *in_view, const float *proj, const int *level,
                                const float *view_cos, const uint8_t *mpdesc, const int *mp_obs, float th, float nnratio,
                                int *assigned);

/* ---- BoW-guided policies.  A DBoW2::FeatureVector is passed flattened: nn node ids (ascending), nn + 1 offsets,
 * feature indices in each node's own order. */
/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) src/ORBmatcher.cc:248-410; matched_kf[nf] out */
int orc_search_by_bow_kf_frame(const orc_keypoint *kkf, const uint8_t *dkf, int nkf, const uint8_t *kf_has_mp,
                               int nn_kf, const uint32_t *node_kf, const int *beg_kf, const uint32_t *idx_kf,
                               const orc_keypoint *kf_, const uint8_t *df, int nf, int nn_f, const uint32_t *node_f,
                               const int *beg_f, const uint32_t *idx_f, float nnratio, int check_ori, int *matched_kf);
/* ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, ...) :722-866; matches12[n1] out */
int orc_search_by_bow_kf_kf(const orc_keypoint *k1, const uint8_t *d1, int n1, const uint8_t *has_mp1, int nn1,
                            const uint32_t *node1, const int *beg1, const uint32_t *idx1v, const orc_keypoint *k2,
                            const uint8_t *d2, int n2, const uint8_t *has_mp2, int nn2, const uint32_t *node2,
                            const int *beg2, const uint32_t *idx2v, float nnratio, int check_ori, int *matches12);
/* ORBmatcher::SearchForTriangulation :879-1087 with CheckDistEpipolarLine :206-233; matches12[n1] out */
int orc_search_for_triangulation(const orc_keypoint *k1, const uint8_t *d1, int n1, const uint8_t *has_mp1,
                                 const float *ur1, int nn1, const uint32_t *node1, const int *beg1,
                                 const uint32_t *idx1v, const orc_keypoint *k2, const uint8_t *d2, int n2,
                                 const uint8_t *has_mp2, const float *ur2, int nn2, const uint32_t *node2,
                                 const int *beg2, const uint32_t *idx2v, const float *F12, float ex, float ey,
                                 const float *scale_factors2, const float *level_sigma2_2, int only_stereo,
                                 int check_ori, int fp_mode, int *matches12);

/* ---- projection-guided back-end policies, from the point where the reference holds (valid, u, v[, ur], predicted level,
 * descriptor) of every MapPoint (the cv::Mat pose algebra in front stays in the reference's translation unit) */
int orc_fuse(const orc_keypoint *kk, const uint8_t *dk, const float *ur_k, int nk, float minx, float maxx, float miny,
             float maxy, const float *scale_factors, const float *inv_level_sigma2, int np, const uint8_t *valid,
             const float *uv, const float *ur, const int *level, const uint8_t *desc, float th, int fp_mode,
             int *best_idx);
int orc_fuse_sim3(const orc_keypoint *kk, const uint8_t *dk, int nk, float minx, float maxx, float miny, float maxy,
                  const float *scale_factors, int np, const uint8_t *valid, const float *uv, const int *level,
                  const uint8_t *desc, float th, int *best_idx);
int orc_search_by_projection_sim3(const orc_keypoint *kk, const uint8_t *dk, int nk, float minx, float maxx, float miny,
                                  float maxy, const float *scale_factors, int np, const uint8_t *valid, const float *uv,
                                  const int *level, const uint8_t *desc, int th, uint8_t *matched, int *best_idx);
int orc_search_by_sim3(const orc_keypoint *k1, const uint8_t *d1, int n1, const float *bounds1, const float *sf1,
                       const orc_keypoint *k2, const uint8_t *d2, int n2, const float *bounds2, const float *sf2,
                       const uint8_t *valid1, const float *uv1in2, const int *level1in2, const uint8_t *mpdesc1,
                       const uint8_t *valid2, const float *uv2in1, const int *level2in1, const uint8_t *mpdesc2, float th,
                       int *matches12);
int orc_search_by_projection_kf(const orc_keypoint *kc, const uint8_t *dc, int nc, float minx, float maxx, float miny,
                                float maxy, const float *scale_factors, int np, const uint8_t *valid, const float *uv,
                                const int *level, const uint8_t *desc, const float *kf_angle, float th, int orb_dist,
                                int check_ori, uint8_t *cur_has_mp, int *matched_point);

/* ---- DBoW2 TemplatedVocabulary<FORB>::transform on a flattened tree */
void orc_bow_transform(int n_nodes, int L, const int *child_begin, const uint32_t *child_ids, const uint8_t *node_desc,
                       const double *node_weight, const uint32_t *node_word, const uint8_t *desc, int n, int levelsup,
                       uint32_t *word_id, double *weight, uint32_t *node_id);
int orc_bow_vectors(int weighting, int scoring, const uint32_t *word_id, const double *weight, const uint32_t *node_id,
                    int n, uint32_t *bow_word, double *bow_value, int *n_bow, uint32_t *fv_node, int *fv_begin,
                    uint32_t *fv_index, int *n_fv_nodes);

/* ---- Frame::UndistortKeyPoints / ComputeImageBounds: cv::undistortPoints(K, D, R = I, P = K), OpenCV 3.2 (parity unpinned) */
void orc_undistort_points(const float *pts, int n, float fx, float fy, float cx, float cy, const float *dist, int ndist,
                          float *out);
void orc_undistort_keypoints(const orc_keypoint *k, int n, float fx, float fy, float cx, float cy, const float *dist,
                             int ndist, orc_keypoint *out);
void orc_image_bounds(int cols, int rows, float fx, float fy, float cx, float cy, const float *dist, int ndist, float *b4);

/* ---- cv::remap(INTER_LINEAR, CV_32FC1 maps, BORDER_CONSTANT 0) on 8-bit images: EuRoC rectification (parity unpinned) */
void orc_remap_linear_u8(const uint8_t *src, int sw, int sh, int sstride, const float *mapx, const float *mapy, int dw,
                         int dh, uint8_t *dst, int dstride);

#ifdef __cplusplus
}
#endif
#endif
