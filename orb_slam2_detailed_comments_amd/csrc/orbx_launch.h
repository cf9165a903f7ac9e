// orbx_launch.h -- kernel launch wrappers (defined in orbx_kernels.hip, used by orbx_api.cpp)
#pragma once
#include <hip/hip_runtime.h>
#include "orbx_device.h"

hipError_t orbx_upload_pattern();
size_t orbx_quadtree_smem(int ncap, int lds_keys);
hipError_t orbx_quadtree_prepare(size_t smem);
void orbx_launch_clear(hipStream_t s, int *a, int na, int *b, int nb, int *c, int nc);
void orbx_launch_pyr_l0(hipStream_t s, const DGeom &g, int B, const uint8_t *imgs, int W, int H, int stride,
                        long long frame_stride, uint8_t *pyr, int *status, int *cand_cursor);
void orbx_launch_pyr_l0_color(hipStream_t s, const DGeom &g, int B, const uint8_t *imgs, int W, int H, int stride,
                              long long frame_stride, uint8_t *pyr, int nch, int r_off, int b_off, int *status, int *cand_cursor);
void orbx_launch_pyr_l0_remap(hipStream_t s, const DGeom &g, int B, const uint8_t *imgs, int W, int H, int stride,
                              long long frame_stride, uint8_t *pyr, const uint2 *rect, int *status, int *cand_cursor);
void orbx_launch_pyr_resize(hipStream_t s, const DGeom &g, int B, int level, const OrbxTap *taps, uint8_t *pyr, bool narrow);
void orbx_launch_fast_rows(hipStream_t s, const DGeom &g, int B, const OrbxCell *cells, const OrbxFastGroup *groups,
                           int ngroups, const uint8_t *pyr, uint2 *cand, int *cand_cursor, int *status, int max_ch, int lcap,
                           int dbg_stop);
void orbx_launch_undistort(hipStream_t s, int B, int max_n, int cap, const double *K4, const double *k14, int identity,
                           const orbx_keypoint *kps, const int *counts, orbx_keypoint *out);
void orbx_launch_bow_transform(hipStream_t s, int B, int max_n, const int *child_begin, const uint32_t *child_ids,
                               const uint8_t *node_desc, int n_nodes, int L, const uint8_t *desc, const int *counts,
                               long long frame_stride, int levelsup, uint32_t *out_leaf, uint32_t *out_nid, int out_stride);
void orbx_launch_quadtree(hipStream_t s, const DGeom &g, int B, const uint2 *dense, const int *cand_count, uint32_t *lvl_kp, int *lvl_count,
                          int *status, uint16_t *knode_glob, int ncap, int lds_keys, int level_begin, int level_count);
void orbx_launch_blur(hipStream_t s, const DGeom &g, int B, const uint8_t *pyr, uint8_t *blur);
void orbx_launch_describe(hipStream_t s, const DGeom &g, int B, const uint8_t *pyr, const uint32_t *lvl_kp,
                          const int *lvl_count, float *lvl_angle, orbx_keypoint *kps, uint8_t *desc,
                          int *counts, int *status, int cap);
void orbx_launch_match(hipStream_t s, int npairs, int max_nq, const uint8_t *q, const int *nq, long long q_stride,
                       const uint8_t *t, const int *nt, long long t_stride, int *best_idx, int *best_dist,
                       int *second_dist, int out_stride, void *workspace, int kernel);
size_t orbx_match_workspace_bytes(int npairs, int out_stride);
void orbx_launch_grid_build(hipStream_t s, const DGrid &gp, int nframes, const orbx_keypoint *kps, const int *counts, int fixed_n,
                            int cap, int *cell_begin, uint16_t *items);
void orbx_launch_gate(hipStream_t s, const DGrid &gp, const orbx_keypoint *kps, const uint8_t *desc, const int *cell_begin,
                      const uint16_t *items, const DGateQuery *q, const uint8_t *qdesc, int nq, uint2 *span, uint32_t *cursor,
                      uint32_t *out_items, uint32_t cap, int fstride);
void orbx_launch_block_dist(hipStream_t s, const uint8_t *d1, const uint8_t *d2, const DDistRow *rows, const uint32_t *col_idx,
                            int nrows, uint16_t *out);
void orbx_launch_hamming_matrix(hipStream_t s, const uint8_t *q, int nq, const uint8_t *t, int nt, uint16_t *dist);

void orbx_launch_stereo_batch(hipStream_t s, const OrbxStereoGeom &sg, int npairs, int cap, const orbx_keypoint *kL,
                              const uint8_t *dL, const int *nL, const orbx_keypoint *kR, const uint8_t *dR, const int *nR,
                              const uint8_t *pyrL, const uint8_t *pyrR, long long pyr_bytes, float *uRight, float *depth,
                              int *sad, int *nmatches, int *row_begin, uint2 *row_items);
int orbx_stereo_items_per_pair(const OrbxStereoGeom &sg, int cap);
void orbx_launch_stereo(hipStream_t s, const OrbxStereoGeom &sg, const orbx_keypoint *kL, const uint8_t *dL, int nL,
                        const orbx_keypoint *kR, const uint8_t *dR, int nR, const uint8_t *pyrL, const uint8_t *pyrR,
                        float *uRight, float *depth, int *sad, int *row_begin, uint2 *row_items);
