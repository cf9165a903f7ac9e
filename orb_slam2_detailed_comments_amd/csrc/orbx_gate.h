// orbx_gate.h -- device-side candidate gating shared by orbx_api.cpp and orbx_policies.cpp (internal, not part of the ABI).
//
// Every projection-guided ORBmatcher policy asks, per query, Frame::GetFeaturesInArea (reference src/Frame.cc:633-717) and then
// evaluates DescriptorDistance for the returned candidates in that order.  OrbxGateLists is exactly that, for all queries of a
// call at once, computed on the GPU (k_grid_build + k_gate): per query the candidates in the reference's visiting order, each
// as feature index (low 16 bits) | Hamming distance (high 16 bits).  The BoW-guided policies ask for the distances between the
// features under the same vocabulary node instead: orbx_block_distances (k_block_dist).
#pragma once
#include <cstdint>
#include <vector>
#include "orbx_device.h"

struct OrbxGateLists {
    std::vector<uint2> span;       // per query: (offset into items, count) -- queries are stored in the order their waves reserved
    std::vector<uint32_t> items;   // idx | dist << 16
    int count(int q) const { return (int)span[(size_t)q].y; }
    const uint32_t *list(int q) const { return items.data() + span[(size_t)q].x; }
    static int idx(uint32_t v) { return (int)(v & 0xffffu); }
    static int dist(uint32_t v) { return (int)(v >> 16); }
};
struct orbx_handle;
struct OrbxGateTarget { const orbx_keypoint *keys; const uint8_t *desc; int n; };   // one target (keyframe / frame) of a batched call
// K targets sharing the image bounds; q[i].frame names the target query i searches; lists come back per query as below.
// nqdesc >= 0: qdesc holds nqdesc rows and every query names its row (q[i].desc); -1: one row per query, in query order
orbx_status orbx_gate_lists_batch(orbx_handle *h, const OrbxGateTarget *tg, int K, float min_x, float max_x, float min_y, float max_y,
                                  const DGateQuery *q, const uint8_t *qdesc, int nq, OrbxGateLists &out, int nqdesc = -1);
// target = (keys, desc, nt) with the image bounds of its grid; queries q[nq] (r < 0: switched off) with descriptors qdesc[nq][32]
orbx_status orbx_gate_lists(orbx_handle *h, const orbx_keypoint *tkeys, const uint8_t *tdesc, int nt, float min_x, float max_x,
                            float min_y, float max_y, const DGateQuery *q, const uint8_t *qdesc, int nq, OrbxGateLists &out);
// rows[r]: descriptor rows[r].q of d1 against col_idx[col_begin .. +ncol) of d2, distances at out[out_off ..); total = out size
orbx_status orbx_block_distances(orbx_handle *h, const uint8_t *d1, int n1, const uint8_t *d2, int n2,
                                 const std::vector<DDistRow> &rows, const std::vector<uint32_t> &col_idx, size_t total,
                                 std::vector<uint16_t> &out);
