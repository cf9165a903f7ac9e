// orbx_policies.cpp -- the BoW-guided ORBmatcher policies behind the C ABI:
//   SearchByBoW(KeyFrame*, Frame&)        src/ORBmatcher.cc:248-410   (relocalisation / TrackReferenceKeyFrame)
//   SearchByBoW(KeyFrame*, KeyFrame*)     src/ORBmatcher.cc:722-866   (loop-closure candidates)
//   SearchForTriangulation                src/ORBmatcher.cc:879-1087  (LocalMapping::CreateNewMapPoints)
// Host: the merge walk over the two FeatureVectors (which pairs can meet: the features under a common vocabulary node).
// GPU: the Hamming distances of exactly those pairs, one block per common node (k_block_dist through orbx_block_distances;
// round 1 computed the dense n1 x n2 matrix and copied it to the host).
// Host: the order-dependent selection (a feature of the second set can only be taken once, in the order the reference visits
// them), exactly as the reference runs it.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "orbx_internal.h"
#include "orbx_gate.h"

extern "C" void orbx_three_maxima(const int32_t *sizes, int L, int *ind1, int *ind2, int *ind3);
orbx_status orbx_fail(orbx_status s, const std::string &msg);   // orbx_api.cpp: records orbx_last_error()
int orbx_handle_fp_mode(const orbx_handle *h);

namespace {
const int HISTO = 30, TH_LOW_ = 50;

struct RotHist {
    std::vector<int> bins[HISTO];
    void push(float a1, float a2, int value) {   // src/ORBmatcher.cc:340-351
        float rot = a1 - a2;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * (HISTO / 360.0f));
        if (bin == HISTO) bin = 0;
        if (bin >= 0 && bin < HISTO) bins[bin].push_back(value);
    }
    template <class F> void reject_minor(F &&drop) {   // ComputeThreeMaxima + the removal loop (:380-398)
        int32_t sizes[HISTO]; int i1, i2, i3;
        for (int i = 0; i < HISTO; ++i) sizes[i] = (int32_t)bins[i].size();
        orbx_three_maxima(sizes, HISTO, &i1, &i2, &i3);
        for (int i = 0; i < HISTO; ++i) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int v : bins[i]) drop(v);
        }
    }
};

bool featvec_ok(const orbx_featvec_view &fv, int nfeat) {
    if (fv.n_nodes < 0) return false;
    if (fv.n_nodes == 0) return true;
    if (!fv.node_id || !fv.begin || (fv.begin[fv.n_nodes] > 0 && !fv.index)) return false;
    if (fv.begin[0] != 0) return false;
    for (int i = 0; i < fv.n_nodes; ++i) {
        if (fv.begin[i + 1] < fv.begin[i]) return false;
        if (i > 0 && fv.node_id[i] <= fv.node_id[i - 1]) return false;   // std::map order
    }
    for (int i = 0; i < fv.begin[fv.n_nodes]; ++i)
        if (fv.index[i] >= (uint32_t)nfeat) return false;
    return true;
}

// the reference's merge walk with map::lower_bound; `on_node(a, b)` for every common node id
template <class F> void walk_common_nodes(const orbx_featvec_view &f1, const orbx_featvec_view &f2, F &&on_node) {
    int a = 0, b = 0;
    while (a < f1.n_nodes && b < f2.n_nodes) {
        if (f1.node_id[a] == f2.node_id[b]) { on_node(a, b); ++a; ++b; }
        else if (f1.node_id[a] < f2.node_id[b]) { while (a < f1.n_nodes && f1.node_id[a] < f2.node_id[b]) ++a; }
        else { while (b < f2.n_nodes && f2.node_id[b] < f1.node_id[a]) ++b; }
    }
}

// distance blocks of the common nodes: for node pair (a, b) block[(ik - begin_a) * nb + (jf - begin_b)]
struct NodeBlocks {
    std::vector<size_t> off_a;            // per node a of f1: offset of its block (SIZE_MAX = no common node)
    std::vector<int> nb_a;                // per node a: number of features of f2's node
    std::vector<uint16_t> D;              // the distances (single call) ...
    const uint16_t *Dp = nullptr;         // ... or a view into a batch's shared download
    // rows / column indices of this pair's blocks, appended to a (possibly shared) request: second-set feature i is row
    // col_offset + i of the uploaded descriptor block, distances start at `total` of the shared output
    void plan(const orbx_featvec_view &f1, const orbx_featvec_view &f2, uint32_t col_offset, std::vector<DDistRow> &rows,
              std::vector<uint32_t> &cols, size_t &total) {
        off_a.assign((size_t)std::max(f1.n_nodes, 0), (size_t)-1);
        nb_a.assign((size_t)std::max(f1.n_nodes, 0), 0);
        walk_common_nodes(f1, f2, [&](int a, int b) {
            const int na = f1.begin[a + 1] - f1.begin[a], nb = f2.begin[b + 1] - f2.begin[b];
            if (na <= 0 || nb <= 0) return;
            off_a[(size_t)a] = total; nb_a[(size_t)a] = nb;
            const uint32_t cb = (uint32_t)cols.size();
            for (int j = f2.begin[b]; j < f2.begin[b + 1]; ++j) cols.push_back(col_offset + f2.index[j]);
            for (int i = f1.begin[a]; i < f1.begin[a + 1]; ++i) {
                DDistRow r; r.q = f1.index[i]; r.col_begin = cb; r.ncol = (uint32_t)nb; r.out_off = (uint32_t)total;
                rows.push_back(r);
                total += (size_t)nb;
            }
        });
    }
    orbx_status build(orbx_handle *h, const uint8_t *d1, int n1, const orbx_featvec_view &f1, const uint8_t *d2, int n2,
                      const orbx_featvec_view &f2) {
        std::vector<DDistRow> rows;
        std::vector<uint32_t> cols;
        size_t total = 0;
        plan(f1, f2, 0u, rows, cols, total);
        if (total > 0xffffffffull) return orbx_fail(ORBX_UNSUPPORTED, "too many descriptor pairs under common vocabulary nodes");
        const orbx_status st = orbx_block_distances(h, d1, n1, d2, n2, rows, cols, total, D);
        Dp = D.data();
        return st;
    }
    const uint16_t *row(int a, int i_in_node) const { return Dp + off_a[(size_t)a] + (size_t)i_in_node * (size_t)nb_a[(size_t)a]; }
};
}  // namespace

extern "C" orbx_status orbx_search_by_bow_keyframe_frame(orbx_handle *h, const orbx_keyframe_view *kf,
                                                         const orbx_keypoint *f_keys, const uint8_t *f_desc, int nf,
                                                         const orbx_featvec_view *f_fv, float nnratio,
                                                         int check_orientation, int32_t *matched_kf, int *nmatches_out) {
    if (!h) return orbx_fail(ORBX_BAD_ARGUMENT, "null handle");
    if (!kf || !f_fv || nf < 0 || kf->n < 0 || !matched_kf || !nmatches_out || (nf > 0 && (!f_keys || !f_desc)) ||
        (kf->n > 0 && (!kf->keys_un || !kf->desc || !kf->has_map_point)))
        return orbx_fail(ORBX_BAD_ARGUMENT, "bad argument");
    if (!featvec_ok(kf->feat_vec, kf->n) || !featvec_ok(*f_fv, nf)) return orbx_fail(ORBX_BAD_ARGUMENT, "malformed feature vector");
    *nmatches_out = 0;
    for (int i = 0; i < nf; ++i) matched_kf[i] = -1;
    if (nf == 0 || kf->n == 0) return ORBX_OK;
    const orbx_featvec_view &fk = kf->feat_vec;
    NodeBlocks nbk;
    orbx_status st = nbk.build(h, kf->desc, kf->n, fk, f_desc, nf, *f_fv);
    if (st != ORBX_OK) return st;
    int nmatches = 0;
    RotHist hist;
    walk_common_nodes(fk, *f_fv, [&](int a, int b) {
        for (int ik = fk.begin[a]; ik < fk.begin[a + 1]; ++ik) {
            const uint32_t iKF = fk.index[ik];
            if (!kf->has_map_point[iKF]) continue;
            int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
            if (f_fv->begin[b + 1] == f_fv->begin[b]) continue;
            const uint16_t *row = nbk.row(a, ik - fk.begin[a]);
            for (int jf = f_fv->begin[b]; jf < f_fv->begin[b + 1]; ++jf) {
                const uint32_t iF = f_fv->index[jf];
                if (matched_kf[iF] >= 0) continue;
                const int dist = row[jf - f_fv->begin[b]];
                if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)iF; }
                else if (dist < bestDist2) bestDist2 = dist;
            }
            if (bestDist1 <= TH_LOW_ && (float)bestDist1 < nnratio * (float)bestDist2) {
                matched_kf[bestIdxF] = (int32_t)iKF;
                if (check_orientation) hist.push(kf->keys_un[iKF].angle, f_keys[bestIdxF].angle, bestIdxF);
                nmatches++;
            }
        }
    });
    if (check_orientation) hist.reject_minor([&](int iF) { matched_kf[iF] = -1; nmatches--; });
    *nmatches_out = nmatches;
    return ORBX_OK;
}

extern "C" orbx_status orbx_search_by_bow_keyframes(orbx_handle *h, const orbx_keyframe_view *kf1,
                                                    const orbx_keyframe_view *kf2, float nnratio, int check_orientation,
                                                    int32_t *matches12, int *nmatches_out) {
    if (!h) return orbx_fail(ORBX_BAD_ARGUMENT, "null handle");
    if (!kf1 || !kf2 || kf1->n < 0 || kf2->n < 0 || !matches12 || !nmatches_out ||
        (kf1->n > 0 && (!kf1->keys_un || !kf1->desc || !kf1->has_map_point)) ||
        (kf2->n > 0 && (!kf2->keys_un || !kf2->desc || !kf2->has_map_point)))
        return orbx_fail(ORBX_BAD_ARGUMENT, "bad argument");
    if (!featvec_ok(kf1->feat_vec, kf1->n) || !featvec_ok(kf2->feat_vec, kf2->n)) return orbx_fail(ORBX_BAD_ARGUMENT, "malformed feature vector");
    *nmatches_out = 0;
    for (int i = 0; i < kf1->n; ++i) matches12[i] = -1;
    if (kf1->n == 0 || kf2->n == 0) return ORBX_OK;
    const int n2 = kf2->n;
    const orbx_featvec_view &f1 = kf1->feat_vec, &f2 = kf2->feat_vec;
    NodeBlocks nbk;
    orbx_status st = nbk.build(h, kf1->desc, kf1->n, f1, kf2->desc, n2, f2);
    if (st != ORBX_OK) return st;
    int nmatches = 0;
    RotHist hist;
    std::vector<uint8_t> matched2((size_t)n2, 0);
    walk_common_nodes(f1, f2, [&](int a, int b) {
        for (int p = f1.begin[a]; p < f1.begin[a + 1]; ++p) {
            const uint32_t i1 = f1.index[p];
            if (!kf1->has_map_point[i1]) continue;
            int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
            if (f2.begin[b + 1] == f2.begin[b]) continue;
            const uint16_t *row = nbk.row(a, p - f1.begin[a]);
            for (int q = f2.begin[b]; q < f2.begin[b + 1]; ++q) {
                const uint32_t i2 = f2.index[q];
                if (matched2[i2] || !kf2->has_map_point[i2]) continue;
                const int dist = row[q - f2.begin[b]];
                if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = (int)i2; }
                else if (dist < bestDist2) bestDist2 = dist;
            }
            if (bestDist1 < TH_LOW_ && (float)bestDist1 < nnratio * (float)bestDist2) {   // strict '<' (:802)
                matches12[i1] = bestIdx2;
                matched2[bestIdx2] = 1;
                if (check_orientation) hist.push(kf1->keys_un[i1].angle, kf2->keys_un[bestIdx2].angle, (int)i1);
                nmatches++;
            }
        }
    });
    if (check_orientation) hist.reject_minor([&](int i1) { matches12[i1] = -1; nmatches--; });
    *nmatches_out = nmatches;
    return ORBX_OK;
}

// ORBmatcher::CheckDistEpipolarLine (src/ORBmatcher.cc:206-233); the final comparison is in double (3.84 is a double)
static bool check_dist_epipolar(const orbx_keypoint &kp1, const orbx_keypoint &kp2, const float *F12, float sigma2, bool fma_mode) {
    float a, b, c, num, den;
    if (fma_mode) {   // g++ -O3 -march=native fuses the first product of each sum
        a = std::fmaf(kp1.x, F12[0], kp1.y * F12[3]) + F12[6];
        b = std::fmaf(kp1.x, F12[1], kp1.y * F12[4]) + F12[7];
        c = std::fmaf(kp1.x, F12[2], kp1.y * F12[5]) + F12[8];
        num = std::fmaf(a, kp2.x, b * kp2.y) + c;
        den = std::fmaf(a, a, b * b);
    } else {
        a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
        b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
        c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
        num = a * kp2.x + b * kp2.y + c;
        den = a * a + b * b;
    }
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return (double)dsqr < 3.84 * (double)sigma2;
}

namespace {
bool triangulation_views_ok(const orbx_keyframe_view *kf1, const orbx_keyframe_view *kf2) {
    return kf1 && kf2 && kf1->n >= 0 && kf2->n >= 0 &&
           !(kf1->n > 0 && (!kf1->keys_un || !kf1->desc || !kf1->has_map_point || !kf1->u_right)) &&
           !(kf2->n > 0 && (!kf2->keys_un || !kf2->desc || !kf2->has_map_point || !kf2->u_right || !kf2->scale_factors || !kf2->level_sigma2)) &&
           featvec_ok(kf1->feat_vec, kf1->n) && featvec_ok(kf2->feat_vec, kf2->n);
}
// ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:879-1087) from the merge walk on, over the distance blocks `nbk`
void triangulation_select(bool fma_mode, const orbx_keyframe_view *kf1, const orbx_keyframe_view *kf2, const NodeBlocks &nbk,
                          const float *F12, float ex, float ey, int only_stereo, int check_orientation, int32_t *matches12,
                          int *nmatches_out) {
    const int n2 = kf2->n;
    const orbx_featvec_view &f1 = kf1->feat_vec, &f2 = kf2->feat_vec;
    int nmatches = 0;
    RotHist hist;
    std::vector<uint8_t> matched2((size_t)n2, 0);
    walk_common_nodes(f1, f2, [&](int a, int b) {
        for (int p = f1.begin[a]; p < f1.begin[a + 1]; ++p) {
            const uint32_t i1 = f1.index[p];
            if (kf1->has_map_point[i1]) continue;                       // already triangulated (:929-932)
            const bool stereo1 = kf1->u_right[i1] >= 0;
            if (only_stereo && !stereo1) continue;
            const orbx_keypoint &kp1 = kf1->keys_un[i1];
            int bestDist = TH_LOW_, bestIdx2 = -1;
            if (f2.begin[b + 1] == f2.begin[b]) continue;
            const uint16_t *row = nbk.row(a, p - f1.begin[a]);
            for (int q = f2.begin[b]; q < f2.begin[b + 1]; ++q) {
                const uint32_t i2 = f2.index[q];
                if (matched2[i2] || kf2->has_map_point[i2]) continue;
                const bool stereo2 = kf2->u_right[i2] >= 0;
                if (only_stereo && !stereo2) continue;
                const int dist = row[q - f2.begin[b]];
                if (dist > TH_LOW_ || dist > bestDist) continue;
                const orbx_keypoint &kp2 = kf2->keys_un[i2];
                if (!stereo1 && !stereo2) {                              // too close to the epipole (:996-1003)
                    const float distex = ex - kp2.x, distey = ey - kp2.y;
                    const float d2e = fma_mode ? std::fmaf(distex, distex, distey * distey) : distex * distex + distey * distey;
                    if (d2e < 100 * kf2->scale_factors[kp2.octave]) continue;
                }
                if (check_dist_epipolar(kp1, kp2, F12, kf2->level_sigma2[kp2.octave], fma_mode)) { bestIdx2 = (int)i2; bestDist = dist; }
            }
            if (bestIdx2 >= 0) {
                matches12[i1] = bestIdx2;
                matched2[bestIdx2] = 1;                                  // fork (:1022)
                nmatches++;
                if (check_orientation) hist.push(kp1.angle, kf2->keys_un[bestIdx2].angle, (int)i1);
            }
        }
    });
    if (check_orientation)
        hist.reject_minor([&](int i1) { matched2[matches12[i1]] = 0; matches12[i1] = -1; nmatches--; });   // fork (:1069)
    *nmatches_out = nmatches;
}
}  // namespace

extern "C" orbx_status orbx_search_for_triangulation(orbx_handle *h, const orbx_keyframe_view *kf1,
                                                     const orbx_keyframe_view *kf2, const float *F12, float ex, float ey,
                                                     int only_stereo, int check_orientation, int32_t *matches12,
                                                     int *nmatches_out) {
    if (!h) return orbx_fail(ORBX_BAD_ARGUMENT, "null handle");
    if (!F12 || !matches12 || !nmatches_out || !kf1 || !kf2 || kf1->n < 0 || kf2->n < 0) return orbx_fail(ORBX_BAD_ARGUMENT, "bad argument");
    if (!triangulation_views_ok(kf1, kf2)) return orbx_fail(ORBX_BAD_ARGUMENT, "bad keyframe view / malformed feature vector");
    *nmatches_out = 0;
    for (int i = 0; i < kf1->n; ++i) matches12[i] = -1;
    if (kf1->n == 0 || kf2->n == 0) return ORBX_OK;
    NodeBlocks nbk;
    orbx_status st = nbk.build(h, kf1->desc, kf1->n, kf1->feat_vec, kf2->desc, kf2->n, kf2->feat_vec);
    if (st != ORBX_OK) return st;
    triangulation_select(orbx_handle_fp_mode(h) == ORBX_FP_GCC_FMA, kf1, kf2, nbk, F12, ex, ey, only_stereo, check_orientation, matches12, nmatches_out);
    return ORBX_OK;
}

// Batched form for the loop of LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:375-430: the current keyframe against each
// of its 10-20 best covisible neighbours).  The Hamming distances of a pair depend on descriptors and feature vectors only, so
// they are computed for ALL neighbours in one device round trip (orbx_triangulation_batch_create); the selection of neighbour k
// -- which skips features that already have a MapPoint, and the loop creates MapPoints for the matches of every neighbour before
// it searches the next -- runs on the host when the caller asks for it, with the has_map_point flags as they are THEN
// (orbx_triangulation_batch_select): the same matches as K single calls, with one ~60 us round trip instead of K.
struct orbx_triangulation_batch {
    bool fma_mode = false;
    int n1 = 0;
    std::vector<int> n2;
    std::vector<NodeBlocks> blocks;   // per neighbour: views into D
    std::vector<uint16_t> D;
};
extern "C" orbx_status orbx_triangulation_batch_create(orbx_handle *h, const orbx_keyframe_view *kf1, int nproblems,
                                                       const orbx_keyframe_view *const *kf2, orbx_triangulation_batch **out) {
    if (!h) return orbx_fail(ORBX_BAD_ARGUMENT, "null handle");
    if (!out || !kf1 || nproblems < 0 || (nproblems > 0 && !kf2)) return orbx_fail(ORBX_BAD_ARGUMENT, "bad argument");
    *out = nullptr;
    for (int k = 0; k < nproblems; ++k)
        if (!triangulation_views_ok(kf1, kf2[k])) return orbx_fail(ORBX_BAD_ARGUMENT, "bad keyframe view / malformed feature vector");
    orbx_triangulation_batch *b = new orbx_triangulation_batch();
    b->fma_mode = orbx_handle_fp_mode(h) == ORBX_FP_GCC_FMA;
    b->n1 = kf1->n;
    b->n2.resize((size_t)nproblems);
    b->blocks.resize((size_t)nproblems);
    // one request: the neighbours' descriptors back to back, every neighbour's blocks appended to the same row / column lists
    std::vector<DDistRow> rows;
    std::vector<uint32_t> cols;
    std::vector<uint8_t> d2;
    size_t total = 0, nd2 = 0;
    for (int k = 0; k < nproblems; ++k) nd2 += (size_t)kf2[k]->n;
    d2.resize(nd2 * 32);
    uint32_t off = 0;
    for (int k = 0; k < nproblems; ++k) {
        b->n2[(size_t)k] = kf2[k]->n;
        if (kf2[k]->n > 0) memcpy(d2.data() + (size_t)off * 32, kf2[k]->desc, (size_t)kf2[k]->n * 32);
        if (kf1->n > 0 && kf2[k]->n > 0) b->blocks[(size_t)k].plan(kf1->feat_vec, kf2[k]->feat_vec, off, rows, cols, total);
        off += (uint32_t)kf2[k]->n;
    }
    if (total > 0xffffffffull) { delete b; return orbx_fail(ORBX_UNSUPPORTED, "too many descriptor pairs under common vocabulary nodes"); }
    if (total > 0) {
        const orbx_status st = orbx_block_distances(h, kf1->desc, kf1->n, d2.data(), (int)nd2, rows, cols, total, b->D);
        if (st != ORBX_OK) { delete b; return st; }
    }
    for (NodeBlocks &nb : b->blocks) nb.Dp = b->D.data();
    *out = b;
    return ORBX_OK;
}
extern "C" orbx_status orbx_triangulation_batch_select(const orbx_triangulation_batch *b, int k, const orbx_keyframe_view *kf1,
                                                       const orbx_keyframe_view *kf2, const float *F12, float ex, float ey,
                                                       int only_stereo, int check_orientation, int32_t *matches12, int *nmatches_out) {
    if (!b || k < 0 || k >= (int)b->blocks.size() || !F12 || !matches12 || !nmatches_out) return orbx_fail(ORBX_BAD_ARGUMENT, "bad argument");
    if (!triangulation_views_ok(kf1, kf2) || kf1->n != b->n1 || kf2->n != b->n2[(size_t)k])
        return orbx_fail(ORBX_BAD_ARGUMENT, "the views of a selection must describe the keyframes the batch was created for");
    *nmatches_out = 0;
    for (int i = 0; i < kf1->n; ++i) matches12[i] = -1;
    if (kf1->n == 0 || kf2->n == 0) return ORBX_OK;
    triangulation_select(b->fma_mode, kf1, kf2, b->blocks[(size_t)k], F12, ex, ey, only_stereo, check_orientation, matches12, nmatches_out);
    return ORBX_OK;
}
extern "C" void orbx_triangulation_batch_destroy(orbx_triangulation_batch *b) { delete b; }

// ================================================================================================================
// Projection-guided back-end policies (second half): Fuse (both overloads),
// SearchByProjection(KeyFrame*, Scw, ...), SearchBySim3, SearchByProjection(Frame&, KeyFrame*, ...).
// The pose algebra in front of them is cv::Mat / cv::norm / MapPoint::PredictScale code that stays in the maintainer's
// shim (it IS the reference's code); the entry points start where the reference holds, per MapPoint, the flag "passed every
// geometric test", the projection, the predicted level and the representative descriptor.
// GPU: the target's grid, GetFeaturesInArea per projected point and the Hamming distances of its candidates, in the
// reference's visiting order (orbx_gate_lists; round 1 copied a dense points x features matrix to the host).
// Host: level band, chi2 gate, first-best selection and the order-dependent bookkeeping, as the reference runs them.
// ================================================================================================================

namespace {
const int TH_HIGH_ = 100;

bool target_ok(const orbx_target_view *t) {
    return t && t->n >= 0 && t->scale_factors && (t->n == 0 || (t->keys_un && t->desc)) && t->max_x > t->min_x && t->max_y > t->min_y;
}
bool points_ok(const orbx_projected_points *p) {
    return p && p->n >= 0 && (p->n == 0 || (p->valid && p->uv && p->level && p->desc));
}

// GetFeaturesInArea(u, v, radius, lo, hi) + DescriptorDistance of every valid point against the target's features, on the
// GPU (orbx_gate_lists): `radius_of(i)` is the policy's window, (lo, hi) its level arguments (-1, -1 = the plain overload)
template <class R, class LV>
orbx_status point_lists(orbx_handle *h, const orbx_target_view *t, const orbx_projected_points *p, R &&radius_of, LV &&levels_of,
                        OrbxGateLists &gl) {
    std::vector<DGateQuery> gq((size_t)p->n);
    for (int i = 0; i < p->n; ++i) {
        gq[i].x = p->uv[2 * i]; gq[i].y = p->uv[2 * i + 1]; gq[i].r = -1.0f; gq[i].min_level = gq[i].max_level = -1;
        if (!p->valid[i]) continue;
        gq[i].r = radius_of(i);
        levels_of(i, gq[i].min_level, gq[i].max_level);
    }
    return orbx_gate_lists(h, t->keys_un, t->desc, t->n, t->min_x, t->max_x, t->min_y, t->max_y, gq.data(), p->desc, p->n, gl);
}

// the reference's selection loops over candidate lists; queries of the problem start at entry q0 of the lists (batched calls)
// level band [pred - 1, pred], no chi2 gate; `taken` as the reference's vpMatched (NULL = not used)
void pick_select(const orbx_target_view *t, const orbx_projected_points *p, const OrbxGateLists &gl, int q0, int init_best, int max_dist,
                 uint8_t *taken, int32_t *best_idx, int *count) {
    int n = 0;
    for (int i = 0; i < p->n; ++i) {
        if (!p->valid[i]) continue;
        const int pred = p->level[i];
        const int nc = gl.count(q0 + i);
        const uint32_t *cl = gl.list(q0 + i);
        int bestDist = init_best, bestIdx = -1;
        for (int c = 0; c < nc; ++c) {
            const int idx = OrbxGateLists::idx(cl[c]);
            if (taken && taken[idx]) continue;
            const int lvl = t->keys_un[idx].octave;
            if (lvl < pred - 1 || lvl > pred) continue;
            const int dist = OrbxGateLists::dist(cl[c]);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= max_dist) {
            best_idx[i] = bestIdx;
            if (taken) taken[bestIdx] = 1;
            n++;
        }
    }
    *count = n;
}
orbx_status project_and_pick(orbx_handle *h, const orbx_target_view *t, const orbx_projected_points *p, float th, int init_best,
                             int max_dist, uint8_t *taken, int32_t *best_idx, int *count) {
    *count = 0;
    for (int i = 0; i < p->n; ++i) best_idx[i] = -1;
    if (p->n == 0 || t->n == 0) return ORBX_OK;
    OrbxGateLists gl;
    orbx_status st = point_lists(h, t, p, [&](int i) { return th * t->scale_factors[p->level[i]]; }, [](int, int &, int &) {}, gl);
    if (st != ORBX_OK) return st;
    pick_select(t, p, gl, 0, init_best, max_dist, taken, best_idx, count);
    return ORBX_OK;
}
// ORBmatcher::Fuse(KeyFrame *, vpMapPoints, th), lines :1168-1245 of the reference, over the candidate lists
void fuse_select(bool fma_mode, const orbx_target_view *kf, const orbx_projected_points *pts, const OrbxGateLists &gl, int q0,
                 int32_t *best_idx, int *nfused) {
    int n = 0;
    for (int i = 0; i < pts->n; ++i) {
        if (!pts->valid[i]) continue;
        const float u = pts->uv[2 * i], v = pts->uv[2 * i + 1];
        const int pred = pts->level[i];
        const int nc = gl.count(q0 + i);
        const uint32_t *cl = gl.list(q0 + i);
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nc; ++c) {
            const int idx = OrbxGateLists::idx(cl[c]);
            const orbx_keypoint &kp = kf->keys_un[idx];
            const int lvl = kp.octave;
            if (lvl < pred - 1 || lvl > pred) continue;
            if (kf->u_right[idx] >= 0) {       // stereo observation: 3-dof chi2 at 95 % (src/ORBmatcher.cc:1198-1212)
                const float ex = u - kp.x, ey = v - kp.y, er = pts->u_right[i] - kf->u_right[idx];
                const float e2 = fma_mode ? std::fmaf(er, er, std::fmaf(ex, ex, ey * ey)) : ex * ex + ey * ey + er * er;
                if ((double)(e2 * kf->inv_level_sigma2[lvl]) > 7.8) continue;
            } else {                            // monocular: 2-dof
                const float ex = u - kp.x, ey = v - kp.y;
                const float e2 = fma_mode ? std::fmaf(ex, ex, ey * ey) : ex * ex + ey * ey;
                if ((double)(e2 * kf->inv_level_sigma2[lvl]) > 5.99) continue;
            }
            const int dist = OrbxGateLists::dist(cl[c]);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW_) { best_idx[i] = bestIdx; n++; }
    }
    *nfused = n;
}
bool fuse_views_ok(const orbx_target_view *kf, const orbx_projected_points *pts) {
    return target_ok(kf) && points_ok(pts) && kf->inv_level_sigma2 && !(kf->n > 0 && !kf->u_right) && !(pts->n > 0 && !pts->u_right);
}
// K (target, point set) problems through ONE upload / grid build / gate launch / download (orbx_gate_lists_batch): the flat
// query list holds problem k's points at [q0[k], q0[k] + pts[k]->n), window th x scale factor of the predicted level
orbx_status batch_lists(orbx_handle *h, int K, const orbx_target_view *const *kfs, const orbx_projected_points *const *pts, float th,
                        std::vector<int> &q0, OrbxGateLists &gl) {
    std::vector<OrbxGateTarget> tg((size_t)K);
    q0.assign((size_t)K + 1, 0);
    for (int k = 0; k < K; ++k) {
        if (kfs[k]->min_x != kfs[0]->min_x || kfs[k]->max_x != kfs[0]->max_x || kfs[k]->min_y != kfs[0]->min_y || kfs[k]->max_y != kfs[0]->max_y)
            return orbx_fail(ORBX_BAD_ARGUMENT, "the targets of a batched call must share the image bounds");
        tg[(size_t)k] = {kfs[k]->keys_un, kfs[k]->desc, kfs[k]->n};
        q0[(size_t)k + 1] = q0[(size_t)k] + pts[k]->n;
    }
    const int nq = q0[(size_t)K];
    // SearchInNeighbors projects ONE point set into every neighbour: when all problems hand over the same descriptor block it is
    // uploaded once and every query names its row
    bool shared = K > 1;
    for (int k = 1; k < K && shared; ++k) shared = pts[k]->desc == pts[0]->desc && pts[k]->n == pts[0]->n;
    std::vector<DGateQuery> gq((size_t)nq);
    std::vector<uint8_t> qd;
    if (!shared) qd.resize((size_t)nq * 32);
    for (int k = 0; k < K; ++k) {
        const orbx_projected_points *p = pts[k];
        for (int i = 0; i < p->n; ++i) {
            DGateQuery &g = gq[(size_t)(q0[(size_t)k] + i)];
            g.x = p->uv[2 * i]; g.y = p->uv[2 * i + 1]; g.r = -1.0f; g.min_level = g.max_level = -1; g.frame = k;
            g.desc = shared ? i : -1;
            if (p->valid[i] && kfs[k]->n > 0) g.r = th * kfs[k]->scale_factors[p->level[i]];
        }
        if (!shared && p->n > 0) memcpy(qd.data() + (size_t)q0[(size_t)k] * 32, p->desc, (size_t)p->n * 32);
    }
    return orbx_gate_lists_batch(h, tg.data(), K, kfs[0]->min_x, kfs[0]->max_x, kfs[0]->min_y, kfs[0]->max_y, gq.data(),
                                 shared ? pts[0]->desc : qd.data(), nq, gl, shared ? pts[0]->n : -1);
}
}  // namespace

extern "C" orbx_status orbx_fuse(orbx_handle *h, const orbx_target_view *kf, const orbx_projected_points *pts, float th,
                                 int32_t *best_idx, int *nfused) {
    if (!h) return orbx_fail(ORBX_BAD_ARGUMENT, "null handle");
    if (!fuse_views_ok(kf, pts) || !best_idx || !nfused) return orbx_fail(ORBX_BAD_ARGUMENT, "bad argument");
    *nfused = 0;
    for (int i = 0; i < pts->n; ++i) best_idx[i] = -1;
    if (pts->n == 0 || kf->n == 0) return ORBX_OK;
    OrbxGateLists gl;
    orbx_status st = point_lists(h, kf, pts, [&](int i) { return th * kf->scale_factors[pts->level[i]]; }, [](int, int &, int &) {}, gl);
    if (st != ORBX_OK) return st;
    fuse_select(orbx_handle_fp_mode(h) == ORBX_FP_GCC_FMA, kf, pts, gl, 0, best_idx, nfused);
    return ORBX_OK;
}

extern "C" orbx_status orbx_fuse_sim3(orbx_handle *h, const orbx_target_view *kf, const orbx_projected_points *pts, float th,
                                      int32_t *best_idx, int *nfused) {
    if (!h) return orbx_fail(ORBX_BAD_ARGUMENT, "null handle");
    if (!target_ok(kf) || !points_ok(pts) || !best_idx || !nfused) return orbx_fail(ORBX_BAD_ARGUMENT, "bad argument");
    return project_and_pick(h, kf, pts, th, INT32_MAX, TH_LOW_, nullptr, best_idx, nfused);
}

// Batched forms.  The reference calls Fuse once per neighbour keyframe in a loop -- LocalMapping::
// SearchInNeighbors (src/LocalMapping.cc:750-768: every target keyframe against the current keyframe's MapPoints) and
// LoopClosing::SearchAndFuse (src/LoopClosing.cc: every connected keyframe against the loop MapPoints) -- and a synchronous
// host-buffer call costs ~60 us before it has done any work.  Here the K problems share one upload, one k_grid_build launch (K
// grids), one k_gate launch and one download; problem k's outputs are exactly what the single call returns for (kfs[k], pts[k]).
// What the loop's earlier iterations change in the map (Replace / AddObservation) reaches a later iteration only through
// pMP->isBad() / IsInKeyFrame(), which the caller re-checks when it applies the results in order (compat/ORBmatcher.h FuseBatch).
static orbx_status fuse_batch_common(orbx_handle *h, int K, const orbx_target_view *const *kfs, const orbx_projected_points *const *pts,
                                     float th, int32_t *const *best_idx, int *nfused, bool sim3) {
    if (!h) return orbx_fail(ORBX_BAD_ARGUMENT, "null handle");
    if (K < 0 || (K > 0 && (!kfs || !pts || !best_idx || !nfused))) return orbx_fail(ORBX_BAD_ARGUMENT, "bad argument");
    for (int k = 0; k < K; ++k) {
        if (!kfs[k] || !pts[k] || !(sim3 ? (target_ok(kfs[k]) && points_ok(pts[k])) : fuse_views_ok(kfs[k], pts[k])) || (pts[k]->n > 0 && !best_idx[k]))
            return orbx_fail(ORBX_BAD_ARGUMENT, "bad argument");
        nfused[k] = 0;
        for (int i = 0; i < pts[k]->n; ++i) best_idx[k][i] = -1;
    }
    if (K == 0) return ORBX_OK;
    std::vector<int> q0;
    OrbxGateLists gl;
    orbx_status st = batch_lists(h, K, kfs, pts, th, q0, gl);
    if (st != ORBX_OK) return st;
    const bool fma_mode = orbx_handle_fp_mode(h) == ORBX_FP_GCC_FMA;
    for (int k = 0; k < K; ++k) {
        if (pts[k]->n == 0 || kfs[k]->n == 0) continue;
        if (sim3) pick_select(kfs[k], pts[k], gl, q0[(size_t)k], INT32_MAX, TH_LOW_, nullptr, best_idx[k], &nfused[k]);
        else fuse_select(fma_mode, kfs[k], pts[k], gl, q0[(size_t)k], best_idx[k], &nfused[k]);
    }
    return ORBX_OK;
}
extern "C" orbx_status orbx_fuse_batch(orbx_handle *h, int nproblems, const orbx_target_view *const *kfs,
                                       const orbx_projected_points *const *pts, float th, int32_t *const *best_idx, int *nfused) {
    return fuse_batch_common(h, nproblems, kfs, pts, th, best_idx, nfused, false);
}
extern "C" orbx_status orbx_fuse_sim3_batch(orbx_handle *h, int nproblems, const orbx_target_view *const *kfs,
                                            const orbx_projected_points *const *pts, float th, int32_t *const *best_idx, int *nfused) {
    return fuse_batch_common(h, nproblems, kfs, pts, th, best_idx, nfused, true);
}

extern "C" orbx_status orbx_search_by_projection_sim3(orbx_handle *h, const orbx_target_view *kf,
                                                      const orbx_projected_points *pts, int th, uint8_t *matched,
                                                      int32_t *best_idx, int *nmatches) {
    if (!h) return orbx_fail(ORBX_BAD_ARGUMENT, "null handle");
    if (!target_ok(kf) || !points_ok(pts) || !best_idx || !nmatches || (kf->n > 0 && !matched)) return orbx_fail(ORBX_BAD_ARGUMENT, "bad argument");
    return project_and_pick(h, kf, pts, (float)th, 256, TH_LOW_, matched, best_idx, nmatches);
}

extern "C" orbx_status orbx_search_by_sim3(orbx_handle *h, const orbx_target_view *kf1, const orbx_target_view *kf2,
                                           const orbx_projected_points *pts1_in_2, const orbx_projected_points *pts2_in_1,
                                           float th, int32_t *matches12, int *nfound) {
    if (!h) return orbx_fail(ORBX_BAD_ARGUMENT, "null handle");
    if (!target_ok(kf1) || !target_ok(kf2) || !points_ok(pts1_in_2) || !points_ok(pts2_in_1) || !matches12 || !nfound ||
        pts1_in_2->n != kf1->n || pts2_in_1->n != kf2->n)
        return orbx_fail(ORBX_BAD_ARGUMENT, "bad argument (one projected point per keyframe feature is expected)");
    std::vector<int32_t> m1((size_t)kf1->n + 1), m2((size_t)kf2->n + 1);
    int c1 = 0, c2 = 0;
    orbx_status st = project_and_pick(h, kf2, pts1_in_2, th, INT32_MAX, TH_HIGH_, nullptr, m1.data(), &c1);
    if (st != ORBX_OK) return st;
    st = project_and_pick(h, kf1, pts2_in_1, th, INT32_MAX, TH_HIGH_, nullptr, m2.data(), &c2);
    if (st != ORBX_OK) return st;
    int n = 0;
    for (int i1 = 0; i1 < kf1->n; ++i1) {           // mutual agreement (:1670-1688)
        matches12[i1] = -1;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && m2[idx2] == i1) { matches12[i1] = idx2; n++; }
    }
    *nfound = n;
    return ORBX_OK;
}

extern "C" orbx_status orbx_search_by_projection_keyframe(orbx_handle *h, const orbx_target_view *cur,
                                                          const orbx_projected_points *pts, float th, int orb_dist,
                                                          int check_orientation, uint8_t *cur_has_map_point,
                                                          int32_t *matched_point, int *nmatches_out) {
    if (!h) return orbx_fail(ORBX_BAD_ARGUMENT, "null handle");
    if (!target_ok(cur) || !points_ok(pts) || !matched_point || !nmatches_out || (cur->n > 0 && !cur_has_map_point) ||
        (check_orientation && pts->n > 0 && !pts->angle))
        return orbx_fail(ORBX_BAD_ARGUMENT, "bad argument");
    *nmatches_out = 0;
    for (int i = 0; i < cur->n; ++i) matched_point[i] = -1;
    if (pts->n == 0 || cur->n == 0) return ORBX_OK;
    OrbxGateLists gl;
    orbx_status st = point_lists(h, cur, pts, [&](int i) { return th * cur->scale_factors[pts->level[i]]; },
                                 [&](int i, int &lo, int &hi) { lo = pts->level[i] - 1; hi = pts->level[i] + 1; }, gl);
    if (st != ORBX_OK) return st;
    RotHist hist;
    int nmatches = 0;
    for (int i = 0; i < pts->n; ++i) {
        if (!pts->valid[i]) continue;
        const int nc = gl.count(i);
        const uint32_t *cl = gl.list(i);
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; ++c) {
            const int i2 = OrbxGateLists::idx(cl[c]);
            if (cur_has_map_point[i2]) continue;
            const int dist = OrbxGateLists::dist(cl[c]);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= orb_dist) {
            cur_has_map_point[bestIdx2] = 1;
            matched_point[bestIdx2] = i;
            nmatches++;
            if (check_orientation) hist.push(pts->angle[i], cur->keys_un[bestIdx2].angle, bestIdx2);
        }
    }
    if (check_orientation)
        hist.reject_minor([&](int i2) { cur_has_map_point[i2] = 0; matched_point[i2] = -1; nmatches--; });
    *nmatches_out = nmatches;
    return ORBX_OK;
}
