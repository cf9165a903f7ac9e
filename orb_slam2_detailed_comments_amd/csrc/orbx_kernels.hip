// orbx_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ORB front-end.
//
// Integer / byte work bound by HBM + LDS, no MFMA.  One launch handles a whole batch of frames
// (grid.y/z = frame), image tiles are staged in LDS, per-cell keypoint compaction uses wave ballot +
// mbcnt prefix, and the quadtree of one (frame, level) runs inside one workgroup with its node list in LDS.
//
// Built with -ffp-contract=off: no implicit FMA anywhere; the only fused operations are the explicit
// __builtin_fmaf calls of the descriptor taps (fp_mode GCC_FMA) and the fma() of the pinned sincos.
#include "orbx_device.h"
#include "../../include/orbx_pattern_data.h"

// ------------------------------------------------------------------------------------------------
// K0: level 0 = reflect-101 border of the input (reference src/ORBextractor.cc:2159-2163)
// one thread -> 4 horizontally adjacent padded pixels (one dword store)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pyr_l0(DGeom g, const uint8_t *__restrict__ imgs, int W, int H, int stride,
                                                long long frame_stride, uint8_t *__restrict__ pyr) {
    const DLevel &L = g.lv[0];
    const int X = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int Y = blockIdx.y;
    const int f = blockIdx.z;
    if (X >= L.pw) return;
    const uint8_t *src = imgs + (long long)f * frame_stride + (long long)orbx_reflect101(Y - ORBX_EDGE, H) * stride;
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = X + i;
        uint32_t p = 0;
        if (x < L.pw) p = src[orbx_reflect101(x - ORBX_EDGE, W)];
        v |= p << (8 * i);
    }
    *(uint32_t *)(pyr + (long long)f * g.pyr_bytes + L.off + (long long)Y * L.pitch + X) = v;
}

// ------------------------------------------------------------------------------------------------
// K1: level l = cv::resize(INTER_LINEAR) of the PADDED level l-1 into the centre + reflect-101 border,
// in one pass: the border is produced by evaluating the bilinear formula at the reflected coordinate
// (taps precomputed per padded coordinate on the host).  (reference :2119-2143)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pyr_resize(DGeom g, int level, const OrbxTap *__restrict__ taps,
                                                    uint8_t *__restrict__ pyr) {
    const DLevel &L = g.lv[level];
    const DLevel &S = g.lv[level - 1];
    const int X = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int Y = blockIdx.y;
    const int f = blockIdx.z;
    if (X >= L.pw) return;
    uint8_t *base = pyr + (long long)f * g.pyr_bytes;
    const OrbxTap ty = taps[L.tapy + Y];
    const uint8_t *r0 = base + S.off + (long long)ty.s0 * S.pitch;
    const uint8_t *r1 = base + S.off + (long long)ty.s1 * S.pitch;
    const int b0 = ty.a0, b1 = ty.a1;
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = X + i;
        uint32_t p = 0;
        if (x < L.pw) {
            const OrbxTap tx = taps[L.tapx + x];
            const int T0 = r0[tx.s0] * tx.a0 + r0[tx.s1] * tx.a1;
            const int T1 = r1[tx.s0] * tx.a0 + r1[tx.s1] * tx.a1;
            p = (uint32_t)((((b0 * (T0 >> 4)) >> 16) + ((b1 * (T1 >> 4)) >> 16) + 2) >> 2) & 0xffu;
        }
        v |= p << (8 * i);
    }
    *(uint32_t *)(base + L.off + (long long)Y * L.pitch + X) = v;
}

// ------------------------------------------------------------------------------------------------
// K2: FAST-9/16 + score + 3x3 strict NMS per cell, with the per-cell threshold retry
// (reference src/ORBextractor.cc:1465-1548; cv::FAST semantics SURVEY App. B.1).
// One wave (64 lanes) per cell; the cell tile, its score map and the corner list live in LDS.
// Corners are compacted with wave ballot + mbcnt prefix; scores are only evaluated for compacted corners.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool orbx_arc9(uint32_t mask16) {
    uint32_t m = mask16 | (mask16 << 16);
    uint32_t a = m & (m >> 1);
    a &= a >> 2;
    a &= a >> 4;
    a &= m >> 8;
    return (a & 0xffffu) != 0;
}

// dynamic LDS: tile[rows*TP] | score[rows*TP] | list u16[lcap] | surv u32[scap]; TP = tile pitch (bytes, %4==0)
__global__ __launch_bounds__(64) void k_fast_cells(DGeom g, const OrbxCell *__restrict__ cells,
                                                   const uint8_t *__restrict__ pyr, uint2 *__restrict__ cand,
                                                   int *__restrict__ cand_count, int FAST_TP, int rows, int lcap) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fast_smem[];
    uint32_t *s_tile = (uint32_t *)fast_smem;
    uint8_t *s_score = fast_smem + rows * FAST_TP;
    uint16_t *s_list = (uint16_t *)(fast_smem + 2 * rows * FAST_TP);
    uint32_t *s_surv = (uint32_t *)(fast_smem + 2 * rows * FAST_TP + ((2 * lcap + 3) & ~3));
    const int lane = threadIdx.x;
    const OrbxCell c = cells[blockIdx.x];
    const int f = blockIdx.y;
    const DLevel &L = g.lv[c.level];
    const uint8_t *img = pyr + (long long)f * g.pyr_bytes + L.off;
    const int cw = c.cw, ch = c.ch;
    // ---- stage the tile with aligned dword loads
    const int xa = c.x0 & ~3, shift = c.x0 & 3;
    const int ndw = (shift + cw + 3) >> 2;
    for (int i = lane; i < ch * ndw; i += 64) {
        const int r = i / ndw, d = i - r * ndw;
        s_tile[r * (FAST_TP / 4) + d] = *(const uint32_t *)(img + (long long)(c.y0 + r) * L.pitch + xa + 4 * d);
    }
    const uint8_t *tile = (const uint8_t *)s_tile + shift;
    const int iw = cw - 6, ih = ch - 6, npix = iw * ih;
    // ring offsets in the LDS tile
    const int ro[16] = {3 * FAST_TP,      3 * FAST_TP + 1,  2 * FAST_TP + 2,  FAST_TP + 3, 3,  -FAST_TP + 3,
                        -2 * FAST_TP + 2, -3 * FAST_TP + 1, -3 * FAST_TP,     -3 * FAST_TP - 1, -2 * FAST_TP - 2,
                        -FAST_TP - 3,     -3,               FAST_TP - 3,      2 * FAST_TP - 2,  3 * FAST_TP - 1};
    int nsurv = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const int th = pass == 0 ? g.ini_th : g.min_th;
        for (int i = lane; i < ch * (FAST_TP / 4); i += 64) ((uint32_t *)s_score)[i] = 0;
        __syncthreads();
        // ---- phase 1: corner test, compact (lx,ly) of corners
        int ncorn = 0;
        for (int p0 = 0; p0 < npix; p0 += 64) {
            const int p = p0 + lane;
            bool corner = false;
            int lx = 0, ly = 0;
            if (p < npix) {
                ly = p / iw;
                lx = p - ly * iw + 3;
                ly += 3;
                const uint8_t *ptr = tile + ly * FAST_TP + lx;
                const int v = ptr[0];
                const int hi = v + th, lo = v - th;
                uint32_t bright = 0, dark = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int x = ptr[ro[k]];
                    bright |= (uint32_t)(x > hi) << k;
                    dark |= (uint32_t)(x < lo) << k;
                }
                corner = orbx_arc9(bright) || orbx_arc9(dark);
            }
            const unsigned long long bal = __ballot(corner);
            if (corner) {
                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
                s_list[ncorn + rank] = (uint16_t)((ly << 8) | lx);
            }
            ncorn += __popcll(bal);
        }
        __syncthreads();
        // ---- phase 2: score of every compacted corner: max(th, max_arc min d, max_arc min -d) - 1
        for (int e = lane; e < ncorn; e += 64) {
            const int lx = s_list[e] & 0xff, ly = s_list[e] >> 8;
            const uint8_t *ptr = tile + ly * FAST_TP + lx;
            const int v = ptr[0];
            int d[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) d[k] = v - (int)ptr[ro[k]];
            int mn2[16], mx2[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { mn2[k] = min(d[k], d[(k + 1) & 15]); mx2[k] = max(d[k], d[(k + 1) & 15]); }
            int mn4[16], mx4[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { mn4[k] = min(mn2[k], mn2[(k + 2) & 15]); mx4[k] = max(mx2[k], mx2[(k + 2) & 15]); }
            int a0 = th, b0 = -th;  // a0: best "ring darker" arc, b0: -(best "ring brighter" arc)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int mn9 = min(min(mn4[k], mn4[(k + 4) & 15]), d[(k + 8) & 15]);
                const int mx9 = max(max(mx4[k], mx4[(k + 4) & 15]), d[(k + 8) & 15]);
                a0 = max(a0, mn9);
                b0 = min(b0, mx9);
            }
            const int score = max(a0, -b0) - 1;
            s_score[ly * FAST_TP + lx] = (uint8_t)score;
        }
        __syncthreads();
        // ---- phase 3: 3x3 strict NMS among the corners of THIS cell only
        nsurv = 0;
        for (int e0 = 0; e0 < ncorn; e0 += 64) {
            const int e = e0 + lane;
            bool keep = false;
            uint32_t rec = 0;
            if (e < ncorn) {
                const int lx = s_list[e] & 0xff, ly = s_list[e] >> 8;
                const uint8_t *sp = s_score + ly * FAST_TP + lx;
                const int s = sp[0];
                keep = s > sp[1] && s > sp[-1] && s > sp[-FAST_TP - 1] && s > sp[-FAST_TP] && s > sp[-FAST_TP + 1] &&
                       s > sp[FAST_TP - 1] && s > sp[FAST_TP] && s > sp[FAST_TP + 1];
                rec = (uint32_t)lx | ((uint32_t)ly << 8) | ((uint32_t)s << 16);
            }
            const unsigned long long bal = __ballot(keep);
            if (keep) {
                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
                s_surv[nsurv + rank] = rec;
            }
            nsurv += __popcll(bal);
        }
        __syncthreads();
        if (nsurv > 0 || g.min_th == g.ini_th) break;  // vKeysCell.empty() -> retry with minThFAST (:1519-1527)
    }
    if (nsurv == 0) return;
    // ---- emit: one atomic per cell reserves a contiguous block of candidate slots
    int base = 0;
    if (lane == 0) base = atomicAdd(&cand_count[f * g.nlevels + c.level], nsurv);
    base = __shfl(base, 0, 64);
    uint2 *out = cand + (long long)f * g.cand_total + L.cand_begin;
    for (int e = lane; e < nsurv; e += 64) {
        const uint32_t rec = s_surv[e];
        const int lx = rec & 0xff, ly = (rec >> 8) & 0xff, s = rec >> 16;
        const int slot = base + e;
        if (slot < L.cand_cap) {
            uint2 o;
            o.x = (uint32_t)(lx + c.offx) | ((uint32_t)(ly + c.offy) << 12) | ((uint32_t)s << 24);
            o.y = ((uint32_t)c.idx_in_level << 12) | ((uint32_t)ly << 6) | (uint32_t)lx;  // emission order key
            out[slot] = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K3: DistributeOctTree (reference src/ORBextractor.cc:1050-1417) -- one workgroup per (frame, level).
//
// The reference mutates a std::list sequentially; its observable result (which nodes exist, their list
// order, which key each keeps) is reproduced here by level-synchronous passes:
//   * a key never moves: it only carries the list position of its node (knode[k]);
//   * one pass = every expandable node counts its 4 quadrants with LDS atomics, then a prefix scan over the
//     list gives each child its creation rank r; push_front order means child r lands at position C-1-r
//     and surviving old nodes follow in their old relative order;
//   * the "careful" phase (:1284-1375) sorts the expandable nodes by (count, creation order) and splits the
//     largest first until the list holds N nodes: all candidates are split speculatively, a scan over the
//     sorted order finds the cut, and only the nodes before the cut are materialised;
//   * the address tie-break of std::sort over pair<int,ExtractorNode*> is defined as creation order (F3).
// Selection (:1387-1413) = max response, first in emission order, via one 64-bit LDS atomicMax per key.
// ------------------------------------------------------------------------------------------------
#define QT_THREADS 512

struct QtShared {
    int size, prev_size, n_expand, ctot, nmtot, jstar, m, finish;
    uint32_t wsum[QT_THREADS / 64];
    uint32_t carry;
};

// exclusive prefix sum of in[0..n) -> out[0..n) (both LDS), returns total; all threads must call
__device__ uint32_t qt_block_scan(const uint32_t *in, uint32_t *out, int n, QtShared *sh) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) sh->carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += QT_THREADS) {
        const int i = base + tid;
        const uint32_t v = i < n ? in[i] : 0;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) sh->wsum[w] = incl;
        __syncthreads();
        uint32_t off = sh->carry;
        for (int j = 0; j < w; ++j) off += sh->wsum[j];
        if (i < n) out[i] = off + incl - v;
        __syncthreads();
        if (tid == 0) {
            uint32_t t = 0;
            for (int j = 0; j < QT_THREADS / 64; ++j) t += sh->wsum[j];
            sh->carry += t;
        }
        __syncthreads();
    }
    return sh->carry;
}

__device__ __forceinline__ int qt_quadrant(uint32_t pos, uint32_t b0, uint32_t b1) {
    const int x = pos & 0xfff, y = (pos >> 12) & 0xfff;
    const int x0 = b0 & 0xffff, y0 = b0 >> 16, x1 = b1 & 0xffff, y1 = b1 >> 16;
    const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);  // ceil(d/2), DivideNode :968-969
    return (x < mx ? 0 : 1) + (y < my ? 0 : 2);
}

__device__ __forceinline__ void qt_child_box(uint32_t b0, uint32_t b1, int q, uint32_t &c0, uint32_t &c1) {
    const int x0 = b0 & 0xffff, y0 = b0 >> 16, x1 = b1 & 0xffff, y1 = b1 >> 16;
    const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
    const int cx0 = (q & 1) ? mx : x0, cx1 = (q & 1) ? x1 : mx;
    const int cy0 = (q & 2) ? my : y0, cy1 = (q & 2) ? y1 : my;
    c0 = (uint32_t)cx0 | ((uint32_t)cy0 << 16);
    c1 = (uint32_t)cx1 | ((uint32_t)cy1 << 16);
}

__global__ __launch_bounds__(QT_THREADS) void k_quadtree(DGeom g, const uint2 *__restrict__ cand_all,
                                                        const int *__restrict__ cand_count,
                                                        uint32_t *__restrict__ lvl_kp, int *__restrict__ lvl_count,
                                                        int *__restrict__ status, uint16_t *__restrict__ knode_glob,
                                                        int ncap, int lds_keys) {
    extern __shared__ __attribute__((aligned(16))) uint8_t qt_smem[];
    const int level = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const DLevel &L = g.lv[level];
    const int N = L.nfeat;
    int K = cand_count[f * g.nlevels + level];
    if (K > L.cand_cap) {
        if (tid == 0) atomicMax(&status[f], (int)ORBX_CAPACITY);
        K = L.cand_cap;
    }
    const uint2 *cand = cand_all + (long long)f * g.cand_total + L.cand_begin;
    // ---- LDS carve-up (all arrays have ncap entries unless noted)
    uint8_t *sp = qt_smem;
    QtShared *sh = (QtShared *)sp;                 sp += 128;
    unsigned long long *best = (unsigned long long *)sp; sp += 8 * (size_t)ncap;
    uint32_t *boxA0 = (uint32_t *)sp;              sp += 4 * (size_t)ncap;
    uint32_t *boxA1 = (uint32_t *)sp;              sp += 4 * (size_t)ncap;
    uint32_t *cntA = (uint32_t *)sp;               sp += 4 * (size_t)ncap;
    uint32_t *metaA = (uint32_t *)sp;              sp += 4 * (size_t)ncap;   // crank | F << 16
    uint32_t *boxB0 = (uint32_t *)sp;              sp += 4 * (size_t)ncap;
    uint32_t *boxB1 = (uint32_t *)sp;              sp += 4 * (size_t)ncap;
    uint32_t *cntB = (uint32_t *)sp;               sp += 4 * (size_t)ncap;
    uint32_t *metaB = (uint32_t *)sp;              sp += 4 * (size_t)ncap;
    uint32_t *cc = (uint32_t *)sp;                 sp += 16 * (size_t)ncap;  // [ncap][4] quadrant counts
    uint32_t *newpos = (uint32_t *)sp;             sp += 16 * (size_t)ncap;  // [ncap][4] child list positions; [p][0] for survivors
    uint32_t *t0 = (uint32_t *)sp;                 sp += 4 * (size_t)ncap;   // scan input
    uint32_t *t1 = (uint32_t *)sp;                 sp += 4 * (size_t)ncap;   // scan output E
    uint32_t *t2 = (uint32_t *)sp;                 sp += 4 * (size_t)ncap;   // scan output NM
    uint32_t *t3 = (uint32_t *)sp;                 sp += 4 * (size_t)ncap;   // careful: rank / by-rank data
    uint32_t *t4 = (uint32_t *)sp;                 sp += 4 * (size_t)ncap;
    uint32_t *t5 = (uint32_t *)sp;                 sp += 4 * (size_t)ncap;
    uint16_t *knode = (K <= lds_keys) ? (uint16_t *)sp
                                      : knode_glob + ((long long)f * g.cand_total + L.cand_begin);
    uint32_t *box0 = boxA0, *box1 = boxA1, *cnt = cntA, *meta = metaA;
    uint32_t *nbox0 = boxB0, *nbox1 = boxB1, *ncnt = cntB, *nmeta = metaB;

    // ---- roots (:1060-1135)
    const int nini = L.nini;
    const float hx = L.hx;
    for (int i = tid; i < nini; i += QT_THREADS) cc[i] = 0;
    __syncthreads();
    for (int k = tid; k < K; k += QT_THREADS) {
        const int x = cand[k].x & 0xfff;
        int b = (int)((float)x / hx);
        b = min(b, nini - 1);
        knode[k] = (uint16_t)b;
        atomicAdd(&cc[b], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        for (int i = 0; i < nini; ++i) {
            if (cc[i] > 0) {
                box0[n] = (uint32_t)(int)(hx * (float)i);                       // UL.x | UL.y(=0) << 16
                box1[n] = (uint32_t)(int)(hx * (float)(i + 1)) | ((uint32_t)L.qt_h << 16);
                cnt[n] = cc[i];
                meta[n] = 0;
                t0[i] = n++;
            } else t0[i] = 0xffff;
        }
        sh->size = n;
        sh->finish = 0;
    }
    __syncthreads();
    for (int k = tid; k < K; k += QT_THREADS) knode[k] = (uint16_t)t0[knode[k]];
    __syncthreads();

    bool careful = false;
    while (true) {
        const int size = sh->size;
        if (size == 0) break;
        // ---- A: which nodes split in this pass; zero their quadrant counters
        //      main pass: every node with more than one key; careful pass: nodes created by the last pass (F)
        for (int p = tid; p < size; p += QT_THREADS) {
            const bool ex = careful ? ((meta[p] >> 16) & 1) != 0 : cnt[p] > 1;
            meta[p] = (meta[p] & 0xffffu) | ((uint32_t)ex << 16);
            cc[4 * p] = cc[4 * p + 1] = cc[4 * p + 2] = cc[4 * p + 3] = 0;
        }
        __syncthreads();
        // ---- B: quadrant census
        for (int k = tid; k < K; k += QT_THREADS) {
            const int p = knode[k];
            if ((meta[p] >> 16) & 1) atomicAdd(&cc[4 * p + qt_quadrant(cand[k].x, box0[p], box1[p])], 1u);
        }
        __syncthreads();
        // ---- C: creation ranks
        int ctot, nmtot, nexp_total = 0;
        if (!careful) {
            // pack: children (low 16) | expandable children (high 16)
            for (int p = tid; p < size; p += QT_THREADS) {
                uint32_t ne = 0, nx = 0;
                if ((meta[p] >> 16) & 1)
                    for (int q = 0; q < 4; ++q) { ne += cc[4 * p + q] > 0; nx += cc[4 * p + q] > 1; }
                t0[p] = ne | (nx << 16);
            }
            __syncthreads();
            const uint32_t tot = qt_block_scan(t0, t1, size, sh);
            ctot = tot & 0xffff;
            nexp_total = tot >> 16;
            for (int p = tid; p < size; p += QT_THREADS) {
                t0[p] = ((meta[p] >> 16) & 1) ? 0 : 1;
                t3[p] = 1;  // processed flag for expandable nodes: all of them
            }
            __syncthreads();
            nmtot = (int)qt_block_scan(t0, t2, size, sh);
        } else {
            // careful phase: order candidates by descending (count, creation rank)
            for (int p = tid; p < size; p += QT_THREADS) {
                uint32_t rank = 0xffffffffu;
                if ((meta[p] >> 16) & 1) {
                    const uint32_t key = (cnt[p] << 16) | (meta[p] & 0xffffu);
                    rank = 0;
                    for (int o = 0; o < size; ++o)
                        if (((meta[o] >> 16) & 1) && ((cnt[o] << 16) | (meta[o] & 0xffffu)) > key) ++rank;
                }
                t3[p] = rank;
            }
            if (tid == 0) { sh->m = 0; sh->jstar = 0x7fffffff; }
            __syncthreads();
            // scatter (children, children-1) by rank
            for (int p = tid; p < size; p += QT_THREADS) {
                if (t3[p] != 0xffffffffu) {
                    uint32_t ne = 0;
                    for (int q = 0; q < 4; ++q) ne += cc[4 * p + q] > 0;
                    t4[t3[p]] = ne | ((ne - 1) << 16);
                    atomicAdd(&sh->m, 1);
                }
            }
            __syncthreads();
            const int M = sh->m;
            qt_block_scan(t4, t5, M, sh);  // exclusive, packed: CE (low 16) | sum(ne-1) before (high 16)
            for (int r = tid; r < M; r += QT_THREADS) {
                const int after = size + (int)(t5[r] >> 16) + (int)(t4[r] >> 16);  // list size after splitting rank r
                if (after >= N) atomicMin(&sh->jstar, r);
            }
            __syncthreads();
            const int jstar = min(sh->jstar, M - 1);  // M == 0 -> -1: nothing is split
            ctot = jstar >= 0 ? (int)(t5[jstar] & 0xffff) + (int)(t4[jstar] & 0xffff) : 0;
            // processed flag + creation base per node
            for (int p = tid; p < size; p += QT_THREADS) {
                const uint32_t r = t3[p];
                const bool proc = r != 0xffffffffu && (int)r <= jstar;
                t1[p] = proc ? (t5[r] & 0xffff) : 0;  // E[p]
                t0[p] = proc ? 0 : 1;                 // survivor
                meta[p] = (meta[p] & 0xffffu) | ((uint32_t)proc << 16);
            }
            __syncthreads();
            nmtot = (int)qt_block_scan(t0, t2, size, sh);
        }
        // ---- D: build the next list (push_front order: child with creation rank r sits at ctot-1-r)
        for (int p = tid; p < size; p += QT_THREADS) {
            if ((meta[p] >> 16) & 1) {
                int r = (int)(t1[p] & 0xffff);
                for (int q = 0; q < 4; ++q) {
                    const uint32_t c = cc[4 * p + q];
                    if (c > 0) {
                        const int pos = ctot - 1 - r;
                        uint32_t c0, c1;
                        qt_child_box(box0[p], box1[p], q, c0, c1);
                        nbox0[pos] = c0; nbox1[pos] = c1; ncnt[pos] = c;
                        nmeta[pos] = (uint32_t)r | ((uint32_t)(c > 1) << 16);
                        newpos[4 * p + q] = pos;
                        ++r;
                    }
                }
            } else {
                const int pos = ctot + (int)t2[p];
                nbox0[pos] = box0[p]; nbox1[pos] = box1[p]; ncnt[pos] = cnt[p];
                nmeta[pos] = 0;
                newpos[4 * p] = pos;
            }
        }
        __syncthreads();
        // ---- E: keys follow their node
        for (int k = tid; k < K; k += QT_THREADS) {
            const int p = knode[k];
            const int q = ((meta[p] >> 16) & 1) ? qt_quadrant(cand[k].x, box0[p], box1[p]) : 0;
            knode[k] = (uint16_t)newpos[4 * p + q];
        }
        __syncthreads();
        { uint32_t *t;
          t = box0; box0 = nbox0; nbox0 = t;  t = box1; box1 = nbox1; nbox1 = t;
          t = cnt; cnt = ncnt; ncnt = t;      t = meta; meta = nmeta; nmeta = t; }
        const int new_size = ctot + nmtot;
        if (tid == 0) sh->size = new_size;
        __syncthreads();
        // ---- termination (:1260-1283, :1363-1372)
        if (new_size >= N || new_size == size) break;
        if (!careful && new_size + 3 * nexp_total > N) careful = true;
    }
    // ---- selection: best response, first in emission order
    const int size = sh->size;
    for (int p = tid; p < size; p += QT_THREADS) best[p] = 0ull;
    __syncthreads();
    for (int k = tid; k < K; k += QT_THREADS) {
        const uint2 c = cand[k];
        const unsigned long long key = ((unsigned long long)(((c.x >> 24) << 24) | (0xffffffu - (c.y & 0xffffffu))) << 32) | c.x;
        atomicMax(&best[knode[k]], key);
    }
    __syncthreads();
    const int nout = min(size, L.kp_cap);
    if (size > L.kp_cap && tid == 0) atomicMax(&status[f], (int)ORBX_CAPACITY);
    uint32_t *out = lvl_kp + (long long)f * g.kp_total + L.kp_begin;
    for (int p = tid; p < nout; p += QT_THREADS) out[p] = (uint32_t)best[p];
    if (tid == 0) lvl_count[f * g.nlevels + level] = nout;
}

// ------------------------------------------------------------------------------------------------
// K4: IC_Angle (reference src/ORBextractor.cc:104-161): one wave per keypoint slot; lanes 0-31 take the
// row +v, lanes 32-63 the row -v; integer moments reduced across the wave; fastAtan2 on every lane.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool orbx_slot_to_level(const DGeom &g, int slot, const int *lvl_count_f, int &level, int &idx) {
    level = 0;
    for (int l = 0; l < g.nlevels; ++l)
        if (slot >= g.lv[l].kp_begin) level = l;
    idx = slot - g.lv[level].kp_begin;
    return idx < lvl_count_f[level];
}

__global__ __launch_bounds__(256) void k_orient(DGeom g, const uint8_t *__restrict__ pyr,
                                                const uint32_t *__restrict__ lvl_kp,
                                                const int *__restrict__ lvl_count, float *__restrict__ lvl_angle) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int f = blockIdx.y;
    if (slot >= g.kp_total) return;
    int level, idx;
    if (!orbx_slot_to_level(g, slot, lvl_count + f * g.nlevels, level, idx)) return;
    const DLevel &L = g.lv[level];
    const uint32_t pos = lvl_kp[(long long)f * g.kp_total + slot];
    const int x = (int)(pos & 0xfff) + (ORBX_EDGE - 3), y = (int)((pos >> 12) & 0xfff) + (ORBX_EDGE - 3);
    const uint8_t *center = pyr + (long long)f * g.pyr_bytes + L.off + (long long)y * L.pitch + x;
    int m10 = 0, m01 = 0;
    const int u = (lane & 31) - ORBX_HALF_PATCH;  // -15..16
    if (lane < 31) m10 += u * (int)center[u];     // row v = 0
    const int sgn = lane < 32 ? 1 : -1;
    for (int v = 1; v <= ORBX_HALF_PATCH; ++v) {
        const int d = g.umax[v];
        if (u >= -d && u <= d) {
            const int val = center[u + sgn * v * L.pitch];
            m10 += u * val;
            m01 += sgn * v * val;
        }
    }
    m10 = orbx_wave_sum(m10);
    m01 = orbx_wave_sum(m01);
    if (lane == 0) lvl_angle[(long long)f * g.kp_total + slot] = orbx_fast_atan2((float)m01, (float)m10);
}

// ------------------------------------------------------------------------------------------------
// K5: GaussianBlur 7x7 sigma 2 (reference :2039-2047; OpenCV 3.2 fixed-point path, SURVEY App. B.4 +
// SSE2 column path: columns x < (w & ~3) accumulate in float with round-to-nearest-even, the last
// (w & 3) columns use the integer (s + 2^15) >> 16 tail).  Tile 64x16 per 256-thread block, staged in LDS.
// ------------------------------------------------------------------------------------------------
#define BL_TW 64
#define BL_TH 16
__global__ __launch_bounds__(256) void k_blur(DGeom g, const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur) {
    __shared__ uint8_t s_src[(BL_TH + 6) * (BL_TW + 8)];
    __shared__ uint16_t s_h[(BL_TH + 6) * BL_TW];
    const int tid = threadIdx.x, f = blockIdx.y;
    int level = 0;
    for (int l = 0; l < g.nlevels; ++l)
        if ((int)blockIdx.x >= g.lv[l].blur_tile_begin) level = l;
    const DLevel &L = g.lv[level];
    const int t = blockIdx.x - L.blur_tile_begin;
    const int ty = t / L.blur_tx, tx = t - ty * L.blur_tx;
    const int X0 = tx * BL_TW, Y0 = ty * BL_TH;
    const uint8_t *img = pyr + (long long)f * g.pyr_bytes + L.off;
    // stage (TH+6) x (TW+6) source pixels, reflect-101 at the padded image bounds
    for (int i = tid; i < (BL_TH + 6) * (BL_TW + 6); i += 256) {
        const int r = i / (BL_TW + 6), c = i - r * (BL_TW + 6);
        const int sy = orbx_reflect101(Y0 + r - 3, L.ph), sx = orbx_reflect101(X0 + c - 3, L.pw);
        s_src[r * (BL_TW + 8) + c] = img[(long long)sy * L.pitch + sx];
    }
    __syncthreads();
    // row pass: kernel {18,34,49,55,49,34,18} (float Gaussian * 256, rounded; sums to 257)
    for (int i = tid; i < (BL_TH + 6) * BL_TW; i += 256) {
        const int r = i / BL_TW, c = i - r * BL_TW;
        const uint8_t *s = s_src + r * (BL_TW + 8) + c;
        const int acc = 18 * (s[0] + s[6]) + 34 * (s[1] + s[5]) + 49 * (s[2] + s[4]) + 55 * s[3];
        s_h[i] = (uint16_t)acc;  // <= 255 * 257 = 65535
    }
    __syncthreads();
    // column pass: 4 horizontally adjacent outputs per thread
    const int cx = (tid & 15) * 4, cy = tid >> 4;
    const int Y = Y0 + cy;
    if (Y >= L.ph || X0 + cx >= L.pw) return;
    const int wv = L.pw & ~3;
    const float k0 = 55.f / 65536.f, k1 = 49.f / 65536.f, k2 = 34.f / 65536.f, k3 = 18.f / 65536.f;
    uint32_t outv = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int X = X0 + cx + i;
        const uint16_t *h = s_h + cy * BL_TW + cx + i;
        const int r0 = h[3 * BL_TW], r1 = h[2 * BL_TW] + h[4 * BL_TW], r2 = h[1 * BL_TW] + h[5 * BL_TW],
                  r3 = h[0] + h[6 * BL_TW];
        int o;
        if (X < wv) {
            float s0 = (float)r0 * k0 + 0.f;
            s0 = s0 + (float)r1 * k1;
            s0 = s0 + (float)r2 * k2;
            s0 = s0 + (float)r3 * k3;
            o = (int)__builtin_rintf(s0);
        } else {
            o = (55 * r0 + 49 * r1 + 34 * r2 + 18 * r3 + (1 << 15)) >> 16;
        }
        o = min(max(o, 0), 255);
        outv |= (uint32_t)o << (8 * i);
    }
    *(uint32_t *)(blur + (long long)f * g.pyr_bytes + L.off + (long long)Y * L.pitch + X0 + cx) = outv;
}

// ------------------------------------------------------------------------------------------------
// K6: steered BRIEF-256 + keypoint assembly (reference computeOrbDescriptor :177-254, operator() :2049-2082).
// One wave per keypoint slot; 4 rounds x 64 lanes, one test pair per lane; __ballot packs 64 descriptor
// bits per round in the reference's bit order (bit i of byte j = pair 8j+i  ==  little-endian u64 words).
// ------------------------------------------------------------------------------------------------
__constant__ signed char c_pattern[1024];

__global__ __launch_bounds__(256) void k_describe(DGeom g, const uint8_t *__restrict__ blur,
                                                  const uint32_t *__restrict__ lvl_kp,
                                                  const int *__restrict__ lvl_count,
                                                  const float *__restrict__ lvl_angle, orbx_keypoint *__restrict__ kps,
                                                  uint8_t *__restrict__ desc, int *__restrict__ counts,
                                                  int *__restrict__ status, int cap) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int f = blockIdx.y;
    if (slot >= g.kp_total) return;
    const int *lc = lvl_count + f * g.nlevels;
    int level, idx;
    const bool valid = orbx_slot_to_level(g, slot, lc, level, idx);
    int before = 0, total = 0;
    for (int l = 0; l < g.nlevels; ++l) {
        const int c = lc[l];
        if (l < level) before += c;
        total += c;
    }
    if (slot == 0 && lane == 0) {
        counts[f] = min(total, cap);
        if (total > cap) atomicMax(&status[f], (int)ORBX_CAPACITY);
    }
    if (!valid) return;
    const int oi = before + idx;
    if (oi >= cap) return;
    const DLevel &L = g.lv[level];
    const uint32_t pos = lvl_kp[(long long)f * g.kp_total + slot];
    const int x = (int)(pos & 0xfff) + (ORBX_EDGE - 3), y = (int)((pos >> 12) & 0xfff) + (ORBX_EDGE - 3);
    const float angle_deg = lvl_angle[(long long)f * g.kp_total + slot];
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float angle = angle_deg * factorPI;
    const OrbxSinCos sc = orbx_sincosf_pinned(angle);
    const float a = sc.c, b = sc.s;
    const uint8_t *center = blur + (long long)f * g.pyr_bytes + L.off + (long long)y * L.pitch + x;
    unsigned long long words[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int pair = r * 64 + lane;
        int tv[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float px = (float)c_pattern[4 * pair + 2 * s], py = (float)c_pattern[4 * pair + 2 * s + 1];
            float fy, fx;
            if (g.fp_mode == ORBX_FP_GCC_FMA) {
                fy = __builtin_fmaf(px, b, py * a);     // vfmadd132ss: x*b + rn(y*a)
                fx = __builtin_fmaf(px, a, -(py * b));  // vfmsub132ss: x*a - rn(y*b)
            } else {
                fy = px * b + py * a;
                fx = px * a - py * b;
            }
            const int iy = (int)__builtin_rintf(fy), ix = (int)__builtin_rintf(fx);
            tv[s] = center[iy * L.pitch + ix];
        }
        words[r] = __ballot(tv[0] < tv[1]);
    }
    if (lane < 4) {
        unsigned long long w = lane == 0 ? words[0] : lane == 1 ? words[1] : lane == 2 ? words[2] : words[3];
        *(unsigned long long *)(desc + ((long long)f * cap + oi) * 32 + 8 * lane) = w;
    }
    if (lane == 0) {
        orbx_keypoint kp;
        float fx = (float)x, fy = (float)y;
        if (level != 0) { fx = fx * L.scale; fy = fy * L.scale; }
        kp.x = fx; kp.y = fy;
        kp.size = L.size;
        kp.angle = angle_deg;
        kp.response = (float)(pos >> 24);
        kp.octave = level;
        kp.class_id = -1;
        kps[(long long)f * cap + oi] = kp;
    }
}

// ------------------------------------------------------------------------------------------------
// K7: brute-force Hamming best / second-best (DescriptorDistance src/ORBmatcher.cc:2073-2093 for every
// pair; bookkeeping of the search loops, e.g. :627-640).  One query per lane (8 dwords in VGPRs), train
// descriptors staged through LDS in tiles and read as wave-wide broadcasts; v_xor + v_bcnt accumulate.
// ------------------------------------------------------------------------------------------------
#define MT_TILE 256
__global__ __launch_bounds__(64) void k_match(int npairs, const uint8_t *__restrict__ q, const int *__restrict__ nq,
                                              long long q_stride, const uint8_t *__restrict__ t,
                                              const int *__restrict__ nt, long long t_stride,
                                              int *__restrict__ best_idx, int *__restrict__ best_dist,
                                              int *__restrict__ second_dist, int out_stride) {
    __shared__ uint4 s_t[MT_TILE * 2];
    const int lane = threadIdx.x, pr = blockIdx.y;
    const int NQ = nq[pr], NT = nt[pr];
    const int qi = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= NQ) return;
    const uint4 *qp = (const uint4 *)(q + (long long)pr * q_stride);
    const uint4 *tp = (const uint4 *)(t + (long long)pr * t_stride);
    uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
    if (qi < NQ) { qa = qp[2 * qi]; qb = qp[2 * qi + 1]; }
    int bd = 0x7fffffff, bd2 = 0x7fffffff, bi = -1;
    for (int t0 = 0; t0 < NT; t0 += MT_TILE) {
        const int n = min(MT_TILE, NT - t0);
        __syncthreads();
        for (int i = lane; i < 2 * n; i += 64) s_t[i] = tp[2 * t0 + i];
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            const uint4 ta = s_t[2 * j], tb = s_t[2 * j + 1];
            int d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                    __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
            if (d < bd) { bd2 = bd; bd = d; bi = t0 + j; }
            else if (d < bd2) bd2 = d;
        }
    }
    if (qi < NQ) {
        const long long o = (long long)pr * out_stride + qi;
        best_idx[o] = bi; best_dist[o] = bd; second_dist[o] = bd2;
    }
}

// full distance matrix (uint16) for host-side sequential policies
__global__ __launch_bounds__(256) void k_hamming_matrix(const uint8_t *__restrict__ q, int nq,
                                                        const uint8_t *__restrict__ t, int nt,
                                                        uint16_t *__restrict__ dist) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)nq * nt) return;
    const int qi = (int)(i / nt), ti = (int)(i - (long long)qi * nt);
    const uint4 *qp = (const uint4 *)q + 2 * qi, *tp = (const uint4 *)t + 2 * ti;
    const uint4 qa = qp[0], qb = qp[1], ta = tp[0], tb = tp[1];
    dist[i] = (uint16_t)(__popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                         __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w));
}

// small helper: zero per-batch counters / status
__global__ void k_clear(int *a, int na, int *b, int nb, int *c, int nc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na) a[i] = 0;
    if (b && i < nb) b[i] = 0;
    if (c && i < nc) c[i] = 0;
}

// ------------------------------------------------------------------------------------------------
// launch wrappers (called from orbx_api.cpp)
// ------------------------------------------------------------------------------------------------
#include "orbx_launch.h"

hipError_t orbx_upload_pattern() {
    return hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), ORBX_PATTERN_I8, 1024);
}

size_t orbx_quadtree_smem(int ncap, int lds_keys) {
    return 128 + (size_t)ncap * (8 + 8 * 4 + 16 + 16 + 6 * 4) + (size_t)lds_keys * 2 + 16;
}

void orbx_launch_clear(hipStream_t s, int *a, int na, int *b, int nb, int *c, int nc) {
    const int n = na > nb ? (na > nc ? na : nc) : (nb > nc ? nb : nc);
    hipLaunchKernelGGL(k_clear, dim3((n + 255) / 256), dim3(256), 0, s, a, na, b, nb, c, nc);
}
void orbx_launch_pyr_l0(hipStream_t s, const DGeom &g, int B, const uint8_t *imgs, int W, int H, int stride,
                        long long frame_stride, uint8_t *pyr) {
    const DLevel &L = g.lv[0];
    dim3 grid((L.pitch / 4 + 255) / 256, L.ph, B);
    hipLaunchKernelGGL(k_pyr_l0, grid, dim3(256), 0, s, g, imgs, W, H, stride, frame_stride, pyr);
}
void orbx_launch_pyr_resize(hipStream_t s, const DGeom &g, int B, int level, const OrbxTap *taps, uint8_t *pyr) {
    const DLevel &L = g.lv[level];
    dim3 grid((L.pitch / 4 + 255) / 256, L.ph, B);
    hipLaunchKernelGGL(k_pyr_resize, grid, dim3(256), 0, s, g, level, taps, pyr);
}
void orbx_launch_fast(hipStream_t s, const DGeom &g, int B, const OrbxCell *cells, const uint8_t *pyr, uint2 *cand,
                      int *cand_count, int max_cw, int max_ch) {
    if (g.ncells == 0) return;
    const int tp = (max_cw + 3 + 3) & ~3;           // +3: dword-alignment shift of the tile origin
    const int lcap = (max_cw - 6) * (max_ch - 6);   // every interior pixel could be a corner
    const int scap = ((max_cw - 6 + 1) / 2) * ((max_ch - 6 + 1) / 2);  // strict 3x3 maxima: <= 1 per 2x2 block
    const size_t smem = (size_t)2 * max_ch * tp + ((2 * lcap + 3) & ~3) + 4 * (size_t)scap;
    hipLaunchKernelGGL(k_fast_cells, dim3(g.ncells, B), dim3(64), smem, s, g, cells, pyr, cand, cand_count, tp,
                       max_ch, lcap);
}
void orbx_launch_quadtree(hipStream_t s, const DGeom &g, int B, const uint2 *cand, const int *cand_count,
                          uint32_t *lvl_kp, int *lvl_count, int *status, uint16_t *knode_glob, int ncap,
                          int lds_keys) {
    const size_t smem = orbx_quadtree_smem(ncap, lds_keys);
    hipLaunchKernelGGL(k_quadtree, dim3(g.nlevels, B), dim3(QT_THREADS), smem, s, g, cand, cand_count, lvl_kp,
                       lvl_count, status, knode_glob, ncap, lds_keys);
}
hipError_t orbx_quadtree_prepare(size_t smem) {
    return hipFuncSetAttribute((const void *)k_quadtree, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
}
void orbx_launch_orient(hipStream_t s, const DGeom &g, int B, const uint8_t *pyr, const uint32_t *lvl_kp,
                        const int *lvl_count, float *lvl_angle) {
    hipLaunchKernelGGL(k_orient, dim3((g.kp_total + 3) / 4, B), dim3(256), 0, s, g, pyr, lvl_kp, lvl_count, lvl_angle);
}
void orbx_launch_blur(hipStream_t s, const DGeom &g, int B, const uint8_t *pyr, uint8_t *blur) {
    hipLaunchKernelGGL(k_blur, dim3(g.blur_tiles, B), dim3(256), 0, s, g, pyr, blur);
}
void orbx_launch_describe(hipStream_t s, const DGeom &g, int B, const uint8_t *blur, const uint32_t *lvl_kp,
                          const int *lvl_count, const float *lvl_angle, orbx_keypoint *kps, uint8_t *desc,
                          int *counts, int *status, int cap) {
    hipLaunchKernelGGL(k_describe, dim3((g.kp_total + 3) / 4, B), dim3(256), 0, s, g, blur, lvl_kp, lvl_count,
                       lvl_angle, kps, desc, counts, status, cap);
}
void orbx_launch_match(hipStream_t s, int npairs, int max_nq, const uint8_t *q, const int *nq, long long q_stride,
                       const uint8_t *t, const int *nt, long long t_stride, int *best_idx, int *best_dist,
                       int *second_dist, int out_stride) {
    if (npairs <= 0 || max_nq <= 0) return;
    hipLaunchKernelGGL(k_match, dim3((max_nq + 63) / 64, npairs), dim3(64), 0, s, npairs, q, nq, q_stride, t, nt,
                       t_stride, best_idx, best_dist, second_dist, out_stride);
}
void orbx_launch_hamming_matrix(hipStream_t s, const uint8_t *q, int nq, const uint8_t *t, int nt, uint16_t *dist) {
    const long long n = (long long)nq * nt;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_hamming_matrix, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, q, nq, t, nt, dist);
}
