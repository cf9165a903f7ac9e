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
#define L0_ROWS 8   // rows per thread: 8 independent row loads in flight per lane (the kernel is pure streaming)
// Grid order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so with the FRAME as the fastest grid
// dimension (batch % 8 == 0) every block of one frame lands on the same XCD and the source lines two neighbouring blocks
// both touch (strip seams, reflected rows, the border block's byte gathers) are fetched from HBM once instead of once per XCD.
#ifndef L0_FF
#define L0_FF 1
#endif
#ifndef RR_FF
#define RR_FF 1
#endif
#ifndef QT_FF
#define QT_FF 1
#endif
__global__ __launch_bounds__(256) void k_pyr_l0(DGeom g, const uint8_t *__restrict__ imgs, int W, int H, int stride,
                                                long long frame_stride, uint8_t *__restrict__ pyr, int xe,
                                                int *__restrict__ status, int *__restrict__ cand_cursor) {
    // block = 64 x 4 threads, thread = L0_ROWS rows.  Blocks with blockIdx.x < gridDim.x - 1 copy the interior
    // columns [32, xe) with dword-aligned loads + funnel shifts (16 pixels per thread and row); the LAST block column
    // owns the two border strips [0, 32) and [xe, pitch) where reflect-101 reverses the byte order (byte gathers).
    // Keeping the two roles in different blocks keeps every wave free of divergence.
    const DLevel &L = g.lv[0];
#if L0_FF
    const int f = blockIdx.x, bx = blockIdx.y, by = blockIdx.z, nbx = gridDim.y;
#else
    const int f = blockIdx.z, bx = blockIdx.x, by = blockIdx.y, nbx = gridDim.x;
#endif
    // the per-frame status word (atomicMax'ed by the later kernels of the batch) is reset here: level 0 is the first kernel of
    // every batch, which saves a separate clearing launch in front of it
    if (bx == 0 && by == 0 && threadIdx.y == 0) {
        if (threadIdx.x == 0) status[f] = 0;
        if ((int)threadIdx.x < g.nlevels) cand_cursor[f * g.nlevels + threadIdx.x] = 0;   // appended to by k_fast_rows
    }
    const int Y0 = (by * 4 + threadIdx.y) * L0_ROWS;
    if (Y0 >= L.ph) return;
    const uint8_t *img = imgs + (long long)f * frame_stride;
    uint8_t *dst = pyr + (long long)f * g.pyr_bytes + L.off;
    if (bx + 1 < nbx) {
        const int X = 32 + (bx * 64 + threadIdx.x) * 16;
        if (X >= xe) return;
        uint4 v[L0_ROWS];
#pragma unroll
        for (int r = 0; r < L0_ROWS; ++r) {
            const int Y = min(Y0 + r, L.ph - 1);
            const uint8_t *sp = img + (long long)orbx_reflect101(Y - ORBX_EDGE, H) * stride + (X - ORBX_EDGE);
            const uint32_t m = (uint32_t)((unsigned long long)sp & 3ull);
            const uint32_t *ap = (const uint32_t *)(sp - m);
            const uint32_t q0 = ap[0], q1 = ap[1], q2 = ap[2], q3 = ap[3], q4 = ap[4];
            v[r] = make_uint4(__builtin_amdgcn_alignbyte(q1, q0, m), __builtin_amdgcn_alignbyte(q2, q1, m),
                              __builtin_amdgcn_alignbyte(q3, q2, m), __builtin_amdgcn_alignbyte(q4, q3, m));
        }
#pragma unroll
        for (int r = 0; r < L0_ROWS; ++r)
            if (Y0 + r < L.ph) *(uint4 *)(dst + (long long)(Y0 + r) * L.pitch + X) = v[r];   // pitch % 64 == 0
    } else {
        // border strips: dword e < 8 -> X = 4e (left 32 px); e >= 8 -> X = xe + 4(e - 8) (right, up to the pitch)
        const int e = threadIdx.x;
        const int X = e < 8 ? 4 * e : xe + 4 * (e - 8);
        if (X >= L.pitch) return;
#pragma unroll
        for (int r = 0; r < L0_ROWS; ++r) {
            const int Y = Y0 + r;
            if (Y >= L.ph) break;
            const uint8_t *src = img + (long long)orbx_reflect101(Y - ORBX_EDGE, H) * stride;
            uint32_t w = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int x = X + i;
                const uint32_t p = x < L.pw ? src[orbx_reflect101(x - ORBX_EDGE, W)] : 0u;
                w |= p << (8 * i);
            }
            *(uint32_t *)(dst + (long long)Y * L.pitch + X) = w;
        }
    }
}

// K0c: level 0 straight from an interleaved colour frame: cv::cvtColor(CV_RGB2GRAY / BGR / RGBA / BGRA) of
// Tracking::GrabImage* (reference src/Tracking.cc:245-271, 302-320, 372-385) fused into the border kernel.
// OpenCV 3.2 8-bit formula: gray = (R*4899 + G*9617 + B*1868 + 8192) >> 14.  thread = 4 padded pixels.
__global__ __launch_bounds__(256) void k_pyr_l0_color(DGeom g, const uint8_t *__restrict__ imgs, int W, int H, int stride,
                                                      long long frame_stride, uint8_t *__restrict__ pyr, int nch,
                                                      int r_off, int b_off, int *__restrict__ status, int *__restrict__ cand_cursor) {
    const DLevel &L = g.lv[0];
    const int X = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int Y = blockIdx.y * 4 + threadIdx.y;
    const int f = blockIdx.z;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.y == 0) {
        if (threadIdx.x == 0) status[f] = 0;
        if ((int)threadIdx.x < g.nlevels) cand_cursor[f * g.nlevels + threadIdx.x] = 0;
    }
    if (X >= L.pw || Y >= L.ph) return;
    const uint8_t *src = imgs + (long long)f * frame_stride + (long long)orbx_reflect101(Y - ORBX_EDGE, H) * stride;
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = X + i;
        uint32_t p = 0;
        if (x < L.pw) {
            const uint8_t *px = src + (long long)orbx_reflect101(x - ORBX_EDGE, W) * nch;
            p = (uint32_t)(px[r_off] * 4899 + px[1] * 9617 + px[b_off] * 1868 + 8192) >> 14;
        }
        v |= p << (8 * i);
    }
    *(uint32_t *)(pyr + (long long)f * g.pyr_bytes + L.off + (long long)Y * L.pitch + X) = v;
}

// cv::remap(INTER_LINEAR, float maps) of the EuRoC rectification (reference Examples/Stereo/stereo_euroc.cc:183-194)
// fused into the border kernel: padded pixel P of level 0 = rectified pixel at reflect101(P - 19), and a rectified pixel is
// the 5-bit fixed-point bilinear sample of the RAW image the host pre-digested map entry names (OpenCV 3.2 arithmetic:
// weights (32-fx)(32-fy)*32 ... summing to 2^15, + 2^14, >> 15; taps outside the raw image read 0).  The fx = fy = 0 entry of
// OpenCV's table is {32767, 0, 0, 1}: for 8-bit data that gives the same value as {32768, 0, 0, 0} (oracle keeps it literal).
// rect entry: .x = ix | iy << 16 (int16 each), .y = fy << 5 | fx.  thread = 4 padded pixels.
__global__ __launch_bounds__(256) void k_pyr_l0_remap(DGeom g, const uint8_t *__restrict__ imgs, int W, int H, int stride,
                                                      long long frame_stride, uint8_t *__restrict__ pyr,
                                                      const uint2 *__restrict__ rect, int *__restrict__ status,
                                                      int *__restrict__ cand_cursor) {
    const DLevel &L = g.lv[0];
    const int X = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int Y = blockIdx.y * 4 + threadIdx.y;
    const int f = blockIdx.z;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.y == 0) {
        if (threadIdx.x == 0) status[f] = 0;
        if ((int)threadIdx.x < g.nlevels) cand_cursor[f * g.nlevels + threadIdx.x] = 0;
    }
    if (X >= L.pw || Y >= L.ph) return;
    const uint8_t *src = imgs + (long long)f * frame_stride;
    const uint2 *mrow = rect + (long long)orbx_reflect101(Y - ORBX_EDGE, H) * W;
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = X + i;
        uint32_t p = 0;
        if (x < L.pw) {
            const uint2 m = mrow[orbx_reflect101(x - ORBX_EDGE, W)];
            const int ix = (int)(short)(m.x & 0xffffu), iy = (int)(short)(m.x >> 16);
            const uint32_t fx = m.y & 31u, fy = (m.y >> 5) & 31u;
            const bool x0 = ix >= 0 && ix < W, x1 = ix + 1 >= 0 && ix + 1 < W, y0 = iy >= 0 && iy < H, y1 = iy + 1 >= 0 && iy + 1 < H;
            const uint8_t *r0 = src + (long long)iy * stride + ix, *r1 = r0 + stride;
            const uint32_t p00 = (x0 && y0) ? r0[0] : 0u, p01 = (x1 && y0) ? r0[1] : 0u, p10 = (x0 && y1) ? r1[0] : 0u,
                           p11 = (x1 && y1) ? r1[1] : 0u;
            const uint32_t acc = __umul24((32u - fx) * (32u - fy) * 32u, p00) + __umul24(fx * (32u - fy) * 32u, p01) +
                                 __umul24((32u - fx) * fy * 32u, p10) + __umul24(fx * fy * 32u, p11);
            p = (acc + (1u << 14)) >> 15;
        }
        v |= p << (8 * i);
    }
    *(uint32_t *)(pyr + (long long)f * g.pyr_bytes + L.off + (long long)Y * L.pitch + X) = v;
}

// ------------------------------------------------------------------------------------------------
// K1: level l = cv::resize(INTER_LINEAR) of the PADDED level l-1 into the centre + reflect-101 border,
// in one pass: the border is produced by evaluating the bilinear formula at the reflected coordinate
// (taps precomputed per padded coordinate on the host).  (reference :2119-2143)
// ------------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(1))) orbx_uint2_u { uint32_t x, y; };   // 8 bytes at any byte address (global memory takes unaligned accesses)
struct __attribute__((packed, aligned(1))) orbx_uint3_u { uint32_t x, y, z; };
__device__ __forceinline__ uint2 orbx_load8(const void *p) { const orbx_uint2_u v = *(const orbx_uint2_u *)p; return make_uint2(v.x, v.y); }
struct __attribute__((aligned(4))) orbx_uint3_a { uint32_t x, y, z; };   // 12 bytes at a dword-aligned address
typedef unsigned short orbx_v2u16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t orbx_udot2(uint32_t a, uint32_t b) {   // a.lo * b.lo + a.hi * b.hi, exact
    return __builtin_amdgcn_udot2(__builtin_bit_cast(orbx_v2u16, a), __builtin_bit_cast(orbx_v2u16, b), 0u, false);
}
#define RS_ROWS 2   // destination rows per thread: the horizontal taps and byte selectors are shared, 12 loads in flight
__global__ __launch_bounds__(256) void k_pyr_resize(DGeom g, int level, const OrbxTap *__restrict__ taps,
                                                    uint8_t *__restrict__ pyr) {
    // block = 64 x 4 threads; thread = 4 horizontally adjacent destination pixels x RS_ROWS rows (dword stores).
    // Products are written with __mul24 (weights <= 2048, T >> 4 <= 32640): v_mul_lo_u32 is a quarter-rate instruction.
    // The <= 8 source pixels a thread needs per source row sit inside 3 aligned dwords: 6 dword loads per
    // destination row replace 16 byte gathers (byte-gather fallback for exotic scale factors whose footprint
    // exceeds 12 bytes).
    const DLevel &L = g.lv[level];
    const DLevel &S = g.lv[level - 1];
    const int X = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int Y0 = (blockIdx.y * 4 + threadIdx.y) * RS_ROWS;
    const int f = blockIdx.z;
    if (X >= L.pw || Y0 >= L.ph) return;
    uint8_t *base = pyr + (long long)f * g.pyr_bytes;
    OrbxTap tx[4];
    int smin = 0x7fff, smax = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        tx[i] = taps[L.tapx + min(X + i, L.pw - 1)];
        smin = min(smin, (int)tx[i].s0);
        smax = max(smax, (int)tx[i].s0);
    }
    OrbxTap ty[RS_ROWS];
    if (smax + 2 - smin <= 8) {
        // ---- narrow footprint (every usual scale factor): the <= 8 source bytes a thread needs per source row come
        // with ONE unaligned 8-byte load; v_perm_b32 puts the two taps of a destination pixel into the 16-bit halves
        // of a register and v_dot2_u32_u16 against the packed weights (a0 | a1 << 16 = the second dword of the tap
        // record) is the horizontal pass: two instructions per pixel and source row, no selects, no multiplies.
        uint32_t sel[4], wgt[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t d = (uint32_t)(tx[i].s0 - smin);              // 0..6; second tap = next byte (weight 0 when clamped)
            sel[i] = d | (0x0cu << 8) | ((d + 1u) << 16) | (0x0cu << 24);   // 0x0c selects the constant 0x00
            wgt[i] = (uint32_t)(uint16_t)tx[i].a0 | ((uint32_t)(uint16_t)tx[i].a1 << 16);
        }
        uint2 u[RS_ROWS], w[RS_ROWS];
#pragma unroll
        for (int r = 0; r < RS_ROWS; ++r) {
            ty[r] = taps[L.tapy + min(Y0 + r, L.ph - 1)];
            u[r] = orbx_load8(base + S.off + (long long)ty[r].s0 * S.pitch + smin);
            w[r] = orbx_load8(base + S.off + (long long)ty[r].s1 * S.pitch + smin);
        }
#pragma unroll
        for (int r = 0; r < RS_ROWS; ++r) {
            const int Y = Y0 + r;
            if (Y >= L.ph) break;
            const uint32_t b0 = (uint32_t)ty[r].a0 & 0xfffu, b1 = (uint32_t)ty[r].a1 & 0xfffu;
            uint32_t v = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t T0 = orbx_udot2(__builtin_amdgcn_perm(u[r].y, u[r].x, sel[i]), wgt[i]);
                const uint32_t T1 = orbx_udot2(__builtin_amdgcn_perm(w[r].y, w[r].x, sel[i]), wgt[i]);
                const uint32_t p = ((__umul24(b0, T0 >> 4) >> 16) + (__umul24(b1, T1 >> 4) >> 16) + 2u) >> 2;   // <= 255
                v |= p << (8 * i);
            }
            *(uint32_t *)(base + L.off + (long long)Y * L.pitch + X) = v;
        }
        return;
    }
    const int xb = smin & ~3;
    const bool windowed = smax + 1 - xb < 12;
    uint32_t u[RS_ROWS][3], w[RS_ROWS][3];
#pragma unroll
    for (int r = 0; r < RS_ROWS; ++r) {
        ty[r] = taps[L.tapy + min(Y0 + r, L.ph - 1)];
        if (windowed) {
            const uint32_t *q0 = (const uint32_t *)(base + S.off + (long long)ty[r].s0 * S.pitch + xb);
            const uint32_t *q1 = (const uint32_t *)(base + S.off + (long long)ty[r].s1 * S.pitch + xb);
            u[r][0] = q0[0]; u[r][1] = q0[1]; u[r][2] = q0[2];
            w[r][0] = q1[0]; w[r][1] = q1[1]; w[r][2] = q1[2];
        }
    }
#pragma unroll
    for (int r = 0; r < RS_ROWS; ++r) {
        const int Y = Y0 + r;
        if (Y >= L.ph) break;
        const int b0 = ty[r].a0, b1 = ty[r].a1;
        uint32_t v = 0;
        if (windowed) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tx[i].s0 - xb;  // 0..10; the second tap is the next byte (weight 0 whenever clamped)
                const uint32_t ulo = idx < 4 ? u[r][0] : idx < 8 ? u[r][1] : u[r][2], uhi = idx < 4 ? u[r][1] : idx < 8 ? u[r][2] : 0u;
                const uint32_t wlo = idx < 4 ? w[r][0] : idx < 8 ? w[r][1] : w[r][2], whi = idx < 4 ? w[r][1] : idx < 8 ? w[r][2] : 0u;
                const uint32_t pu = __builtin_amdgcn_alignbyte(uhi, ulo, (uint32_t)idx & 3u);
                const uint32_t pw_ = __builtin_amdgcn_alignbyte(whi, wlo, (uint32_t)idx & 3u);
                const int T0 = (int)(pu & 0xff) * tx[i].a0 + (int)((pu >> 8) & 0xff) * tx[i].a1;
                const int T1 = (int)(pw_ & 0xff) * tx[i].a0 + (int)((pw_ >> 8) & 0xff) * tx[i].a1;
                const uint32_t p = (uint32_t)((((((b0 & 0xfff) * ((T0 >> 4) & 0xffff))) >> 16) + ((((b1 & 0xfff) * ((T1 >> 4) & 0xffff))) >> 16) + 2) >> 2) & 0xffu;
                v |= p << (8 * i);
            }
        } else {
            const uint8_t *r0 = base + S.off + (long long)ty[r].s0 * S.pitch;
            const uint8_t *r1 = base + S.off + (long long)ty[r].s1 * S.pitch;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int T0 = r0[tx[i].s0] * tx[i].a0 + r0[tx[i].s1] * tx[i].a1;
                const int T1 = r1[tx[i].s0] * tx[i].a0 + r1[tx[i].s1] * tx[i].a1;
                const uint32_t p = (uint32_t)((((((b0 & 0xfff) * ((T0 >> 4) & 0xffff))) >> 16) + ((((b1 & 0xfff) * ((T1 >> 4) & 0xffff))) >> 16) + 2) >> 2) & 0xffu;
                v |= p << (8 * i);
            }
        }
        *(uint32_t *)(base + L.off + (long long)Y * L.pitch + X) = v;
    }
}

// Row-walking form of the narrow-footprint path (the one every usual geometry takes; the host checks the tap table and
// launches k_pyr_resize above otherwise).  k_pyr_resize is latency-bound: tap records -> source loads -> store is a
// chain of two dependent global loads per 512 bytes written.  Here a wave keeps its column strip's horizontal selectors /
// weights in registers and walks `rpw` destination rows: the vertical taps are scalar loads (the row is wave-uniform),
// the source rows of step k+1 are requested before step k is evaluated, and nothing in the loop waits on a table.
// (A variant that lays (row pair, dword) items out linearly over the lanes to remove the idle lanes of odd level widths
// measured 258 us against 207 us: vector tap loads, rows split across a wave.)
#ifndef RR_REVERSE
#define RR_REVERSE 1
#endif
#ifndef RR_WPB
#define RR_WPB 1   // waves (column strips x row ranges) per block; nothing is shared between them, and one-wave blocks
                   // are placed as soon as any SIMD has room (201 us against 209 with 4)
#endif
__global__ __launch_bounds__(64 * RR_WPB) void k_pyr_resize_rows(DGeom g, int level, const OrbxTap *__restrict__ taps,
                                                         uint8_t *__restrict__ pyr, int rpw) {
    const DLevel &L = g.lv[level];
    const DLevel &S = g.lv[level - 1];
    const int lane = threadIdx.x;
#if RR_FF
    // Odd levels walk their row ranges bottom-up: with the frame as the fastest grid index the launch that wrote level l-1
    // finished with the bottom rows of every frame, and what the memory-side cache still holds of that level is those rows
    const int f = blockIdx.x, bx = blockIdx.y;
    const int by = (RR_REVERSE && (level & 1)) ? (int)gridDim.z - 1 - (int)blockIdx.z : (int)blockIdx.z;
#else
    const int f = blockIdx.z, bx = blockIdx.x, by = blockIdx.y;
#endif
    const int X = (bx * 64 + lane) * 4;
    const int y_begin = __builtin_amdgcn_readfirstlane((by * RR_WPB + threadIdx.y) * rpw);
    if (y_begin >= L.ph) return;
    const int y_end = min(y_begin + rpw, L.ph);
    const bool on = X < L.pw;
    uint8_t *base = pyr + (long long)f * g.pyr_bytes;
    const uint8_t *src = base + S.off;
    // Addresses = the level's wave-uniform base (the scalar operand of the loads and stores, saddr form) + a 32-bit offset
    // (scalar row * pitch + the lane's column: one v_add per access; a level is far below 4 GB) -- round 2's form paid a 64-bit
    // vector multiply-add per row load (v_mad_i64_i32) and per store (v_mad_u64_u32)
    uint8_t *dst = base + L.off;
    const uint32_t ldst = (uint32_t)X;
    uint32_t sel[4], wgt[4];
    int smin = 0x7fff;
    {
        const uint2 *tq = (const uint2 *)taps + L.tapx;
        uint2 t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = tq[min(X + i, L.pw - 1)];   // .x = s0 | s1 << 16, .y = a0 | a1 << 16
#pragma unroll
        for (int i = 0; i < 4; ++i) smin = min(smin, (int)(t[i].x & 0xffffu));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t d = (t[i].x & 0xffffu) - (uint32_t)smin;          // 0..6 (checked on the host)
            sel[i] = d | (0x0cu << 8) | ((d + 1u) << 16) | (0x0cu << 24);
            wgt[i] = t[i].y;
        }
    }
    // A lane's 8 source bytes start at byte smin of the row: read as the dword-ALIGNED 12-byte window around them and
    // funnel-shifted in registers (two v_alignbyte).  The per-lane unaligned 8-byte load this replaces (a 4.8-byte lane
    // stride, 1-byte alignment) was what bounded the kernel: the address unit serves such a wave-load lane by lane.
    const uint32_t sh = (uint32_t)(smin & 3);
    const uint32_t lsrc = (uint32_t)(smin & ~3);
    const uint2 *ty = (const uint2 *)taps + L.tapy;
    // Two destination rows per step.  The horizontal pass of a SOURCE row (h[i] = (a0 * p[s0] + a1 * p[s0+1]) >> 4 for the
    // lane's four columns) is what cv::resize keeps in its row buffers: destination row Y+1 usually starts on the source row
    // destination row Y ended on (scale 1.2: 2.4 new source rows per two destination rows instead of 4), so the last row's
    // h values stay in registers (hp, source row `pid`) and only rows not seen yet are loaded and filtered -- 40 % fewer
    // loads, 22 % fewer vector instructions per pixel.  All conditions are wave-uniform (scalar branches).
#ifndef RR_R
#define RR_R 2
#endif
#define RR_H(dst, v)                                                                                                     \
    {                                                                                                                   \
        const uint32_t lo_ = __builtin_amdgcn_alignbyte((v).y, (v).x, sh), hi_ = __builtin_amdgcn_alignbyte((v).z, (v).y, sh); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                   \
            dst[i] = orbx_udot2(__builtin_amdgcn_perm(hi_, lo_, sel[i]), wgt[i]) >> 4;                                  \
    }
#define RR_V(out, h0, h1, w0, w1)                                                                                       \
    {                                                                                                                   \
        out = 0;                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                   \
            out |= (((__umul24(w0, h0[i]) >> 16) + (__umul24(w1, h1[i]) >> 16) + 2u) >> 2) << (8 * i);                  \
    }
    // RR_R destination rows per step (even): hb[0] / hb[1] alternate as "row s0" / "row s1" so that no value is ever copied
    uint2 t[RR_R];
    orbx_uint3_a u[RR_R], w[RR_R];
    int pid = -1;                                   // source row whose horizontal pass hb[0] holds at the top of a step
    uint32_t hb[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int r = 0; r < RR_R; ++r) {
        t[r] = ty[min(y_begin + r, L.ph - 1)];
        u[r].x = u[r].y = u[r].z = 0;
        if (r == 0 || (int)(t[r].x & 0xffffu) != (int)(t[r - 1].x >> 16)) u[r] = *(const orbx_uint3_a *)(src + (uint32_t)((t[r].x & 0xffffu) * (uint32_t)S.pitch + lsrc));
        w[r] = *(const orbx_uint3_a *)(src + (uint32_t)((t[r].x >> 16) * (uint32_t)S.pitch + lsrc));
    }
    for (int Y = y_begin; Y < y_end; Y += RR_R) {
        uint2 ct[RR_R];
        orbx_uint3_a cu[RR_R], cw[RR_R];
#pragma unroll
        for (int r = 0; r < RR_R; ++r) { ct[r] = t[r]; cu[r] = u[r]; cw[r] = w[r]; }
        if (Y + RR_R < y_end) {   // next step's rows, in flight while this step is evaluated; rows this step leaves in registers are skipped
            int last = (int)(ct[RR_R - 1].x >> 16);
#pragma unroll
            for (int r = 0; r < RR_R; ++r) {
                t[r] = ty[min(Y + RR_R + r, L.ph - 1)];
                const int s0 = (int)(t[r].x & 0xffffu), s1 = (int)(t[r].x >> 16);
                if (s0 != last) u[r] = *(const orbx_uint3_a *)(src + (uint32_t)((uint32_t)s0 * (uint32_t)S.pitch + lsrc));
                w[r] = *(const orbx_uint3_a *)(src + (uint32_t)((uint32_t)s1 * (uint32_t)S.pitch + lsrc));
                last = s1;
            }
        }
#pragma unroll
        for (int r = 0; r < RR_R; ++r) {
            uint32_t (&h0)[4] = hb[r & 1], (&h1)[4] = hb[(r & 1) ^ 1];
            const int s0 = (int)(ct[r].x & 0xffffu);
            if (s0 != pid) RR_H(h0, cu[r])
            RR_H(h1, cw[r])
            uint32_t v;
            RR_V(v, h0, h1, ct[r].y & 0xfffu, (ct[r].y >> 16) & 0xfffu)
            pid = (int)(ct[r].x >> 16);
            if (on && Y + r < y_end) *(uint32_t *)(dst + (uint32_t)((uint32_t)(Y + r) * (uint32_t)L.pitch + ldst)) = v;
        }
    }
#undef RR_H
#undef RR_V
}

// ------------------------------------------------------------------------------------------------
// K2: FAST-9/16 + score + 3x3 strict NMS per cell, with the per-cell threshold retry
// (reference src/ORBextractor.cc:1465-1548; cv::FAST semantics SURVEY App. B.1): helpers, then k_fast_rows.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool orbx_arc9(uint32_t mask16) {
    uint32_t m = mask16 | (mask16 << 16);
    uint32_t a = m & (m >> 1);
    a &= a >> 2;
    a &= a >> 4;
    a &= m >> 8;
    return (a & 0xffffu) != 0;
}

// dynamic LDS: tile[rows*TP] | score[rows*TP] | list u16[lcap] (two-ended) | corn u16[lcap]; TP = tile pitch (%4 == 0)
// ballot straight from the compare's lane mask (HIP's __ballot goes through an int and costs v_cndmask + v_cmp)
__device__ __forceinline__ unsigned long long orbx_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int orbx_wave_rank(unsigned long long bal) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
}
// LDS traffic inside one wave is ordered by the hardware; this only stops the compiler from reordering across phases
__device__ __forceinline__ void orbx_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------
// k_fast_rows: per-cell FAST-9/16 + score + NMS + threshold retry, organised so that the dense part costs a dozen vector
// instructions per image ROW (the first version, one wave per cell with a compass test per pixel pair, took 532 us per
// 256 frames against 390 us; git history has it):
//   * one wave owns a GROUP of one or two horizontally adjacent cells (OrbxFastGroup); lane i is interior column i of
//     the group and walks down the rows with the column's 7-row window in registers (one LDS byte per row), the two
//     horizontal compass pixels come from LDS with immediate offsets -- no per-pixel address arithmetic;
//   * the compass pre-test runs for ONE threshold (iniThFAST); the 4.6 % of cells that come out empty repeat the walk
//     with minThFAST (reference :1519-1527), instead of every pixel paying for both thresholds;
//   * 16-bit min/max (full rate on gfx950; the 32-bit forms are not) and a per-row ballot compaction into a bounded
//     LDS list; the list is flushed through the full ring test + score whenever the next row might not fit, so the
//     LDS footprint is independent of how many pixels pass;
//   * NMS walks the corner list when every corner of the group fitted it, otherwise it rescans the score map.
// ------------------------------------------------------------------------------------------------
#ifndef FR_TP
#define FR_TP 76        // LDS tile pitch: 64 interior columns + 6 ring + 3 alignment bytes -> 19 dwords (odd: rows spread over all banks)
#endif
// (Staging the tile by LDS-DMA -- global_load_lds_dwordx4 at pitch 80, no VGPR round trip, no ds_write pass -- was built and
// measured in round 2: bit-exact, but 408-414 us against 402 us for the register-prefetch staging below, also with the next
// tile requested before the NMS of the current one: the staging instructions are ~5 % of the kernel's issue slots and the
// register prefetch hides the load latency better than a wait on vmcnt does.  Not kept; git history has it.)
#ifndef FR_WPS
#define FR_WPS 5
#endif
#ifndef FR_CCAP
#define FR_CCAP 512     // corner-list entries per group (see orbx_launch_fast_rows): 8 192 bytes of LDS per wave at 640x480, still 20 waves per CU
#endif
#ifndef FR_GPW
#define FR_GPW 2        // groups per wave
#endif
typedef unsigned short fr_u16;
typedef short fr_i16;
typedef __attribute__((address_space(3))) uint16_t fr_lds_u16;
__device__ __forceinline__ fr_u16 fr_max(fr_u16 a, fr_u16 b) { return a > b ? a : b; }
__device__ __forceinline__ fr_u16 fr_min(fr_u16 a, fr_u16 b) { return a < b ? a : b; }
__device__ __forceinline__ fr_i16 fr_smax(fr_i16 a, fr_i16 b) { return a > b ? a : b; }
__device__ __forceinline__ fr_i16 fr_smin(fr_i16 a, fr_i16 b) { return a < b ? a : b; }

static_assert(64 + ORBX_FAST_XCOLS + 6 + 3 <= FR_TP, "a cell pair's tile row must fit the LDS tile pitch");
__device__ __forceinline__ bool colx_on_mask(unsigned act) { return (act & 2u) != 0; }   // columns >= 64 belong to the second cell
struct FrCtx {
    const uint8_t *tile;   // LDS tile, byte (0,0) = sub-mat origin of the group's first cell
    uint8_t *score;        // LDS score map, same coordinates
    uint16_t *list;        // LDS candidate / corner work list (lcap entries)
    uint16_t *corn;        // LDS corner list of the whole group (lcap entries)
    int lcap;
    int lane;
};

#ifndef FR_UNIFIED
#define FR_UNIFIED 1
#endif
#if FR_UNIFIED
// Corner test and score of list[0..n) at threshold th in ONE pass per candidate; corners compacted IN PLACE to the front of the
// list, their scores written to the score map; returns the number of corners.
//   cv::FAST's test (9 contiguous ring pixels all brighter than v + th, or all darker than v - th) is "the largest over the 16
// arcs of the smallest signed difference on the arc exceeds th" -- which is cv::cornerScore's own quantity (score + 1).  So the
// min/max network of the score IS the test: 16-bit min / max issue at the full rate, where the ring masks of the separate test
// (v_cmp + v_addc per ring pixel and polarity, fr_ring_and_score below) are all half-rate, and the second pass over the corners
// (17 LDS reads per corner again) disappears.
//   One polarity per candidate: d = ring - v, and the compass pixels (ring 0 / 4 / 8 / 12: the walk's pre-test) say which margin
// is the larger one, s1 = min(max(d0, d8), max(d4, d12)) (brighter) or -s2 = -max(min(d0, d8), min(d4, d12)) (darker).  Darker
// candidates run the same network on ~d = (v - ring) - 1 with the start value th - 1 (one xor per ring pixel, no negation):
// corner <=> result > start, score = result - 1 - f with f = 0 / -1.  A candidate that passes BOTH pre-tests (a pixel half way
// up a strong edge: < 1 % of the candidates) and fails its larger polarity may still be a corner of the other one: it is pushed on
// a small stack (the 64 dummy dwords the walk's masked lanes write to, free during these rounds) and re-run with the polarity
// forced the other way in a later round; the stack is drained whenever it could not take another round's worth.
// one round of up to 64 candidates.  FULL: 64 entries of the main list, every lane busy, nothing selected per lane.  Otherwise lanes
// [0, cm) take main entries and lanes [cm, cm + cs) entries popped from the stack, which run with the polarity forced the other way.
template <bool FULL>
__device__ __forceinline__ void fr_round(const FrCtx &c, const uint16_t *src_main, int cm, const uint16_t *src_stack, int cs,
                                         int th, uint16_t *stack, int &ncorn, int &nredo) {
    const int rc = 3 * FR_TP + 3;
    const int ro[16] = {rc + 3 * FR_TP,      rc + 3 * FR_TP + 1,  rc + 2 * FR_TP + 2,  rc + FR_TP + 3, rc + 3,  rc - FR_TP + 3,
                        rc - 2 * FR_TP + 2, rc - 3 * FR_TP + 1, rc - 3 * FR_TP,     rc - 3 * FR_TP - 1, rc - 2 * FR_TP - 2,
                        rc - FR_TP - 3,     rc - 3,             rc + FR_TP - 3,      rc + 2 * FR_TP - 2,  rc + 3 * FR_TP - 1};
    const bool is_main = FULL || c.lane < cm;
    uint16_t code;
    fr_i16 thl;
    if (FULL) {
        code = src_main[c.lane];
        thl = (fr_i16)th;
    } else {
        const bool valid = c.lane < cm + cs;
        const uint16_t *src = is_main ? src_main + c.lane : src_stack + (c.lane - cm);
        code = valid ? *src : (uint16_t)(3 << 8);
        // lanes without an entry carry a threshold no 8-bit difference reaches: no lane mask in the tests below
        thl = valid ? (fr_i16)th : (fr_i16)0x4000;
    }
    const int off = ((code >> 8) - 3) * FR_TP + (code & 0xff);   // window corner: row - 3, column - 3 (rows start at 3)
    const uint8_t *ptr = c.tile + off;
#ifndef FR_SGN
#define FR_SGN 1
#endif
#if FR_SGN
    // d = +-(ring - v) as ONE multiply-add per ring pixel (sgn * x - sgn * v), the sign from the four compass pixels first
    const int v = ptr[rc];
    int x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = ptr[ro[k]];
    const fr_i16 c0 = (fr_i16)(x[0] - v), c4 = (fr_i16)(x[4] - v), c8 = (fr_i16)(x[8] - v), c12 = (fr_i16)(x[12] - v);
    const fr_i16 s1 = fr_smin(fr_smax(c0, c8), fr_smax(c4, c12));
    const fr_i16 s2 = fr_smax(fr_smin(c0, c8), fr_smin(c4, c12));
    // both pre-tests pass (th < 256: no 16-bit overflow anywhere here)
    const bool both = fr_smin(s1, (fr_i16)(-s2)) > thl;
    // -1: the darker margin is the larger one; an entry from the stack takes the other polarity
    int sgn = ((int)(fr_i16)(s1 + s2) >> 15) | 1;
    if (!FULL) sgn = is_main ? sgn : -sgn;
    const int cv = __mul24(-sgn, v);
    fr_i16 d[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int t;   // (as written in C the compiler factors the sign out again: a subtraction and a multiplication per pixel)
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t) : "v"(x[k]), "v"(sgn), "v"(cv));
        d[k] = (fr_i16)t;
    }
    const fr_i16 start = thl;
    const fr_i16 f = 0;
#else
    const fr_i16 v = (fr_i16)ptr[rc];
    fr_i16 d[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) d[k] = (fr_i16)((fr_i16)ptr[ro[k]] - v);
    const fr_i16 s1 = fr_smin(fr_smax(d[0], d[8]), fr_smax(d[4], d[12]));
    const fr_i16 s2 = fr_smax(fr_smin(d[0], d[8]), fr_smin(d[4], d[12]));
    // both pre-tests pass (th < 256: no 16-bit overflow anywhere here)
    const bool both = fr_smin(s1, (fr_i16)(-s2)) > thl;
    // -1: the darker margin is the larger one; an entry from the stack takes the other polarity
    uint32_t f32 = (uint32_t)(int)(fr_i16)((fr_i16)(s1 + s2) >> 15);
    if (!FULL) f32 ^= is_main ? 0u : 0xffffffffu;
    asm("" : "+v"(f32));   // ONE register: the 16 flips below stay plain v_xor (the compiler otherwise folds the terms into 16 three-input ops)
    const fr_i16 f = (fr_i16)f32;
#pragma unroll
    for (int k = 0; k < 16; ++k) d[k] = (fr_i16)(d[k] ^ f);
    const fr_i16 start = (fr_i16)(thl + f);
#endif
    fr_i16 a0 = start;
    fr_i16 m2[16], m4[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) m2[k] = fr_smin(d[k], d[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) m4[k] = fr_smin(m2[k], m2[(k + 2) & 15]);
#ifndef FR_NOMIN3
#define FR_NOMIN3 1
#endif
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#if FR_NOMIN3
        // two-input forms only: v_min3_i16 / v_max3_i16 issue at a QUARTER of the rate of v_min_i16 on gfx950 (8.4 against 2.8
        // cycles per wave, tools/valu_rate6.hip) -- three two-input operations cost as much as one three-input one
        fr_i16 m8 = fr_smin(m4[k], m4[(k + 4) & 15]);
        asm("" : "+v"(m8));
        fr_i16 w9 = fr_smin(m8, d[(k + 8) & 15]);
        asm("" : "+v"(w9));
        a0 = fr_smax(a0, w9);
        asm("" : "+v"(a0));
#else
        a0 = fr_smax(a0, fr_smin(fr_smin(m4[k], m4[(k + 4) & 15]), d[(k + 8) & 15]));
#endif
    }
    const bool corner = a0 > start;
    const unsigned long long m = orbx_ballot(corner);
    if (corner) {
        c.list[ncorn + orbx_wave_rank(m)] = code;   // index < the entries already read: this round's are in registers
        c.score[off + rc] = (uint8_t)(a0 - 1 - f);
    }
    ncorn += __popcll(m);
    // (the mask of the re-runs from the two compare masks on the scalar unit; per-lane work only inside the rare branch)
    unsigned long long mr = orbx_ballot(both) & ~m;
    if (!FULL) mr &= cm >= 64 ? ~0ull : ((1ull << cm) - 1ull);
    if (mr != 0ull) {
        const bool redo = both && !corner && is_main;
        if (redo) stack[nredo + orbx_wave_rank(mr)] = code;
        nredo += __popcll(mr);
    }
}
__device__ __forceinline__ int fr_ring_and_score(const FrCtx &c, int n, int th, int dbg_stop) {
    uint16_t *const stack = c.list + c.lcap;   // 128 entries
    int ncorn = 0, nredo = 0, e0 = 0;
    // full rounds of the main list (a round pushes at most 64 entries: with at most 64 on the stack before it, 128 hold them)
    for (; n - e0 >= 64 && nredo <= 64; e0 += 64) fr_round<true>(c, c.list + e0, 64, stack, 0, th, stack, ncorn, nredo);
    // the rest of the main list and the stack, the stack in whatever lanes the main entries leave free
    for (;;) {
        const int cm = nredo > 64 ? 0 : min(n - e0, 64);
        const int cs = min(nredo, 64 - cm);
        if (cm + cs == 0) break;
        nredo -= cs;
        fr_round<false>(c, c.list + e0, cm, stack + nredo, cs, th, stack, ncorn, nredo);
        e0 += cm;
    }
    orbx_wave_sync();
    (void)dbg_stop;
    return ncorn;
}
#else
// full 16-ring test of list[0..n) at threshold th, corners compacted IN PLACE to the front of the list, their scores
// written to the score map; returns the number of corners
__device__ __forceinline__ int fr_ring_and_score(const FrCtx &c, int n, int th, int dbg_stop) {
    // offsets from the TOP-LEFT corner of the pixel's 7x7 window: none is negative, so every ring read is base + immediate (a DS
    // offset is unsigned: the seven negative offsets of a centre-based table cost a v_add each, per round)
    const int rc = 3 * FR_TP + 3;
    const int ro[16] = {rc + 3 * FR_TP,      rc + 3 * FR_TP + 1,  rc + 2 * FR_TP + 2,  rc + FR_TP + 3, rc + 3,  rc - FR_TP + 3,
                        rc - 2 * FR_TP + 2, rc - 3 * FR_TP + 1, rc - 3 * FR_TP,     rc - 3 * FR_TP - 1, rc - 2 * FR_TP - 2,
                        rc - FR_TP - 3,     rc - 3,             rc + FR_TP - 3,      rc + 2 * FR_TP - 2,  rc + 3 * FR_TP - 1};
    int ncorn = 0;
    for (int e0 = 0; e0 < n; e0 += 64) {
        const int e = e0 + c.lane;
        const bool valid = e < n;
        const uint16_t code = valid ? c.list[e] : (uint16_t)(3 << 8);
        const uint8_t *ptr = c.tile + ((code >> 8) - 3) * FR_TP + (code & 0xff);   // window corner: row - 3, column - 3 (rows start at 3)
        const int v = ptr[rc];
        const int hi = v + th, lo = v - th;
        // ring masks by shift-in: mask = 2*mask + (compare) is one v_cmp + one v_addc per ring pixel and polarity
        // (ring position k lands on bit 15-k; a circular run of 9 is a run of 9 in either direction)
        uint32_t bright = 0, dark = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int x = ptr[ro[k]];
            // (the compiler's own lowering is v_cmp -> SGPR, s_nop hazard padding, v_cndmask, v_or3, v_lshl)
            asm("v_cmp_gt_i32 vcc, %2, %3\n\t"
                "v_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                "v_cmp_lt_i32 vcc, %2, %4\n\t"
                "v_addc_co_u32 %1, vcc, %1, %1, vcc"
                : "+v"(bright), "+v"(dark)
                : "v"(x), "v"(hi), "v"(lo)
                : "vcc");
        }
        // (a pixel cannot have 9 brighter AND 9 darker ring pixels: the polarity of a corner is unique)
        const bool cb = orbx_arc9(bright), cd = orbx_arc9(dark);
        const bool corner = (int)valid & ((int)cb | (int)cd);
        const unsigned long long m = orbx_ballot(corner);
        // index <= e: this round's entries are already in registers.  Bit 15 = ring brighter than the centre.
        if (corner) c.list[ncorn + orbx_wave_rank(m)] = (uint16_t)(code | (cb ? 0x8000u : 0u));
        ncorn += __popcll(m);
    }
    orbx_wave_sync();
    if (dbg_stop == 3) return ncorn;
    // score = (largest t for which the corner test still passes) = max over the 16 arcs of the min over the arc of the
    // signed difference, minus 1 (cv::cornerScore<16>).  The polarity that failed the test cannot exceed th, so only
    // the corner's own polarity is evaluated: d = +-(v - ring), sliding 9-window minima by doubling (2, 4, 8, +1).
    for (int e = c.lane; e < ncorn; e += 64) {
        const uint16_t code = c.list[e];
        const int off = ((code >> 8) & 0x7f) * FR_TP + 3 + (code & 0xff);
        const uint8_t *ptr = c.tile + off - rc;
        // darker ring: complement both sides (255 - x) - (255 - v) = v - x, so one instruction stream serves both cases
        const fr_u16 flip = (code & 0x8000u) != 0 ? (fr_u16)0 : (fr_u16)0xff;
        const fr_i16 v = (fr_i16)((fr_u16)ptr[rc] ^ flip);
        fr_i16 d[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k] = (fr_i16)((fr_i16)((fr_u16)ptr[ro[k]] ^ flip) - v);
        fr_i16 a0 = (fr_i16)th;
        fr_i16 m2[16], m4[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) m2[k] = fr_smin(d[k], d[(k + 1) & 15]);
#pragma unroll
        for (int k = 0; k < 16; ++k) m4[k] = fr_smin(m2[k], m2[(k + 2) & 15]);
#pragma unroll
        for (int k = 0; k < 16; ++k) a0 = fr_smax(a0, fr_smin(fr_smin(m4[k], m4[(k + 4) & 15]), d[(k + 8) & 15]));
        c.score[off] = (uint8_t)(a0 - 1);
    }
    orbx_wave_sync();
    return ncorn;
}
#endif


// strict 3x3 NMS of list[0..n) among the corners of the SAME cell, survivors straight to the cell's slot range
struct FrCells {
    int iw0;                        // interior columns of the first cell (64 when the group is a single cell)
    int offx0, offx1, offy;         // j*wCell of the two cells, i*hCell
    int ord0, ord1;                 // idx_in_level
    int cap0, cap1;                 // slot_cap: survivors a cell may report (exact NMS worst case unless max_cand_per_cell cut it)
    uint2 *out;                     // dense candidate array of this (frame, level)
    int *cursor;                    // its fill count: one returning atomicAdd per NMS round reserves the round's survivors
    int out_cap;                    // entries the array holds
};
__device__ __forceinline__ void fr_nms(const FrCtx &c, const FrCells &gc, const uint16_t *list, int n, int &ns0, int &ns1) {
    for (int e0 = 0; e0 < n; e0 += 64) {
        const int e = e0 + c.lane;
        const bool valid = e < n;
        const uint16_t code = valid ? list[e] : (uint16_t)(3 << 8);
        const int col = code & 0xff, ly = (code >> 8) & 0x7f;
        const bool second = col >= gc.iw0;
        const uint8_t *sp = c.score + (ly - 1) * FR_TP + 2 + col;   // top-left of the 3x3 neighbourhood: immediates only
        const int sc = sp[FR_TP + 1];
        const int l0 = sp[0], l1 = sp[FR_TP], l2 = sp[2 * FR_TP];
        const int r0 = sp[2], r1 = sp[FR_TP + 2], r2 = sp[2 * FR_TP + 2];
        const int u = sp[1], dn = sp[2 * FR_TP + 1];
        // the score map is shared by the two cells: the neighbours across the seam belong to the other cell's cv::FAST call
        // (strictly greater than all eight <=> strictly greater than their maximum: one select per side, not one per neighbour)
        const bool seam_l = col == gc.iw0, seam_r = col == gc.iw0 - 1;
        const int ml = seam_l ? 0 : max(max(l0, l1), l2), mr = seam_r ? 0 : max(max(r0, r1), r2);
        const bool keep = (int)valid & (int)(sc > max(max(ml, mr), max(u, dn)));
        const unsigned long long m = orbx_ballot(keep), msec = orbx_ballot(keep && second);
        const unsigned long long mfirst = m & ~msec;
        const int slot = second ? ns1 + orbx_wave_rank(msec) : ns0 + orbx_wave_rank(mfirst);   // ordinal inside the cell
        const bool ok = keep && slot < (second ? gc.cap1 : gc.cap0);
        const unsigned long long mok = orbx_ballot(ok);
        if (mok != 0ull) {
            // The survivors go straight into the level's dense key array (any order: every record carries its emission-order
            // key, and the quadtree is order-free): no per-cell slot ranges, no gather pass in front of the quadtree.
            int base = 0;
            if (c.lane == 0) base = atomicAdd(gc.cursor, (int)__popcll(mok));
            base = __builtin_amdgcn_readfirstlane(base);
            const int pos = base + orbx_wave_rank(mok);
            if (ok && pos < gc.out_cap) {
                const int lx = 3 + col - (second ? gc.iw0 : 0);
                uint2 o;
                o.x = (uint32_t)(lx + (second ? gc.offx1 : gc.offx0)) | ((uint32_t)(ly + gc.offy) << 12) | ((uint32_t)sc << 24);
                o.y = ((uint32_t)(second ? gc.ord1 : gc.ord0) << 12) | ((uint32_t)ly << 6) | (uint32_t)lx;   // emission order key
                gc.out[pos] = o;
            }
        }
        ns0 += __popcll(mfirst);
        ns1 += __popcll(msec);
    }
}

__global__ __launch_bounds__(64, FR_WPS) void k_fast_rows(DGeom g, const OrbxCell *__restrict__ cells,
                                                          const OrbxFastGroup *__restrict__ groups,
                                                          const uint8_t *__restrict__ pyr, uint2 *__restrict__ cand,
                                                          int *__restrict__ cand_cursor, int *__restrict__ status, int rows,
                                                          int lcap, int ngroups, int gpw, int dbg_stop, int ccap) {
    // dbg_stop (ORBX_FAST_STOP, phase-timing builds only, -DORBX_TIMING_KNOBS; results are wrong unless 0): 1 = after
    // staging, 2 = after the pre-test, 3 = after the ring test, 4 = before NMS.  The shipped library pins it to 0.
#ifndef ORBX_TIMING_KNOBS
    dbg_stop = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) uint8_t fast_smem[];
    uint32_t *s_tile = (uint32_t *)fast_smem;
    // tile | score map (16-byte aligned: cleared with 16-byte stores) | work list (lcap entries + one private dummy dword per
    // lane) | corner list of the group (ccap entries: a group with more corners takes the dense NMS rescan)
    const int map_bytes = (rows * FR_TP + 15) & ~15;
    uint8_t *s_score = fast_smem + map_bytes;
    uint16_t *s_list = (uint16_t *)(s_score + map_bytes);
    uint16_t *s_corn = s_list + lcap + 128;
    const int lane = threadIdx.x;
    const int f = blockIdx.x;   // frame fastest: all groups of one frame share one XCD's L2
    // FR_GPW groups per wave, one after the other: the next group's tile is fetched into registers while this one
    // is processed, so the global-load latency is never waited for
    const int g0 = blockIdx.y * gpw;
    const int ng = min(gpw, ngroups - g0);
    // staging: a lane loads 12 bytes (one load, a third of the address arithmetic and of the load instructions of a
    // dword per lane), 7 lanes cover the 19 dwords of a tile row, 9 rows per step, 5 steps = 45 rows in registers.
    // Row offsets are 32-bit adds from the first row's offset, clamped to the cell's last row; the address is the
    // frame's scalar base + that 32-bit offset (global_load saddr form, no 64-bit vector arithmetic).  The lanes of a
    // row's last 12-byte piece read past the tile into the level's next bytes (the pyramid slab has slack at its end).
    const int rq = (lane * 37) >> 8, dq = lane - 7 * rq;   // lane / 7, lane % 7  (lane 63: rq = 9, idle)
    orbx_uint3_u tv[5];
    OrbxFastGroup grp_n = groups[g0];
    OrbxCell c0_n = cells[grp_n.cell0], c1_n = cells[grp_n.cell0 + grp_n.ncell - 1];
    const uint8_t *fbase = pyr + (long long)f * g.pyr_bytes;
#define FR_PREFETCH()                                                                                                     \
    {                                                                                                                     \
        const DLevel &Ln = g.lv[c0_n.level];                                                                             \
        const uint8_t *srcn = fbase + Ln.off;                                                                             \
        uint32_t vpitch9 = (uint32_t)(9 * Ln.pitch);                                                                      \
        asm("" : "+v"(vpitch9));   /* in a VGPR: a VOP2 add with an SGPR source issues at the slow rate */                   \
        const uint32_t olast = (uint32_t)(__mul24((int)c0_n.y0 + (int)c0_n.ch - 1, Ln.pitch) + (c0_n.x0 & ~3) + 12 * dq); \
        uint32_t o = (uint32_t)(__mul24((int)c0_n.y0 + min(rq, 8), Ln.pitch) + (c0_n.x0 & ~3) + 12 * dq);                \
        _Pragma("unroll") for (int k = 0; k < 5; ++k) {                                                                   \
            tv[k] = *(const orbx_uint3_u *)(srcn + min(o, olast));                                                        \
            o += vpitch9;                                                                                                 \
        }                                                                                                                 \
    }
    FR_PREFETCH()
  for (int gi = 0; gi < ng; ++gi) {
    const OrbxFastGroup grp = grp_n;
    const OrbxCell c0 = c0_n, c1 = c1_n;
    const DLevel &L = g.lv[c0.level];
    const int tw = c1.x0 + c1.cw - c0.x0, th_rows = c0.ch;
    const int niw = tw - 6;                                   // interior columns of the group (<= 64)
    const int iw0 = grp.ncell == 2 ? c0.cw - 6 : 64;
    // ---- stage the tile: prefetched registers -> LDS
    {
        const int xa = c0.x0 & ~3;
        uint32_t *trow = s_tile + rq * (FR_TP / 4) + 3 * dq;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            if (rq < 9 && 9 * k + rq < th_rows) {
                uint32_t *d = trow + 9 * k * (FR_TP / 4);
                d[0] = tv[k].x;
                if (dq < 6) { d[1] = tv[k].y; d[2] = tv[k].z; }   // a row has 19 dwords: the 7th piece is one dword
            }
        }
        if (th_rows > 45) {   // cells taller than the register window (tiny pyramid levels only)
            const uint8_t *src = fbase + L.off + (long long)c0.y0 * L.pitch + xa + 12 * dq;
            for (int r = 45 + rq; r < th_rows; r += 9) {
                if (rq < 9) {
                    const orbx_uint3_u v = *(const orbx_uint3_u *)(src + (long long)r * L.pitch);
                    uint32_t *d = s_tile + r * (FR_TP / 4) + 3 * dq;
                    d[0] = v.x;
                    if (dq < 6) { d[1] = v.y; d[2] = v.z; }
                }
            }
        }
        // score map cleared with 16-byte stores (the launcher rounds the row count to a multiple of 4, so the map is
        // 16-byte aligned and a few bytes past the cell's last row still belong to it or to the not-yet-used list)
        for (int i = lane; i < (th_rows * (FR_TP / 4) + 3) / 4; i += 64) ((uint4 *)s_score)[i] = make_uint4(0, 0, 0, 0);
    }
    if (gi + 1 < ng) {
        grp_n = groups[g0 + gi + 1];
        c0_n = cells[grp_n.cell0]; c1_n = cells[grp_n.cell0 + grp_n.ncell - 1];
        FR_PREFETCH()
    }
    FrCtx cx;
    cx.tile = (const uint8_t *)s_tile + (c0.x0 & 3);
    cx.score = s_score;
    cx.list = s_list;
    cx.corn = s_corn;
    cx.lcap = lcap;
    cx.lane = lane;
    FrCells gc;
    gc.iw0 = iw0;
    gc.offx0 = c0.offx; gc.offx1 = c1.offx; gc.offy = c0.offy;
    gc.ord0 = c0.idx_in_level; gc.ord1 = c1.idx_in_level;
    gc.cap0 = c0.slot_cap; gc.cap1 = c1.slot_cap;
    gc.out = cand + (long long)f * g.cand_total + L.cand_begin;
    gc.cursor = cand_cursor + f * g.nlevels + c0.level;
    gc.out_cap = L.cand_cap;
    orbx_wave_sync();
    if (dbg_stop == 1) continue;
    const bool two_th = g.min_th != g.ini_th;
    const bool colv = lane < niw;
    const bool second = lane >= iw0;
    const uint8_t *pc = cx.tile + 3 + (colv ? lane : 0);   // this lane's column, row 0
    const int yend = th_rows - 3;
    int ns0 = 0, ns1 = 0;
    unsigned act = grp.ncell == 2 ? 3u : 1u;   // cells still to be detected in this pass
    for (int pass = 0; pass < 2; ++pass) {
        const int th = pass == 0 ? g.ini_th : g.min_th;
        const bool lane_on = colv && ((act >> (second ? 1 : 0)) & 1u);
        int nctot = 0;
        bool overflow = false;
        int y = 3;
        // lanes that are switched off (outside the group, or a cell that already has keypoints) carry a threshold no
        // 8-bit difference reaches: the pre-test needs no separate lane mask
        const fr_i16 thv = lane_on ? (fr_i16)th : (fr_i16)0x4000;
        // A pair of cells may be up to ORBX_FAST_XCOLS columns wider than the wave (two 33-column cells: level 2 and 5 of 640x480 would
        // otherwise run one cell per wave, half the lanes idle; ORBX_FAST_XCOLS).  The row walk covers columns 0..63; the columns beyond are
        // tested afterwards with the lanes as ROWS, one column per step, into the same work list.
        int xc = 64;
        const fr_i16 thx = (colx_on_mask(act) && 3 + lane < yend) ? (fr_i16)th : (fr_i16)0x4000;
        while (y < yend || xc < niw) {
            // ---- compass pre-test, one row per step: a 9-arc of the 16-ring contains one pixel of every opposite
            // pair, so max(min(max(r0,r8),max(r4,r12)) - v, v - max(min(r0,r8),min(r4,r12))) > th is necessary.
            // The column window (rows y-3 .. y+3) rotates through seven registers: one new LDS byte per row.
            // Scalar instructions issue at the same rate as vector ones, so the step is written to need only two
            // (popcount + list cursor): no per-row exit test, no exec masking (lanes without a candidate write to a
            // private dummy slot), the lane mask folded into the threshold.
            const uint8_t *pr = pc + y * FR_TP;
            fr_u16 w0 = pr[-3 * FR_TP], w1 = pr[-2 * FR_TP], w2 = pr[-FR_TP], w3 = pr[0], w4 = pr[FR_TP], w5 = pr[2 * FR_TP], w6;
            // list cursor as an LDS byte ADDRESS (scalar): a lane's slot is one v_lshl_add away
            const uint32_t list0 = (uint32_t)(uintptr_t)(fr_lds_u16 *)s_list;
            uint32_t nb = list0;
            uint32_t code = (uint32_t)((y << 8) | lane);
            const uint32_t dummy = list0 + 2u * (uint32_t)lcap + 4u * (uint32_t)lane;
            // software-pipelined: the three LDS bytes of row y+1 are requested before row y is evaluated (the list
            // store could alias them as far as the compiler knows, so it would not hoist them itself); the row after
            // the last one is read but never used (it is the first row of the score map)
            uint32_t nx0 = pr[3 * FR_TP], nx4 = pr[3], nx12 = pr[-3];   // (32-bit carriers: no re-extension in the loop)
            uint32_t probe_s = 0, probe_v = 0; (void)probe_s; (void)probe_v;
#ifndef FR_WALK_IF
#define FR_WALK_IF 1
#endif
// issue-cost probes (measurement builds only; the result bits do not change): FR_PROBE_S = 4 scalar, FR_PROBE_V = 4 vector
// filler instructions per walked row (profiles/r03_fast_issue_probe.md)
#if defined(FR_PROBE_S)
#define FR_PROBE() asm volatile("s_xor_b32 %0, %0, 1\n\ts_xor_b32 %0, %0, 1\n\ts_xor_b32 %0, %0, 1\n\ts_xor_b32 %0, %0, 1" : "+s"(probe_s));
#elif defined(FR_PROBE_V)
#define FR_PROBE() asm volatile("v_xor_b32 %0, 1, %0\n\tv_xor_b32 %0, 1, %0\n\tv_xor_b32 %0, 1, %0\n\tv_xor_b32 %0, 1, %0" : "+v"(probe_v));
#else
#define FR_PROBE()
#endif
#if FR_WALK_IF
#define FR_STORE(cnd, m, nb, dummy, code) (void)(dummy); if (cnd) *(fr_lds_u16 *)(uintptr_t)((nb) + 2u * (uint32_t)orbx_wave_rank(m)) = (uint16_t)(code);
#else
#define FR_STORE(cnd, m, nb, dummy, code) *(fr_lds_u16 *)(uintptr_t)((cnd) ? (nb) + 2u * (uint32_t)orbx_wave_rank(m) : (dummy)) = (uint16_t)(code);
#endif
#define FR_STEP(R8, C, R0)                                                                                              \
            {                                                                                                           \
                R0 = (fr_u16)nx0;                                                                                       \
                const fr_u16 r4 = (fr_u16)nx4, r12 = (fr_u16)nx12;                                                      \
                pr += FR_TP;                                                                                            \
                nx0 = pr[3 * FR_TP]; nx4 = pr[3]; nx12 = pr[-3];                                                        \
                asm("" : "+v"(nx0), "+v"(nx4), "+v"(nx12));   /* keeps the carriers 32-bit (no v_and / SDWA re-extension) */ \
                const fr_u16 A = fr_min(fr_max(R0, R8), fr_max(r4, r12));                                               \
                const fr_u16 Bm = fr_max(fr_min(R0, R8), fr_min(r4, r12));                                              \
                const bool cnd = fr_smax((fr_i16)(A - C), (fr_i16)(C - Bm)) > thv;                                      \
                const unsigned long long m = orbx_ballot(cnd);                                                          \
                FR_STORE(cnd, m, nb, dummy, code)                                                                       \
                { const uint32_t pc_ = (uint32_t)__popcll(m);                                                           \
                  asm("s_lshl1_add_u32 %0, %1, %0" : "+s"(nb) : "s"(pc_) : "scc"); }   /* nb += 2 * popcount: ONE scalar instruction, ONE scalar cursor (a lane's slot stays mbcnt, mbcnt, v_lshl_add) */ \
                code += 0x100u;                                                                                         \
                FR_PROBE()                                                                                              \
            }
            // full chunks of 7 rows while the list is guaranteed to take them
            while (y + 7 <= yend && nb + 7u * 128u <= list0 + 2u * (uint32_t)lcap) {
                FR_STEP(w0, w3, w6)
                FR_STEP(w1, w4, w0)
                FR_STEP(w2, w5, w1)
                FR_STEP(w3, w6, w2)
                FR_STEP(w4, w0, w3)
                FR_STEP(w5, w1, w4)
                FR_STEP(w6, w2, w5)
                y += 7;
            }
            // remaining rows (and small work lists) one at a time, shifting the window
            while (y < yend && nb + 128u <= list0 + 2u * (uint32_t)lcap) {
                FR_STEP(w0, w3, w6)
                w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = w5; w5 = w6;
                ++y;
            }
#undef FR_STEP
            // the columns beyond the wave (rows done): lane = row 3 + lane, one column per step while the list takes 64 more
            while (y >= yend && xc < niw && nb + 128u <= list0 + 2u * (uint32_t)lcap) {
                const uint8_t *px = cx.tile + (3 + lane) * FR_TP + 3 + xc;
                const fr_u16 C = px[0], r0 = px[-3 * FR_TP], r8 = px[3 * FR_TP], r4 = px[3], r12 = px[-3];
                const fr_u16 A = fr_min(fr_max(r0, r8), fr_max(r4, r12));
                const fr_u16 Bm = fr_max(fr_min(r0, r8), fr_min(r4, r12));
                const bool cnd = fr_smax((fr_i16)(A - C), (fr_i16)(C - Bm)) > thx;
                const unsigned long long m = orbx_ballot(cnd);
                if (cnd) *(fr_lds_u16 *)(uintptr_t)(nb + 2u * (uint32_t)orbx_wave_rank(m)) = (uint16_t)(((3 + lane) << 8) | xc);
                nb += 2u * (uint32_t)__popcll(m);
                ++xc;
            }
            const int n = (int)((nb - list0) >> 1);
            orbx_wave_sync();
            if (dbg_stop == 2) { if (n == 12345) cand_cursor[0] = n; continue; }
            const int ncorn = fr_ring_and_score(cx, n, th, dbg_stop);
            // keep the corners for the NMS walk while they fit
            if (!overflow && nctot + ncorn <= ccap) {
                for (int e = lane; e < ncorn; e += 64) s_corn[nctot + e] = s_list[e];
                nctot += ncorn;
            } else {
                overflow = true;
            }
            orbx_wave_sync();
        }
        // ---- NMS
        int a0 = 0, a1 = 0;
        if (dbg_stop >= 2) { a0 = a1 = 1; } else
        if (!overflow) {
            fr_nms(cx, gc, s_corn, nctot, a0, a1);
        } else {
            int xs = 64;
            for (int yy = 3; yy < yend || xs < niw;) {
                int n = 0;
                for (; yy < yend && n + 64 <= lcap; ++yy) {
                    const bool cnd = lane_on && s_score[yy * FR_TP + 3 + lane] != 0;
                    const unsigned long long m = orbx_ballot(cnd);
                    if (cnd) s_list[n + orbx_wave_rank(m)] = (uint16_t)((yy << 8) | lane);
                    n += __popcll(m);
                }
                for (; yy >= yend && xs < niw && n + 64 <= lcap; ++xs) {   // columns beyond the wave: lane = row
                    const bool cnd = colx_on_mask(act) && 3 + lane < yend && s_score[(3 + lane) * FR_TP + 3 + xs] != 0;
                    const unsigned long long m = orbx_ballot(cnd);
                    if (cnd) s_list[n + orbx_wave_rank(m)] = (uint16_t)(((3 + lane) << 8) | xs);
                    n += __popcll(m);
                }
                orbx_wave_sync();
                fr_nms(cx, gc, s_list, n, a0, a1);
                orbx_wave_sync();
            }
        }
        if (act & 1u) ns0 = a0;
        if (act & 2u) ns1 = a1;
        if (pass == 1 || !two_th) break;
        // vKeysCell.empty() -> that cell alone repeats with minThFAST (:1519-1527)
        act = (ns0 == 0 ? 1u : 0u) | ((grp.ncell == 2 && ns1 == 0) ? 2u : 0u);
        if (act == 0) break;
        orbx_wave_sync();
        // score map cleared with 16-byte stores (the launcher rounds the row count to a multiple of 4, so the map is
        // 16-byte aligned and a few bytes past the cell's last row still belong to it or to the not-yet-used list)
        for (int i = lane; i < (th_rows * (FR_TP / 4) + 3) / 4; i += 64) ((uint4 *)s_score)[i] = make_uint4(0, 0, 0, 0);
        orbx_wave_sync();
    }
    // more survivors than the cell may report (only when max_cand_per_cell cut the exact worst case): reported, not silent
    if (lane == 0 && (ns0 > gc.cap0 || (grp.ncell == 2 && ns1 > gc.cap1))) atomicMax(&status[f], (int)ORBX_CAPACITY);
    orbx_wave_sync();   // the next group overwrites tile / score / lists
  }
#undef FR_PREFETCH
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
// Workgroup size is a template parameter, picked per launch (orbx_launch_quadtree): the kernel is a chain of barrier-separated
// stages, so many small workgroups win when the launch has several workgroups per CU to choose from (256 threads: 67 us per
// 256 frames of 640x480 / 1000 features against 82 us with 512), and large ones when there are few workgroups with many keys
// each (1920x1080 / 4000 features, 32 frames = one workgroup per CU: 185 / 112 / 82 us with 256 / 512 / 1024 threads).

struct QtShared {
    int size, prev_size, n_expand, ctot, nmtot, jstar, m, finish;
    uint32_t wsum[16];   // one per wave, up to 1024 threads
    uint32_t carry;
};

// exclusive prefix sum of in[0..n) -> out[0..n) (both LDS, may alias), returns total; all threads must call.
// Two barriers per 512-element chunk: wave scan -> wave totals in LDS -> every thread adds the totals before it.
template <int QT_THREADS>
__device__ uint32_t qt_block_scan(const uint32_t *in, uint32_t *out, int n, QtShared *sh) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint32_t carry = 0;
    for (int base = 0; base < n; base += QT_THREADS) {
        const int i = base + tid;
        const uint32_t v = i < n ? in[i] : 0;
        const uint32_t incl = (uint32_t)orbx_wave_scan((int)v);
        if (lane == 63) sh->wsum[w] = incl;
        __syncthreads();
        uint32_t off = carry, tot = 0;
#pragma unroll
        for (int j = 0; j < QT_THREADS / 64; ++j) {
            const uint32_t t = sh->wsum[j];
            off += j < w ? t : 0u;
            tot += t;
        }
        if (i < n) out[i] = off + incl - v;
        carry += tot;
        __syncthreads();   // results visible to every thread; wsum free for the next chunk / call
    }
    return carry;
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

template <int QT_THREADS>
__global__ __launch_bounds__(QT_THREADS) void k_quadtree(DGeom g, const uint2 *__restrict__ dense_all,
                                                        const int *__restrict__ cand_count,
                                                        uint32_t *__restrict__ lvl_kp, int *__restrict__ lvl_count,
                                                        int *__restrict__ status, uint16_t *__restrict__ knode_glob,
                                                        int ncap, int lds_keys, int level_begin) {
    extern __shared__ __attribute__((aligned(16))) uint8_t qt_smem[];
    // grid: frame fastest, level 0 (the most keys) dispatched first.  With the level fastest, workgroup id % 8 = level for
    // the usual 8 levels: one XCD would get every level-0 workgroup and another every level-7 one.
#if QT_FF
    const int level = level_begin + blockIdx.y, f = blockIdx.x, tid = threadIdx.x;
#else
    const int level = level_begin + blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
#endif
    const DLevel &L = g.lv[level];
    const int N = L.nfeat;
    const uint2 *cand = dense_all + (long long)f * g.cand_total + L.cand_begin;
    // ---- LDS carve-up (all arrays have ncap entries unless noted)
    uint8_t *sp = qt_smem;
    QtShared *sh = (QtShared *)sp;                 sp += 128;
    uint32_t *boxA0 = (uint32_t *)sp;              sp += 4 * (size_t)ncap;
    uint32_t *boxA1 = (uint32_t *)sp;              sp += 4 * (size_t)ncap;
    uint32_t *cntA = (uint32_t *)sp;               sp += 4 * (size_t)ncap;
    uint32_t *metaA = (uint32_t *)sp;              sp += 4 * (size_t)ncap;   // crank | F << 16
    uint32_t *boxB0 = (uint32_t *)sp;              sp += 4 * (size_t)ncap;
    uint32_t *boxB1 = (uint32_t *)sp;              sp += 4 * (size_t)ncap;
    uint32_t *cntB = (uint32_t *)sp;               sp += 4 * (size_t)ncap;
    uint32_t *metaB = (uint32_t *)sp;              sp += 4 * (size_t)ncap;
    uint32_t *cc = (uint32_t *)sp;                 sp += 16 * (size_t)ncap;  // [ncap][4] quadrant counts
    unsigned long long *best = (unsigned long long *)cc;                      // selection keys: the quadrant counts are dead by then
    uint16_t *newpos = (uint16_t *)sp;             sp += 8 * (size_t)ncap + 16;  // [ncap][4] child list positions (< ncap); [p][0] for survivors
    uint32_t *t0 = (uint32_t *)sp;                 sp += 4 * (size_t)ncap;   // scan input
    uint32_t *t1 = (uint32_t *)sp;                 sp += 4 * (size_t)ncap;   // scan output E
    uint32_t *t2 = (uint32_t *)sp;                 sp += 4 * (size_t)ncap;   // scan output NM
    uint32_t *t3 = (uint32_t *)sp;                 sp += 4 * (size_t)ncap;   // careful: rank / by-rank data
    uint32_t *t4 = (uint32_t *)sp;                 sp += 4 * (size_t)ncap;
    uint32_t *t5 = (uint32_t *)sp;                 sp += 4 * (size_t)ncap;
    uint32_t *kpos_lds = (uint32_t *)sp;           sp += 4 * (size_t)lds_keys;  // key positions (x:12 y:12 score:8) when K <= lds_keys
    uint16_t *knode_lds = (uint16_t *)sp;
    uint32_t *box0 = boxA0, *box1 = boxA1, *cnt = cntA, *meta = metaA;
    uint32_t *nbox0 = boxB0, *nbox1 = boxB1, *ncnt = cntB, *nmeta = metaB;

    // ---- keys: k_fast_rows appended this level's survivors to the dense array (count in cand_count); nothing to gather
    const int K = min(cand_count[f * g.nlevels + level], L.cand_cap);
    const bool keys_in_lds = K <= lds_keys;   // wave-uniform: key positions and key->node map live in LDS
    if (keys_in_lds)
        for (int k = tid; k < K; k += QT_THREADS) kpos_lds[k] = cand[k].x;
    uint16_t *knode = keys_in_lds ? knode_lds : knode_glob + ((long long)f * g.cand_total + L.cand_begin);
#define QT_KPOS(k) (keys_in_lds ? kpos_lds[k] : cand[k].x)
    __syncthreads();
    // ---- roots (:1060-1135)
    const int nini = L.nini;
    const float hx = L.hx;
    for (int i = tid; i < nini; i += QT_THREADS) cc[i] = 0;
    __syncthreads();
    for (int k = tid; k < K; k += QT_THREADS) {
        const int x = QT_KPOS(k) & 0xfff;
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
            if ((meta[p] >> 16) & 1) atomicAdd(&cc[4 * p + qt_quadrant(QT_KPOS(k), box0[p], box1[p])], 1u);
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
            const uint32_t tot = qt_block_scan<QT_THREADS>(t0, t1, size, sh);
            ctot = tot & 0xffff;
            nexp_total = tot >> 16;
            for (int p = tid; p < size; p += QT_THREADS) {
                t0[p] = ((meta[p] >> 16) & 1) ? 0 : 1;
                t3[p] = 1;  // processed flag for expandable nodes: all of them
            }
            __syncthreads();
            nmtot = (int)qt_block_scan<QT_THREADS>(t0, t2, size, sh);
        } else {
            // careful phase: order candidates by descending (count, creation rank).  The candidate keys are first
            // compacted into a dense, 16-byte aligned LDS array (scratch = newpos, rewritten in phase D anyway); each
            // candidate then counts the larger keys with ds_read_b128 over M/4 steps instead of walking the whole list.
            uint32_t *dk = (uint32_t *)newpos, *dp = dk + (size_t)ncap + 4;   // dense keys [M + 4], back references [M]: 8 ncap + 16 bytes
            for (int p = tid; p < size; p += QT_THREADS) { t0[p] = (meta[p] >> 16) & 1; t3[p] = 0xffffffffu; }
            __syncthreads();
            const int Mc = (int)qt_block_scan<QT_THREADS>(t0, t1, size, sh);
            for (int p = tid; p < size; p += QT_THREADS)
                if ((meta[p] >> 16) & 1) { dk[t1[p]] = (cnt[p] << 16) | (meta[p] & 0xffffu); dp[t1[p]] = p; }
            if (tid < 4) dk[Mc + tid] = 0u;   // padding compares as "not larger"
            __syncthreads();
            for (int i = tid; i < Mc; i += QT_THREADS) {
                const uint32_t key = dk[i];
                uint32_t rank = 0;
                for (int j = 0; j < Mc; j += 4) {
                    const uint4 v = *(const uint4 *)(dk + j);
                    rank += (v.x > key) + (v.y > key) + (v.z > key) + (v.w > key);
                }
                t3[dp[i]] = rank;
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
            qt_block_scan<QT_THREADS>(t4, t5, M, sh);  // exclusive, packed: CE (low 16) | sum(ne-1) before (high 16)
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
            nmtot = (int)qt_block_scan<QT_THREADS>(t0, t2, size, sh);
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
                        newpos[4 * p + q] = (uint16_t)pos;
                        ++r;
                    }
                }
            } else {
                const int pos = ctot + (int)t2[p];
                nbox0[pos] = box0[p]; nbox1[pos] = box1[p]; ncnt[pos] = cnt[p];
                nmeta[pos] = 0;
                newpos[4 * p] = (uint16_t)pos;
            }
        }
        __syncthreads();
        // ---- E: keys follow their node
        for (int k = tid; k < K; k += QT_THREADS) {
            const int p = knode[k];
            const int q = ((meta[p] >> 16) & 1) ? qt_quadrant(QT_KPOS(k), box0[p], box1[p]) : 0;
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
// K4: IC_Angle (reference src/ORBextractor.cc:104-161) is fused into k_describe (K6): the orientation disc lies
// inside the LDS patch that kernel stages anyway.  Only the slot -> (level, index) helper remains here.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void orbx_slot_to_level(const DGeom &g, int slot, int &level, int &idx) {
    level = 0;
#pragma unroll
    for (int l = 1; l < ORBX_MAX_LEVELS; ++l)
        if (l < g.nlevels && slot >= g.lv[l].kp_begin) level = l;
    idx = slot - g.lv[level].kp_begin;
}

// ------------------------------------------------------------------------------------------------
// K5: GaussianBlur 7x7 sigma 2 (reference :2039-2047; OpenCV 3.2 fixed-point path, SURVEY App. B.4 +
// SSE2 column path: columns x < (w & ~3) accumulate in float with round-to-nearest-even, the last
// (w & 3) columns use the integer (s + 2^15) >> 16 tail).  Tile 64x16 per 256-thread block, staged in LDS.
// ------------------------------------------------------------------------------------------------
#define BL_TW 128
#define BL_TH 32
#define BL_SD ((BL_TW + 8) / 4)  // staged dwords per row: pixels [X0-4, X0+TW+4)
#define BL_ROWS (BL_TH + 6)
__global__ __launch_bounds__(256) void k_blur(DGeom g, const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur) {
    __shared__ uint32_t s_src[BL_ROWS * BL_SD];
    __shared__ __attribute__((aligned(8))) uint16_t s_h[BL_ROWS * BL_TW];
    const int tid = threadIdx.x, f = blockIdx.y;
    int level = 0;
    for (int l = 0; l < g.nlevels; ++l)
        if ((int)blockIdx.x >= g.lv[l].blur_tile_begin) level = l;
    const DLevel &L = g.lv[level];
    const int t = blockIdx.x - L.blur_tile_begin;
    const int ty = t / L.blur_tx, tx = t - ty * L.blur_tx;
    const int X0 = tx * BL_TW, Y0 = ty * BL_TH;
    const uint8_t *img = pyr + (long long)f * g.pyr_bytes + L.off;
    // ---- stage (TH+6) rows of (TW+8) pixels as aligned dwords; reflect-101 only on tiles touching the image edge
    for (int i = tid; i < BL_ROWS * BL_SD; i += 256) {
        const int r = i / BL_SD, d = i - r * BL_SD;
        const int sy = orbx_reflect101(Y0 + r - 3, L.ph);
        const int x = X0 - 4 + 4 * d;
        const uint8_t *row = img + (long long)sy * L.pitch;
        uint32_t v;
        if (x >= 0 && x + 3 < L.pw) {
            v = *(const uint32_t *)(row + x);
        } else {
            v = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) v |= (uint32_t)row[orbx_reflect101(x + k, L.pw)] << (8 * k);
        }
        s_src[i] = v;
    }
    __syncthreads();
    // ---- row pass: kernel {18,34,49,55,49,34,18} (float Gaussian * 256, rounded; sums to 257); 4 pixels per item
    for (int i = tid; i < BL_ROWS * (BL_TW / 4); i += 256) {
        const int r = i >> 5, q = i & 31;
        const uint32_t *w = s_src + r * BL_SD + q;
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
        int b[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            b[k] = (w0 >> (8 * k)) & 0xff; b[4 + k] = (w1 >> (8 * k)) & 0xff; b[8 + k] = (w2 >> (8 * k)) & 0xff;
        }
        uint32_t h[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            h[j] = 18 * (b[j + 1] + b[j + 7]) + 34 * (b[j + 2] + b[j + 6]) + 49 * (b[j + 3] + b[j + 5]) + 55 * b[j + 4];
        uint2 o;
        o.x = h[0] | (h[1] << 16);  // each <= 255 * 257 = 65535
        o.y = h[2] | (h[3] << 16);
        *(uint2 *)(s_h + r * BL_TW + 4 * q) = o;
    }
    __syncthreads();
    // ---- column pass: thread = 4 columns x 4 rows (10 staged rows), one dword store per output row
    const int cg = tid & 31, rg = tid >> 5;
    const int X = X0 + 4 * cg;
    if (X >= L.pw) return;
    const int wv = L.pw & ~3;
    const float k0 = 55.f / 65536.f, k1 = 49.f / 65536.f, k2 = 34.f / 65536.f, k3 = 18.f / 65536.f;
    int hh[10][4];
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint2 v = *(const uint2 *)(s_h + (rg * 4 + r) * BL_TW + 4 * cg);
        hh[r][0] = v.x & 0xffff; hh[r][1] = v.x >> 16; hh[r][2] = v.y & 0xffff; hh[r][3] = v.y >> 16;
    }
    uint8_t *out = blur + (long long)f * g.pyr_bytes + L.off;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int Y = Y0 + rg * 4 + j;
        if (Y >= L.ph) break;
        uint32_t outv = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r0 = hh[j + 3][i], r1 = hh[j + 2][i] + hh[j + 4][i], r2 = hh[j + 1][i] + hh[j + 5][i],
                      r3 = hh[j][i] + hh[j + 6][i];
            int o;
            if (X + i < wv) {
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
        *(uint32_t *)(out + (long long)Y * L.pitch + X) = outv;
    }
}

// ------------------------------------------------------------------------------------------------
// K6: steered BRIEF-256 + keypoint assembly (reference computeOrbDescriptor :177-254, operator() :2049-2082).
// One wave per keypoint slot; 4 rounds x 64 lanes, one test pair per lane; __ballot packs 64 descriptor
// bits per round in the reference's bit order (bit i of byte j = pair 8j+i  ==  little-endian u64 words).
// ------------------------------------------------------------------------------------------------
// pattern rearranged per lane: entry l holds, for round r = 0..3, the pair r*64 + l as (x1, y1, x2, y2) int8
__constant__ int4 c_pattern_lane[64];
// IC_Angle weights (orbx_upload_pattern): lane = (disc row v + 15) * 2 + half; per lane the 5 aligned dwords of the LDS patch
// row it covers, as byte weights for v_dot4_u32_u8: [0..4] = u + 16 inside the disc (|u| <= umax[|v|]) else 0, [5..9] = 1
// inside the disc else 0.  m10 = sum(u * I) = dot(I, u + 16) - 16 * dot(I, 1), m01 = v * dot(I, 1): exact integers.
__constant__ uint32_t c_orient_w[64][12];
// B operands of the row pass on the matrix pipe (DS_MFMA): the banded 7-tap kernel, output column tile j of 16:
// entry [j][lane][.] = the 16 bytes B[k = 16 (lane >> 4) + s][n = lane & 15] = K7[k - (16 j + n)] (0 outside the band)
__constant__ uint32_t c_blur_b[3][64][4];

// The 7x7 Gaussian (reference :2039-2047) is fused in: only a 43x43 neighbourhood of each keypoint is ever sampled
// (|tap| <= 18, +3 px filter support), so each wave stages that patch of the UN-blurred level in LDS, runs the
// fixed-point row pass over it (4 pixels per item, as k_blur does) and evaluates the column pass only at the 512 tap
// positions.  The blurred image is never written: 2P bytes of HBM traffic per frame disappear.  Arithmetic is the
// same as k_blur's (8-bit kernel, float column path for x < (w & ~3), integer tail), so descriptors are unchanged.
#ifndef DS_ORDER
#define DS_ORDER 1
#endif
#define DS_R 21                 // patch radius: 18 (taps) + 3 (filter support)
#define DS_W (2 * DS_R + 1)     // 43
#define DS_PP 44                // LDS patch pitch in bytes (11 dwords; rows start dword-aligned in LDS)
#define DS_HC 40                // row-pass outputs per row (37 needed, computed in groups of 4)   [DS_COLFIRST == 0]
#ifndef DS_COLFIRST
#define DS_COLFIRST 1
#endif
// DS_COLFIRST: the separable Gaussian with the passes in the OTHER order.  The total I = sum_ij k_i k_j p[y+j][x+i] is one exact
// integer whichever pass runs first (every partial sum <= 255 * 257 = 65535), so the result is the same bit for bit -- but with
// the COLUMN pass done over the whole patch (v[r][c] = sum_j k_j p[r+j][c], u16), the seven values a tap needs are horizontal
// neighbours: 14 contiguous bytes = two ds_read2_b32 from a dword-aligned address instead of seven ds_read_u16 at a 80-byte
// stride, folded by four v_dot2_u32_u16 against weights chosen by the parity of the tap's column.  The tap reads are what the
// LDS pipe of this kernel is busy with (56 per lane, 60 % of its active cycles bank conflicts).
#ifndef DS_ALIAS
#define DS_ALIAS 1
#endif
// DS_MFMA (with DS_COLFIRST's tap code): the first pass is the ROW pass again, but on the matrix pipe, which this kernel
// leaves idle while its vector pipe is the busiest port: H = (P - 128) x B + 128 * 257 with P the patch rows as int8 (x ^ 0x80),
// B the banded kernel (c_blur_b), as nine v_mfma_i32_16x16x64_i8 (3 row tiles x 3 column tiles, K = 64 patch columns at once),
// exact in int32.  The accumulator layout hands every lane FOUR CONSECUTIVE ROWS of one column, so the results are stored
// transposed for free -- HT[c][r], one ds_write_b64 per tile -- and the seven values a tap needs (vertical neighbours) are
// contiguous again: the same 16-byte window + four v_dot2 as the column-first form.  ~45 vector instructions instead of ~150.
#ifndef DS_MFMA
#define DS_MFMA 1
#endif
#define DS_VC 44                // columns of v (43 needed: taps -18..18, +-3)
#define DS_VR 40                // rows of v computed (37 needed: five lane groups of 8)
#ifndef DS_WPB
#define DS_WPB 1   // waves (= keypoints) per block.  Nothing is shared between the waves of a block; one-wave blocks let the
                   // dispatcher place every wave as soon as any SIMD has room: 282 us against 299 (4 waves) and 372 (8)
#endif
// (Two or four consecutive keypoints per wave through the same LDS buffers, without prefetch, measured 314 / 319 us against
// 282 us in round 2: the kernel is not bound by the launch rate of its one-wave workgroups.)
// (Round 2 also rebuilt the column pass around the LDS counters -- the LDS pipe is active 87 % of this kernel, 60 % of that
// bank conflicts of the 56 ds_read_u16 per lane at the rotated tap positions: row-pass output stored TRANSPOSED in two
// copies (A[c][r] = h[r][c], B[c][r] = h[r + 2][c]) so that a tap's seven vertical neighbours always sit in one 8-byte
// aligned 16-byte window, fetched by ONE ds_read2_b64 and folded by v_alignbit + four v_dot2_u32_u16; the patch aliased
// into copy B.  Bit-exact, 8 instead of 56 LDS reads per lane -- and 324 us against 281 us: the transposed stores and the
// window addressing cost ~95 more vector instructions per keypoint, and with both pipes near saturation it is the vector
// pipe that sets the time.  Not kept.)
#ifndef DS_WPS
#define DS_WPS 7   // waves per SIMD the register allocation must allow (LDS admits 7 blocks of 4 waves per CU)
#endif
// (Several keypoints per wave with register prefetch of the next patch was measured: 430-440 us at the 4-5 waves per
// SIMD its registers allow against 375 us for one keypoint per wave -- every phase of this kernel is a dependent chain of
// LDS round trips, and resident waves are what hides them.)
// FPM (fp_mode) is a template constant: as a run-time value it costs a scalar branch and both code paths in each of the 8 taps
typedef const __attribute__((address_space(3))) uint16_t *orbx_lds_u16p;
typedef const __attribute__((address_space(3))) uint32_t *orbx_lds_u32p;
// I = sum_i k_i v[r][c + i], i = 0..6, for the u16 element at LDS byte address `adr` (DS_COLFIRST): the 8 elements of the
// dword-aligned 16-byte window around them, against the weights shifted by the parity of c
__device__ __forceinline__ uint32_t orbx_ds_tap(uint32_t adr) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const bool odd = (adr & 2u) != 0;
    const orbx_lds_u32p wp = (orbx_lds_u32p)(uintptr_t)(adr & ~3u);
    const uint32_t d0 = wp[0], d1 = wp[1], d2 = wp[2], d3 = wp[3];
    const uint32_t w0 = odd ? (18u << 16) : (18u | (34u << 16)), w1 = odd ? (34u | (49u << 16)) : (49u | (55u << 16)),
                   w2 = odd ? (55u | (49u << 16)) : (49u | (34u << 16)), w3 = odd ? (34u | (18u << 16)) : 18u;
    uint32_t I = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, d0), __builtin_bit_cast(u16x2, w0), 0u, false);
    I = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, d1), __builtin_bit_cast(u16x2, w1), I, false);
    I = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, d2), __builtin_bit_cast(u16x2, w2), I, false);
    I = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, d3), __builtin_bit_cast(u16x2, w3), I, false);
    return I;   // < 2^24.01
}
template <int FPM>
__global__ __launch_bounds__(64 * DS_WPB, DS_WPS) void k_describe(DGeom g, const uint8_t *__restrict__ pyr,
                                                  const uint32_t *__restrict__ lvl_kp,
                                                  const int *__restrict__ lvl_count,
                                                  float *__restrict__ lvl_angle, orbx_keypoint *__restrict__ kps,
                                                  uint8_t *__restrict__ desc, int *__restrict__ counts,
                                                  int *__restrict__ status, int cap, int dbg_stop, int nframes) {
    // dbg_stop (ORBX_DESC_STOP, phase-timing builds only, -DORBX_TIMING_KNOBS): 1 = after staging, 2 = after orientation,
    // 3 = after the row pass.  The shipped library pins it to 0.
#ifndef ORBX_TIMING_KNOBS
    dbg_stop = 0;
#endif
#if DS_COLFIRST && DS_ALIAS
    // v OVER the patch: the column pass holds every patch row it needs in registers before its first store (a wave's LDS
    // instructions execute in order, and one wave owns the buffers), and nothing reads the patch afterwards -- 3.5 KB of LDS per
    // wave instead of 5.4
    __shared__ __attribute__((aligned(16))) uint16_t s_h[DS_WPB][DS_VR * DS_VC];
#else
    __shared__ uint32_t s_patch[DS_WPB][DS_W * DS_PP / 4 + 4];
    __shared__ __attribute__((aligned(16))) uint16_t s_h[DS_WPB][DS_COLFIRST ? DS_VR * DS_VC : DS_W * DS_HC];
#endif
    // one wave per keypoint, waves indexed by dense OUTPUT position (level-major order of operator(), :2066-2082).
    // The wave index is wave-uniform (which the compiler cannot see): with it scalar, the level search and the position
    // load run on the scalar unit.
    const int lane = threadIdx.x & 63, wv_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Workgroup order (x fastest): frame mod 8, then keypoint, then frame / 8.  Workgroups are dealt round-robin over the 8
    // XCDs, so XCD x works through the keypoints of frame 8k + x one after the other with that frame's 1.2 MB pyramid in its
    // own 4 MB L2, and the chip holds 8 frames at a time.  (With the frame as the fastest index -- the order of the first
    // version -- every frame of the batch has a few keypoints in flight at once: at 1024 frames per launch no cache holds that,
    // and every patch came from HBM, 3.1 MB per frame.)
#if DS_ORDER == 1
    const int f = blockIdx.y * 8 + (blockIdx.x & 7);
    const int oi = (blockIdx.x >> 3) * DS_WPB + wv_id;
    if (f >= nframes) return;
#else
    const int f = blockIdx.x;   // frame fastest: one frame's patches stay in one XCD's L2
    const int oi = blockIdx.y * DS_WPB + wv_id;
#endif
    const int *lc = lvl_count + f * g.nlevels;
    int total = 0, level = 0, slot = oi;
#ifndef DS_LEVELS8
#define DS_LEVELS8 1
#endif
    if (DS_LEVELS8 && g.nlevels == 8) {
        // every ORB-SLAM2 configuration: the eight counts in one scalar load, then compares and selects only, and ONE table
        // load for the level found (the general loop below pays two dependent scalar loads and two branches per level:
        // ~120 scalar instructions and eight memory round trips in front of every keypoint)
        const int4 ca = *(const int4 *)lc, cb = *(const int4 *)(lc + 4);
        const int cnt[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
        int base = 0;
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            if (oi >= total) { level = l; base = total; }
            total += cnt[l];
        }
        slot = g.lv[level].kp_begin + (oi - base);
    } else {
#pragma unroll
    for (int l = 0; l < ORBX_MAX_LEVELS; ++l) {
        if (l < g.nlevels) {
            if (oi >= total) { level = l; slot = g.lv[l].kp_begin + (oi - total); }
            total += lc[l];
        }
    }
    }
    if (oi == 0 && lane == 0) {
        counts[f] = min(total, cap);
        if (total > cap) atomicMax(&status[f], (int)ORBX_CAPACITY);
    }
    if (oi >= min(total, cap) || dbg_stop == 4) return;
#if DS_COLFIRST && DS_ALIAS
    uint32_t *patch = (uint32_t *)s_h[wv_id];
#else
    uint32_t *patch = s_patch[wv_id];
#endif
    uint16_t *hrow = s_h[wv_id];
    const uint32_t pos = lvl_kp[(long long)f * g.kp_total + slot];
    const int4 pat = c_pattern_lane[lane];
    const DLevel &L = g.lv[level];
    const int x = (int)(pos & 0xfff) + (ORBX_EDGE - 3), y = (int)((pos >> 12) & 0xfff) + (ORBX_EDGE - 3);
    const uint8_t *img = pyr + (long long)f * g.pyr_bytes + L.off;
    const int px0 = x - DS_R, py0 = y - DS_R;
    const int xa = px0 & ~3;
    const bool interior = px0 >= 0 && py0 >= 0 && xa + 48 <= L.pitch && x + DS_R < L.pw && y + DS_R < L.ph;
    {
        // ---- stage the 43x43 patch (rows y-21.., columns x-21..): 12 ALIGNED dwords cover the 44 bytes of a patch
        // row; dword d of the LDS row = funnel shift of aligned dwords d, d+1.  A lane loads 3 aligned dwords (one
        // 12-byte load: a third of the load instructions of a dword per lane, the texture-address unit's rate is per
        // lane, not per byte) and takes the 4th from the lane above (one DPP move): 4 lanes per row, 16 rows per step,
        // every load in flight before the first LDS write.
        if (interior) {
            const int dq = lane & 3, rq = lane >> 2;
            const uint32_t shift = (uint32_t)(px0 & 3);
            const uint8_t *p0 = img + (__mul24(py0, L.pitch) + xa + 12 * dq);   // 32-bit offsets: no 64-bit multiplies
            orbx_uint3_u tv[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) tv[k] = *(const orbx_uint3_u *)(p0 + __mul24(min(16 * k + rq, DS_W - 1), L.pitch));
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t nxt = orbx_lane_above(tv[k].x);
                const int r = 16 * k + rq;
                if (r < DS_W) {
                    uint32_t *d = patch + r * (DS_PP / 4) + 3 * dq;
                    d[0] = __builtin_amdgcn_alignbyte(tv[k].y, tv[k].x, shift);
                    d[1] = __builtin_amdgcn_alignbyte(tv[k].z, tv[k].y, shift);
                    if (dq < 3) d[2] = __builtin_amdgcn_alignbyte(nxt, tv[k].z, shift);   // dword 11 of a row does not exist
                }
            }
        } else {
            // image edge: reflect-101 of the padded level, byte by byte
            uint8_t *pb = (uint8_t *)patch;
            for (int i = lane; i < DS_W * DS_W; i += 64) {
                const int r = i / DS_W, c = i - r * DS_W;
                const int sy = orbx_reflect101(py0 + r, L.ph), sx = orbx_reflect101(px0 + c, L.pw);
                pb[r * DS_PP + c] = img[(long long)sy * L.pitch + sx];
            }
        }
        orbx_wave_sync();
        if (dbg_stop == 1) return;
        // ---- orientation (IC_Angle, reference src/ORBextractor.cc:104-161) from the staged UN-blurred patch: lanes 0-31
        // take the row +v, lanes 32-63 the row -v; integer moments reduced across the wave; fastAtan2 on every lane
        float angle_deg;
        {
            // lane = (row, half): 5 aligned dwords of patch row 21 + v, two v_dot4_u32_u8 each against the constant weights
            const int row = lane >> 1, half = lane & 1;                 // row = v + 15 (lanes 62, 63 carry zero weights)
            const uint32_t *pr = patch + min(row + (DS_R - ORBX_HALF_PATCH), DS_W - 1) * (DS_PP / 4) + 1 + 5 * half;
            const uint4 wa = *(const uint4 *)&c_orient_w[lane][0], wb = *(const uint4 *)&c_orient_w[lane][4], wc = *(const uint4 *)&c_orient_w[lane][8];
            const uint32_t p0 = pr[0], p1 = pr[1], p2 = pr[2], p3 = pr[3], p4 = pr[4];
            uint32_t A = __builtin_amdgcn_udot4(p0, wa.x, 0u, false), S = __builtin_amdgcn_udot4(p0, wb.y, 0u, false);
            A = __builtin_amdgcn_udot4(p1, wa.y, A, false); S = __builtin_amdgcn_udot4(p1, wb.z, S, false);
            A = __builtin_amdgcn_udot4(p2, wa.z, A, false); S = __builtin_amdgcn_udot4(p2, wb.w, S, false);
            A = __builtin_amdgcn_udot4(p3, wa.w, A, false); S = __builtin_amdgcn_udot4(p3, wc.x, S, false);
            A = __builtin_amdgcn_udot4(p4, wb.x, A, false); S = __builtin_amdgcn_udot4(p4, wc.y, S, false);
            int m10 = (int)A - 16 * (int)S;
            int m01 = __mul24(row - ORBX_HALF_PATCH, (int)S);
            m10 = orbx_wave_sum(m10);
            m01 = orbx_wave_sum(m01);
            angle_deg = orbx_fast_atan2((float)m01, (float)m10);
            if (lane == 0) lvl_angle[(long long)f * g.kp_total + slot] = angle_deg;
        }
        if (dbg_stop == 2) return;
        // ---- row pass: h[r][c] for patch columns c+3 (c = 0..36 used); 4 outputs per item from 3 aligned dwords via
        // v_alignbyte + v_dot4_u32_u8 with the packed 8-bit kernel {18,34,49,55 | 49,34,18,0}
#if DS_COLFIRST && DS_MFMA
        {
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            const int m = lane & 15, gq = lane >> 4;
            i32x4 A[3], Bm[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {   // A[t]: patch row 16 t + m, bytes 16 gq .. + 15 (rows >= 43 / bytes >= 43: don't-care, zero weights or unused rows)
                const uint32_t *ap = patch + (16 * t + m) * (DS_PP / 4) + 4 * gq;
                A[t][0] = (int)(ap[0] ^ 0x80808080u); A[t][1] = (int)(ap[1] ^ 0x80808080u);
                A[t][2] = (int)(ap[2] ^ 0x80808080u); A[t][3] = (int)(ap[3] ^ 0x80808080u);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const uint4 b4 = *(const uint4 *)&c_blur_b[j][lane][0];
                Bm[j][0] = (int)b4.x; Bm[j][1] = (int)b4.y; Bm[j][2] = (int)b4.z; Bm[j][3] = (int)b4.w;
            }
            const i32x4 c0 = {128 * 257, 128 * 257, 128 * 257, 128 * 257};
            uint32_t *vo = (uint32_t *)hrow + m * (DS_VC / 2) + 2 * gq;   // HT[c = 16 j + m][r = 16 t + 4 gq ..]: u16 index c * 44 + r
#pragma unroll
            for (int j = 0; j < 3; ++j) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const i32x4 acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t], Bm[j], c0, 0, 0, 0);
                    // columns c < 37 and rows r < 44 exist in HT
                    if ((j < 2 || m < DS_W - 6 - 32) && (t < 2 || gq < 3))
                        *(uint2 *)(vo + (16 * j) * (DS_VC / 2) + 8 * t) =
                            make_uint2((uint32_t)acc[0] | ((uint32_t)acc[1] << 16), (uint32_t)acc[2] | ((uint32_t)acc[3] << 16));
                }
            }
        }
#elif DS_COLFIRST
        // ---- column pass over the whole patch: lane = (dword column cg of 11, row group seg of 5), 8 output rows from 14 patch
        // rows held in registers as u16 pairs (v_perm), the filter as packed 16-bit multiply-adds (sums <= 65535)
        {
            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
            const int seg = (lane * 373) >> 12, cg = lane - 11 * seg;   // lane / 11, lane % 11
            if (seg < 5) {
                // (the symmetric sums and the centre product as plain 32-bit operations on the pair -- no carry crosses the halves:
                // <= 510 and <= 14025 -- which issue at the full rate; the three multiply-adds of the sums are packed 16-bit ones)
                uint32_t pa[14], pb[14];
#pragma unroll
                for (int j = 0; j < 14; ++j) {
                    const uint32_t d = patch[min(8 * seg + j, DS_W - 1) * (DS_PP / 4) + cg];
                    pa[j] = __builtin_amdgcn_perm(0u, d, 0x0c010c00u);   // columns 4cg, 4cg+1 as u16 pair
                    pb[j] = __builtin_amdgcn_perm(0u, d, 0x0c030c02u);   // columns 4cg+2, 4cg+3
                }
                const u16x2 k1 = {49, 49}, k2 = {34, 34}, k3 = {18, 18};
                uint32_t *vo = (uint32_t *)hrow + (8 * seg) * (DS_VC / 2) + 2 * cg;
#define DS_CP(P, r)                                                                                                          \
    __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, P[r + 2] + P[r + 4]) * k1 +                                       \
                                     (__builtin_bit_cast(u16x2, P[r + 1] + P[r + 5]) * k2 +                                  \
                                      (__builtin_bit_cast(u16x2, P[r] + P[r + 6]) * k3 + __builtin_bit_cast(u16x2, __umul24(P[r + 3], 55u)))))
#pragma unroll
                for (int r = 0; r < 8; ++r) *(uint2 *)(vo + r * (DS_VC / 2)) = make_uint2(DS_CP(pa, r), DS_CP(pb, r));
#undef DS_CP
            }
        }
#else
        for (int i = lane; i < DS_W * (DS_HC / 8); i += 64) {
            // item = 8 consecutive outputs of one row from 4 aligned dwords: the byte-shifted middle dwords are shared
            // between the two groups of four (9 v_alignbyte + 16 v_dot4 per 8 outputs)
            const int r = __mul24(i, 13108) >> 16, q8 = i - r * (DS_HC / 8);   // i / 5 for i < 16384
            const uint32_t *w = patch + r * (DS_PP / 4) + 2 * q8;
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
            const uint32_t KLO = 18u | (34u << 8) | (49u << 16) | (55u << 24), KHI = 49u | (34u << 8) | (18u << 16);
            uint32_t hv[8];
            hv[0] = __builtin_amdgcn_udot4(w0, KLO, __builtin_amdgcn_udot4(w1, KHI, 0u, false), false);
            hv[4] = __builtin_amdgcn_udot4(w1, KLO, __builtin_amdgcn_udot4(w2, KHI, 0u, false), false);
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                const uint32_t a01 = __builtin_amdgcn_alignbyte(w1, w0, j), a12 = __builtin_amdgcn_alignbyte(w2, w1, j),
                               a23 = __builtin_amdgcn_alignbyte(w3, w2, j);
                hv[j] = __builtin_amdgcn_udot4(a01, KLO, __builtin_amdgcn_udot4(a12, KHI, 0u, false), false);
                hv[4 + j] = __builtin_amdgcn_udot4(a12, KLO, __builtin_amdgcn_udot4(a23, KHI, 0u, false), false);
            }
            uint4 o;
            o.x = hv[0] | (hv[1] << 16);   // each <= 255 * 257 = 65535
            o.y = hv[2] | (hv[3] << 16);
            o.z = hv[4] | (hv[5] << 16);
            o.w = hv[6] | (hv[7] << 16);
            *(uint4 *)(hrow + r * DS_HC + 8 * q8) = o;
        }
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (dbg_stop == 3) return;
        // ---- taps
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        const float angle = angle_deg * factorPI;
                const OrbxSinCos sc = orbx_sincosf_pinned(angle);
        const float a = sc.c, b = sc.s;
        // Column pass at the tap position.  OpenCV's SSE2 column filter accumulates ((r0*k0 + r1*k1) + r2*k2) + r3*k3 in
        // float with k = {55,49,34,18}/65536 and rounds to nearest-even for the columns x < (w & ~3); the last w & 3
        // columns take its integer tail (I + 2^15) >> 16.  Every float product and every partial sum below 2^24 units of
        // 2^-16 is exact, and a total >= 2^24 saturates to 255 either way, so the float path IS round-half-even(I / 65536)
        // with I = 55 r0 + 49 r1 + 34 r2 + 18 r3: one integer formula serves both, the tail only changes the tie rule.
        const int wvec = L.pw & ~3;
        const int pw4[4] = {pat.x, pat.y, pat.z, pat.w};
        unsigned long long words[4];
        if (x + (DS_R - 3) < wvec) {
            // No tap of this keypoint reaches the tail columns (all but keypoints within 18 px of the right edge): the
            // float form of round-half-even is the shortest.  cvRound of the rotated coordinates = adding 1.5 * 2^23
            // (round-to-nearest-even in the add, the integer lands in the low mantissa bits); the 24-bit multiply-add
            // takes those low bits as they are, the exponent bits of the column term cancel in the constant.
#if DS_COLFIRST
            const uint32_t cbias = (uint32_t)(2 * ((DS_R - 3) * DS_VC + (DS_R - 3))) - 0x400000u * (2u * DS_VC) - (0x4B400000u << 1);
#else
            const uint32_t cbias = (uint32_t)(2 * ((DS_R - 3) * DS_HC + (DS_R - 3))) - 0x400000u * (2u * DS_HC) - (0x4B400000u << 1);
#endif
            const uint32_t hbase = (uint32_t)(uintptr_t)(orbx_lds_u16p)hrow + cbias;
            float tv[8];
            float nb = -b;
            asm("" : "+v"(nb));   // a register of its own (the compiler would fold the negation back into every product)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const float px = (float)(signed char)((pw4[r] >> (16 * s2)) & 0xff);
                    const float py = (float)(signed char)((pw4[r] >> (16 * s2 + 8)) & 0xff);
                    float fy, fx;
                    if (FPM == ORBX_FP_GCC_FMA) {
                        fy = __builtin_fmaf(px, b, py * a);     // vfmadd132ss: x*b + rn(y*a)
                        fx = __builtin_fmaf(px, a, py * nb);    // vfmsub132ss: x*a - rn(y*b); -(y*b) = y*(-b) exactly, and a
                                                                // product by a NEGATED REGISTER is a VOP3 encoding (half rate)
                    } else {
                        fy = px * b + py * a;
                        fx = px * a + py * nb;
                    }
                    const uint32_t uy = __float_as_uint(fy + 12582912.0f), ux = __float_as_uint(fx + 12582912.0f);
#if DS_COLFIRST
#if DS_MFMA
                    // LDS byte address of HT[ix+18][iy+18]: the tap's window starts here (two instructions: v_lshl_add + v_mad_u32_u24)
                    uint32_t ay = (uy << 1) + hbase;
                    asm("" : "+v"(ay));
                    const uint32_t adr = __umul24(ux, 2u * DS_VC) + ay;
#else
                    const uint32_t adr = __umul24(uy, 2u * DS_VC) + ((ux << 1) + hbase);   // LDS byte address of v[iy+18][ix+18]: the tap's window starts here
#endif
                    const uint32_t I = orbx_ds_tap(adr);
#else
                    uint32_t adr = __umul24(uy, 2u * DS_HC) + ((ux << 1) + hbase);   // LDS byte address of h[iy+18][ix+18]
                    asm("" : "+v"(adr));   // a plain address from here on: the row offsets fold into the ds_read immediates
                    const orbx_lds_u16p hp = (orbx_lds_u16p)(uintptr_t)adr;
                    const uint32_t r0 = hp[3 * DS_HC], r1 = (uint32_t)hp[2 * DS_HC] + hp[4 * DS_HC],
                                   r2 = (uint32_t)hp[1 * DS_HC] + hp[5 * DS_HC], r3 = (uint32_t)hp[0] + hp[6 * DS_HC];
                    const uint32_t I = __umul24(55u, r0) + __umul24(49u, r1) + __umul24(34u, r2) + __umul24(18u, r3);   // < 2^24.01
#endif
                    // I <= 2^24 converts exactly and I * 2^-16 is exact; above, the conversion's own rounding keeps it >= 256
                    tv[2 * r + s2] = __builtin_fminf(__builtin_rintf((float)I * (1.f / 65536.f)), 255.f);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) words[r] = orbx_ballot(tv[2 * r] < tv[2 * r + 1]);
        } else {
            int tval[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const float px = (float)(signed char)((pw4[r] >> (16 * s2)) & 0xff);
                    const float py = (float)(signed char)((pw4[r] >> (16 * s2 + 8)) & 0xff);
                    float fy, fx;
                    if (FPM == ORBX_FP_GCC_FMA) {
                        fy = __builtin_fmaf(px, b, py * a);
                        fx = __builtin_fmaf(px, a, -(py * b));
                    } else {
                        fy = px * b + py * a;
                        fx = px * a - py * b;
                    }
                    const int iy = (int)__builtin_rintf(fy), ix = (int)__builtin_rintf(fx);
                    // blurred pixel (x+ix, y+iy): column pass over h rows (iy+21-3 .. iy+21+3), h column ix+18
#if DS_COLFIRST
#if DS_MFMA
                    const uint32_t I = orbx_ds_tap((uint32_t)(uintptr_t)(orbx_lds_u16p)hrow + 2u * (uint32_t)(__mul24(ix + DS_R - 3, DS_VC) + (iy + DS_R - 3)));
#else
                    const uint32_t I = orbx_ds_tap((uint32_t)(uintptr_t)(orbx_lds_u16p)hrow + 2u * (uint32_t)(__mul24(iy + DS_R - 3, DS_VC) + (ix + DS_R - 3)));
#endif
#else
                    const uint16_t *hp = hrow + __mul24(iy + DS_R - 3, DS_HC) + (ix + DS_R - 3);
                    const uint32_t r0 = hp[3 * DS_HC], r1 = (uint32_t)hp[2 * DS_HC] + hp[4 * DS_HC],
                                   r2 = (uint32_t)hp[1 * DS_HC] + hp[5 * DS_HC], r3 = (uint32_t)hp[0] + hp[6 * DS_HC];
                    const uint32_t I = __umul24(55u, r0) + __umul24(49u, r1) + __umul24(34u, r2) + __umul24(18u, r3);
#endif
                    const uint32_t tie = x + ix >= wvec ? 1u : ((I >> 16) & 1u);   // half-up in the tail, half-even elsewhere
                    tval[2 * r + s2] = (int)min((I + 0x7fffu + tie) >> 16, 255u);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) words[r] = orbx_ballot(tval[2 * r] < tval[2 * r + 1]);
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
}

// ------------------------------------------------------------------------------------------------
// K7: brute-force Hamming best / second-best (DescriptorDistance src/ORBmatcher.cc:2073-2093 for every
// pair; bookkeeping of the search loops, e.g. :627-640) on the matrix cores.
//
// nq x nt x 256 bit operations per pair is GEMM-shaped and compute-bound (16 MB of descriptors against 65 G bit
// products per 256-frame step), so the distances come from v_mfma_i32_32x32x32_i8:
//     dist(q, t) = popc(q) + popc(t) - 2 * <q, t>           (<.,.> = dot product of the two bit vectors)
//   * A operand = 32 TRAIN descriptors of a tile, bit p of their 32 bytes expanded to bytes 0 / 64 (shift + and per
//     dword; K-step p = bit p of every byte: the K order is irrelevant as long as both operands use the same one);
//   * B operand = 32 QUERY descriptors, expanded ONCE per wave to bytes 0 / -128 and kept in registers (two query tiles
//     per wave: 64 VGPRs), so that every product is 0 or -8192 = -2 << 12;
//   * the accumulator of a tile is not cleared but LOADED with key(t) = (popc(t) + 256) << 12 | index-in-chunk from an
//     LDS table (the C layout puts 4 consecutive train rows in 4 consecutive registers: one ds_read_b128 each), so
//     eight MFMAs leave  key = (dist - popc(q) + 256) << 12 | index  in every accumulator register -- ordered like
//     (dist, index) for the lane's query -- and the bookkeeping is two instructions per distance: second =
//     med3(key, best, second), best = min(best, key).  min(key) is the best match with the lowest index on ties and the
//     second-smallest key carries the second-best distance counting duplicates, exactly the reference loops.
// Measured on 256 x 1000 x 1000 (tools/match_rate.py): 60 us; without the bookkeeping 48 us, without the expansion of the
// train tiles 56 us, MFMAs + loads alone 49 us (= 2.7 Pop/s: the matrix pipe at the clock the chip holds under this load).
// The same kernel on v_mfma_i32_16x16x64_i8 (four 16-column query tiles per wave, K step = two bit planes) is bit-exact
// and slower, 71 us: twice the operand traffic per product and rotates instead of shifts in the expansion.
// 174 -> 60-65 us per 256 x 1000 x 1000 against k_match_valu below (xor + popcount on the vector pipe), which north_star
// names as the form of this kernel ("no MFMA -- this is integer/bitwise work"): the premise does not hold for the matcher
// -- the result is the same integer, bit for bit -- so the MFMA kernel is the default and the vector-pipe kernel stays
// selectable (ORBX_MATCH_KERNEL=valu) and parity-tested, for A/B runs and for readers who want the path as specified.
// ------------------------------------------------------------------------------------------------
#define MT_SPLIT 16     // most ways the train set is split over blockIdx.y; partials merged by k_match_merge.  The launcher picks
                        // the smallest split that still fills the chip (every extra split repeats the query expansion and
                        // one partial record per query)
#define MT_WAVES 4      // waves per block; they share the train key table
// Query column tiles (of 32) per wave, QT: the expanded train tile (A operand: 32 shift-and pairs) and its key registers serve QT
// MFMA chains.  QT = 2: 64 B-operand registers, three waves per SIMD without a spill (round 2 capped the registers for four
// and spilled 11 of them inside the MFMA loop).  QT = 4: a block walks its train tiles ONCE for 512 queries -- half the train
// loads, half the expansions and half the key-table reads per distance -- with 128 B-operand + 64 accumulator registers, two
// waves per SIMD; the launcher takes it when the launch still fills the chip that way.
#define MT_QPW(QT) (32 * (QT))
#define MT_QPB(QT) (MT_WAVES * MT_QPW(QT))
#define MT_CHUNK 4096   // train descriptors per key table (12 index bits in the key)
typedef int mt_v4i __attribute__((ext_vector_type(4)));
typedef int mt_v16i __attribute__((ext_vector_type(16)));
__device__ __forceinline__ uint32_t mt_min2(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t mt_max2(uint32_t a, uint32_t b) { return a > b ? a : b; }
template <int QT>
__global__ __launch_bounds__(64 * MT_WAVES, QT == 2 ? 3 : 2) void k_match(const uint8_t *__restrict__ q, const int *__restrict__ nq,
                                                         long long q_stride, const uint8_t *__restrict__ t,
                                                         const int *__restrict__ nt, long long t_stride,
                                                         uint2 *__restrict__ partial, int *__restrict__ best_idx,
                                                         int *__restrict__ best_dist, int *__restrict__ second_dist,
                                                         int out_stride, int nsplit) {
    __shared__ __attribute__((aligned(16))) int s_tk[MT_CHUNK];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), pr = blockIdx.z, sp = blockIdx.y;
    const int NQ = min(nq[pr], out_stride), NT = min(nt[pr], 1 << 20);   // contract (orbx.h): counts beyond out_stride are ignored
    if ((int)blockIdx.x * MT_QPB(QT) >= NQ) return;
    const int r = lane & 31, h = lane >> 5;
    const uint8_t *qp = q + (long long)pr * q_stride;
    const uint8_t *tp = t + (long long)pr * t_stride;
    // this block's share of the train set, in whole tiles of 32
    const int per = ((NT + 31) / 32 + nsplit - 1) / nsplit * 32;
    const int j0 = min(NT, sp * per), j1 = min(NT, j0 + per);
    // ---- queries: lane (r, h) holds bytes 16h .. 16h+15 of query r of each tile, expanded bit plane by bit plane
    const int qw = blockIdx.x * MT_QPB(QT) + w * MT_QPW(QT);
    const bool wave_on = qw < NQ;
    mt_v4i bq[QT][8];
    int pq[QT];
#pragma unroll
    for (int c = 0; c < QT; ++c) {
        const int qi = qw + 32 * c + r;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (qi < NQ) v = *(const uint4 *)(qp + (long long)qi * 32 + 16 * h);
        pq[c] = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
        pq[c] += __shfl_xor(pq[c], 32);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            bq[c][p][0] = (int)((v.x << (7 - p)) & 0x80808080u);
            bq[c][p][1] = (int)((v.y << (7 - p)) & 0x80808080u);
            bq[c][p][2] = (int)((v.z << (7 - p)) & 0x80808080u);
            bq[c][p][3] = (int)((v.w << (7 - p)) & 0x80808080u);
        }
    }
    // running result over the chunks, in the output format dist << 20 | index
    uint32_t gbest[QT], gsecond[QT];
#pragma unroll
    for (int c = 0; c < QT; ++c) gbest[c] = gsecond[c] = 0xffffffffu;
    for (int c0 = j0; c0 < j1; c0 += MT_CHUNK) {
        const int c1 = min(j1, c0 + MT_CHUNK), n = c1 - c0, npad = (n + 31) & ~31;
        __syncthreads();   // the previous chunk's table is no longer read
        for (int i = threadIdx.x; i < npad; i += 64 * MT_WAVES) {
            int key = 0x7fffffff;   // rows past the end: never the minimum (their descriptor is read as zero below)
            if (i < n) {
                const uint4 a = *(const uint4 *)(tp + (long long)(c0 + i) * 32), b = *(const uint4 *)(tp + (long long)(c0 + i) * 32 + 16);
                const int pt = __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);
                key = ((pt + 256) << 12) | i;
            }
            s_tk[i] = key;
        }
        __syncthreads();
        if (!wave_on) continue;
        int best[QT], second[QT];
#pragma unroll
        for (int c = 0; c < QT; ++c) best[c] = second[c] = 0x7fffffff;
        const uint8_t *trow = tp + (long long)(c0 + r) * 32 + 16 * h;
        uint4 ta = make_uint4(0, 0, 0, 0);
        if (r < n) ta = *(const uint4 *)trow;
        for (int jt = 0; jt < npad; jt += 32) {
            const uint4 tc = ta;
            // next tile's descriptors requested before this tile's MFMAs
            ta = make_uint4(0, 0, 0, 0);
            if (jt + 32 + r < n) ta = *(const uint4 *)(trow + (long long)(jt + 32) * 32);
            mt_v16i acc[QT];
            {
                const int4 *kp = (const int4 *)(s_tk + jt + 4 * h);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int4 k4 = kp[2 * g];   // rows 8g + 4h .. + 3 of the tile = registers 4g .. 4g+3
                    acc[0][4 * g] = k4.x; acc[0][4 * g + 1] = k4.y; acc[0][4 * g + 2] = k4.z; acc[0][4 * g + 3] = k4.w;
                }
#pragma unroll
                for (int c = 1; c < QT; ++c) acc[c] = acc[0];
            }
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                mt_v4i a;
                if (p < 6) {
                    a[0] = (int)((tc.x << (6 - p)) & 0x40404040u); a[1] = (int)((tc.y << (6 - p)) & 0x40404040u);
                    a[2] = (int)((tc.z << (6 - p)) & 0x40404040u); a[3] = (int)((tc.w << (6 - p)) & 0x40404040u);
                } else if (p == 6) {
                    a[0] = (int)(tc.x & 0x40404040u); a[1] = (int)(tc.y & 0x40404040u);
                    a[2] = (int)(tc.z & 0x40404040u); a[3] = (int)(tc.w & 0x40404040u);
                } else {
                    a[0] = (int)((tc.x >> 1) & 0x40404040u); a[1] = (int)((tc.y >> 1) & 0x40404040u);
                    a[2] = (int)((tc.z >> 1) & 0x40404040u); a[3] = (int)((tc.w >> 1) & 0x40404040u);
                }
#pragma unroll
                for (int c = 0; c < QT; ++c) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[c][p], acc[c], 0, 0, 0);
            }
            // best <= second always: the new second-smallest is the median of (best, second, key)  (one v_med3_i32)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
#pragma unroll
                for (int c = 0; c < QT; ++c) {
                    const int kc = acc[c][i];
                    second[c] = max(min(best[c], second[c]), min(max(best[c], second[c]), kc));
                    best[c] = min(best[c], kc);
                }
            }
        }
        // fold the chunk into the running result
#pragma unroll
        for (int c = 0; c < QT; ++c) {
            const uint32_t b = best[c] == 0x7fffffff ? 0xffffffffu
                                                     : ((uint32_t)((best[c] >> 12) + pq[c] - 256) << 20) | (uint32_t)(c0 + (best[c] & 4095));
            const uint32_t s2 = second[c] == 0x7fffffff ? 0xffffffffu
                                                        : ((uint32_t)((second[c] >> 12) + pq[c] - 256) << 20) | (uint32_t)(c0 + (second[c] & 4095));
            gsecond[c] = mt_min2(mt_min2(gsecond[c], s2), mt_max2(gbest[c], b));
            gbest[c] = mt_min2(gbest[c], b);
        }
    }
    if (!wave_on) return;
    // the two lane halves hold different train rows of the same query
#pragma unroll
    for (int c = 0; c < QT; ++c) {
        const uint32_t ob = (uint32_t)__shfl_xor((int)gbest[c], 32), os = (uint32_t)__shfl_xor((int)gsecond[c], 32);
        const uint32_t second = mt_min2(mt_min2(gsecond[c], os), mt_max2(gbest[c], ob));
        const uint32_t best = mt_min2(gbest[c], ob);
        const int qi = qw + 32 * c + r;
        if (h == 0 && qi < NQ) {
            if (nsplit > 1) {
                partial[((long long)pr * nsplit + sp) * out_stride + qi] = make_uint2(best, second);
            } else {
                const long long o = (long long)pr * out_stride + qi;
                best_idx[o] = best == 0xffffffffu ? -1 : (int)(best & 0xfffffu);
                best_dist[o] = best == 0xffffffffu ? 0x7fffffff : (int)(best >> 20);
                second_dist[o] = second == 0xffffffffu ? 0x7fffffff : (int)(second >> 20);
            }
        }
    }
}

// k_match on the FP4 rate of the matrix pipe (round 3).  The same bookkeeping, the same keys, the same result bits -- the
// products come from v_mfma_scale_f32_32x32x64_f8f6f4 with both operands in e2m1: a descriptor bit is the nibble 0b0100 (2.0)
// on the train side and 0b1110 (-4.0) on the query side, the block scale of the train operand is 2^10, so a common bit adds
// -8 * 2^10 = -2 << 12 to the accumulator, which is LOADED with the float of key(t) = (popc(t) + 256) << 12 | index.  Every
// partial sum is an integer below 2^24: exact in f32 (tools/mfma_fp4_probe.hip checks the instruction with such data, layout and
// scale bytes included).  K = 64 descriptor bits per instruction at the cycles of the 32-bit-K int8 form: four MFMAs per
// 32 x 32 distances instead of eight, and the expansion of a train tile is 28 instead of 60 vector instructions (bit 4j+k of a
// dword -> nibble j of register k: shift + and).  Positive floats order like their bit patterns, so best / second stay
// v_min_i32 / v_med3_i32 on the raw bits.
typedef int mt_v8i __attribute__((ext_vector_type(8)));
typedef float mt_v16f __attribute__((ext_vector_type(16)));
#define MT_F4_NONE 0x7f7fffff   // FLT_MAX: "no key yet" / rows past the end (a zero descriptor adds nothing to it)
#ifndef MT_F4_LB2
#define MT_F4_LB2 3
#endif
#ifndef MT_F4_LB4
#define MT_F4_LB4 2
#endif
#ifndef MT_F4_USE4
#define MT_F4_USE4 1
#endif
template <int QT>
__global__ __launch_bounds__(64 * MT_WAVES, QT == 2 ? MT_F4_LB2 : MT_F4_LB4) void k_match_f4(const uint8_t *__restrict__ q, const int *__restrict__ nq,
                                                         long long q_stride, const uint8_t *__restrict__ t,
                                                         const int *__restrict__ nt, long long t_stride,
                                                         uint2 *__restrict__ partial, int *__restrict__ best_idx,
                                                         int *__restrict__ best_dist, int *__restrict__ second_dist,
                                                         int out_stride, int nsplit) {
    __shared__ __attribute__((aligned(16))) float s_tk[MT_CHUNK];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), pr = blockIdx.z, sp = blockIdx.y;
    const int NQ = min(nq[pr], out_stride), NT = min(nt[pr], 1 << 20);
    if ((int)blockIdx.x * MT_QPB(QT) >= NQ) return;
    const int r = lane & 31, h = lane >> 5;
    const uint8_t *qp = q + (long long)pr * q_stride;
    const uint8_t *tp = t + (long long)pr * t_stride;
    const int per = ((NT + 31) / 32 + nsplit - 1) / nsplit * 32;
    const int j0 = min(NT, sp * per), j1 = min(NT, j0 + per);
    // ---- queries: lane (r, h) holds bytes 16h .. 16h+15 of query r of each tile; dword p of them is the K block of MFMA p
    const int qw = blockIdx.x * MT_QPB(QT) + w * MT_QPW(QT);
    const bool wave_on = qw < NQ;
    mt_v4i bq[QT][4];
    int pq[QT];
#pragma unroll
    for (int c = 0; c < QT; ++c) {
        const int qi = qw + 32 * c + r;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (qi < NQ) v = *(const uint4 *)(qp + (long long)qi * 32 + 16 * h);
        pq[c] = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
        pq[c] += __shfl_xor(pq[c], 32);
        const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t e = (vv[p] >> k) & 0x11111111u;
                bq[c][p][k] = (int)((e << 1) | (e << 2) | (e << 3));   // 0b1110 per set bit
            }
        }
    }
    uint32_t gbest[QT], gsecond[QT];
#pragma unroll
    for (int c = 0; c < QT; ++c) gbest[c] = gsecond[c] = 0xffffffffu;
    for (int c0 = j0; c0 < j1; c0 += MT_CHUNK) {
        const int c1 = min(j1, c0 + MT_CHUNK), n = c1 - c0, npad = (n + 31) & ~31;
        __syncthreads();
        for (int i = threadIdx.x; i < npad; i += 64 * MT_WAVES) {
            float key = __int_as_float(MT_F4_NONE);
            if (i < n) {
                const uint4 a = *(const uint4 *)(tp + (long long)(c0 + i) * 32), b = *(const uint4 *)(tp + (long long)(c0 + i) * 32 + 16);
                const int pt = __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);
                key = (float)(((pt + 256) << 12) | i);   // < 2^24: exact
            }
            s_tk[i] = key;
        }
        __syncthreads();
        if (!wave_on) continue;
        int best[QT], second[QT];
#pragma unroll
        for (int c = 0; c < QT; ++c) best[c] = second[c] = MT_F4_NONE;
        const uint8_t *trow = tp + (long long)(c0 + r) * 32 + 16 * h;
        uint4 ta = make_uint4(0, 0, 0, 0);
        if (r < n) ta = *(const uint4 *)trow;
        for (int jt = 0; jt < npad; jt += 32) {
            const uint4 tc = ta;
            ta = make_uint4(0, 0, 0, 0);
            if (jt + 32 + r < n) ta = *(const uint4 *)(trow + (long long)(jt + 32) * 32);
            mt_v16f acc[QT];
            {
                const float4 *kp = (const float4 *)(s_tk + jt + 4 * h);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 k4 = kp[2 * g];
                    acc[0][4 * g] = k4.x; acc[0][4 * g + 1] = k4.y; acc[0][4 * g + 2] = k4.z; acc[0][4 * g + 3] = k4.w;
                }
#pragma unroll
                for (int c = 1; c < QT; ++c) acc[c] = acc[0];
            }
            const uint32_t tw[4] = {tc.x, tc.y, tc.z, tc.w};
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                mt_v8i a = {0, 0, 0, 0, 0, 0, 0, 0};
                a[0] = (int)((tw[p] << 2) & 0x44444444u); a[1] = (int)((tw[p] << 1) & 0x44444444u);
                a[2] = (int)(tw[p] & 0x44444444u);        a[3] = (int)((tw[p] >> 1) & 0x44444444u);
#pragma unroll
                for (int c = 0; c < QT; ++c) {
                    mt_v8i b = {0, 0, 0, 0, 0, 0, 0, 0};
                    b[0] = bq[c][p][0]; b[1] = bq[c][p][1]; b[2] = bq[c][p][2]; b[3] = bq[c][p][3];
                    acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[c], 4, 4, 0, 137, 0, 127);
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
#pragma unroll
                for (int c = 0; c < QT; ++c) {
                    const int kc = __float_as_int(acc[c][i]);
                    second[c] = max(min(best[c], second[c]), min(max(best[c], second[c]), kc));
                    best[c] = min(best[c], kc);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < QT; ++c) {
            const int bk = (int)__int_as_float(best[c]), sk = (int)__int_as_float(second[c]);   // the integer keys (exact; unused when NONE)
            const uint32_t b = best[c] == MT_F4_NONE ? 0xffffffffu
                                                     : ((uint32_t)((bk >> 12) + pq[c] - 256) << 20) | (uint32_t)(c0 + (bk & 4095));
            const uint32_t s2 = second[c] == MT_F4_NONE ? 0xffffffffu
                                                        : ((uint32_t)((sk >> 12) + pq[c] - 256) << 20) | (uint32_t)(c0 + (sk & 4095));
            gsecond[c] = mt_min2(mt_min2(gsecond[c], s2), mt_max2(gbest[c], b));
            gbest[c] = mt_min2(gbest[c], b);
        }
    }
    if (!wave_on) return;
#pragma unroll
    for (int c = 0; c < QT; ++c) {
        const uint32_t ob = (uint32_t)__shfl_xor((int)gbest[c], 32), os = (uint32_t)__shfl_xor((int)gsecond[c], 32);
        const uint32_t second = mt_min2(mt_min2(gsecond[c], os), mt_max2(gbest[c], ob));
        const uint32_t best = mt_min2(gbest[c], ob);
        const int qi = qw + 32 * c + r;
        if (h == 0 && qi < NQ) {
            if (nsplit > 1) {
                partial[((long long)pr * nsplit + sp) * out_stride + qi] = make_uint2(best, second);
            } else {
                const long long o = (long long)pr * out_stride + qi;
                best_idx[o] = best == 0xffffffffu ? -1 : (int)(best & 0xfffffu);
                best_dist[o] = best == 0xffffffffu ? 0x7fffffff : (int)(best >> 20);
                second_dist[o] = second == 0xffffffffu ? 0x7fffffff : (int)(second >> 20);
            }
        }
    }
}

// The same bookkeeping on the vector pipe (the round-1 kernel, ORBX_MATCH_KERNEL=valu): two queries per lane (16 dwords in
// VGPRs); the train descriptor of an iteration is the same for the whole wave, so it is fetched with SCALAR loads
// (s_load_dwordx8 through the scalar cache) and used as the SGPR operand of v_xor: no LDS staging, no barrier.  19 vector
// instructions per pair of distances; partial results always go through k_match_merge.
__global__ __launch_bounds__(64) void k_match_valu(const uint8_t *__restrict__ q, const int *__restrict__ nq,
                                                         long long q_stride, const uint8_t *__restrict__ t,
                                                         const int *__restrict__ nt, long long t_stride,
                                                         uint2 *__restrict__ partial, int out_stride, int nsplit) {
    // key = dist << 20 | index: min(key) is the best match with the lowest index on ties; the second-smallest
    // key carries the second-best distance (counting duplicates), exactly the bookkeeping of the reference loops.
    const int lane = threadIdx.x, pr = blockIdx.z, sp = blockIdx.y;
    const int NQ = min(nq[pr], out_stride), NT = min(nt[pr], 1 << 20);   // contract (orbx.h): counts beyond out_stride are ignored
    const int qw = blockIdx.x * 128;
    if (qw >= NQ) return;
    const int qi0 = qw + lane, qi1 = qi0 + 64;
    const uint4 *qp = (const uint4 *)(q + (long long)pr * q_stride);
    const uint4 *tp = (const uint4 *)(t + (long long)pr * t_stride);
    uint4 qa0 = make_uint4(0, 0, 0, 0), qb0 = qa0, qa1 = qa0, qb1 = qa0;
    if (qi0 < NQ) { qa0 = qp[2 * qi0]; qb0 = qp[2 * qi0 + 1]; }
    if (qi1 < NQ) { qa1 = qp[2 * qi1]; qb1 = qp[2 * qi1 + 1]; }
    const int chunk = (NT + nsplit - 1) / nsplit;
    const int j0 = sp * chunk, j1 = min(NT, j0 + chunk);
    uint32_t best0 = 0xffffffffu, second0 = 0xffffffffu, best1 = 0xffffffffu, second1 = 0xffffffffu;
#pragma unroll 4
    for (int j = j0; j < j1; ++j) {
        const uint4 ta = tp[2 * j], tb = tp[2 * j + 1];   // wave-uniform address: scalar loads
        const uint32_t d0 = __popc(qa0.x ^ ta.x) + __popc(qa0.y ^ ta.y) + __popc(qa0.z ^ ta.z) + __popc(qa0.w ^ ta.w) +
                            __popc(qb0.x ^ tb.x) + __popc(qb0.y ^ tb.y) + __popc(qb0.z ^ tb.z) + __popc(qb0.w ^ tb.w);
        const uint32_t d1 = __popc(qa1.x ^ ta.x) + __popc(qa1.y ^ ta.y) + __popc(qa1.z ^ ta.z) + __popc(qa1.w ^ ta.w) +
                            __popc(qb1.x ^ tb.x) + __popc(qb1.y ^ tb.y) + __popc(qb1.z ^ tb.z) + __popc(qb1.w ^ tb.w);
        const uint32_t key0 = (d0 << 20) | (uint32_t)j, key1 = (d1 << 20) | (uint32_t)j;
        // best <= second always: the new second-smallest is the median of (best, second, key)  (one v_med3_u32)
        second0 = max(min(best0, second0), min(max(best0, second0), key0));
        best0 = min(best0, key0);
        second1 = max(min(best1, second1), min(max(best1, second1), key1));
        best1 = min(best1, key1);
    }
    uint2 *po = partial + ((long long)pr * nsplit + sp) * out_stride;
    if (qi0 < NQ) po[qi0] = make_uint2(best0, second0);
    if (qi1 < NQ) po[qi1] = make_uint2(best1, second1);
}

__global__ __launch_bounds__(256) void k_match_merge(const int *__restrict__ nq, const uint2 *__restrict__ partial,
                                                     int *__restrict__ best_idx, int *__restrict__ best_dist,
                                                     int *__restrict__ second_dist, int out_stride, int nsplit) {
    const int pr = blockIdx.y, qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= min(nq[pr], out_stride)) return;
    uint32_t best = 0xffffffffu, second = 0xffffffffu;
    for (int k = 0; k < nsplit; ++k) {
        const uint2 p = partial[((long long)pr * nsplit + k) * out_stride + qi];
        second = min(min(second, p.y), max(best, p.x));
        best = min(best, p.x);
    }
    const long long o = (long long)pr * out_stride + qi;
    best_idx[o] = best == 0xffffffffu ? -1 : (int)(best & 0xfffffu);
    best_dist[o] = best == 0xffffffffu ? 0x7fffffff : (int)(best >> 20);
    second_dist[o] = second == 0xffffffffu ? 0x7fffffff : (int)(second >> 20);
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

// ------------------------------------------------------------------------------------------------
// K8: Frame::ComputeStereoMatches (reference src/Frame.cc:880-1176), one wave per LEFT keypoint.
//  phase A: every lane scans right keypoints (stride 64); the reference's row table (:926-942) is the predicate
//           floor(yR - r) <= (int)vL <= ceil(yR + r) with r = 2*scale[octaveR]; candidates are visited in ascending
//           index, so min over key = dist << 16 | iR reproduces the strict '<' bookkeeping (:990-1018).
//  phase B: 11x11 patch vs 11 shifts, centre-subtracted L1 (:1040-1101) in exact integer arithmetic; the
//           parabola (:1121-1129) uses the same single-rounded float operations as the reference.
// The final median cut (:1160-1175) is a sort over <= N integers and stays on the host (orbx_api.cpp).
// ------------------------------------------------------------------------------------------------
// 16 lanes per LEFT keypoint, four keypoints per wave: every cross-lane step (candidate minimum, the 11 SAD sums) is a DPP
// rotate inside a row of 16 lanes and serves four keypoints at once, and a wave's chain of dependent loads (keypoint -> row
// table -> candidates -> descriptors -> best match -> patches) is paid once per four keypoints.  (One wave per keypoint: 330 us
// per 256 KITTI pairs, 11 x 7 DPP steps and a 5 us load chain per keypoint.)
__device__ __forceinline__ uint32_t st_row16_min(uint32_t v) {   // all-reduce min inside each row of 16 lanes
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xf, 0xf, false));   // row_ror:1
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xf, 0xf, false));   // row_ror:2
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false));   // row_ror:4
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false));   // row_ror:8
    return v;
}
__device__ __forceinline__ uint32_t st_row16_sum(uint32_t v) {   // all-reduce sum inside each row of 16 lanes
    v += (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false);
    return v;
}
#define ST_KPB 16   // left keypoints per 256-thread block
__device__ __forceinline__ void orbx_stereo_body(const OrbxStereoGeom &sg, const orbx_keypoint *__restrict__ kL,
                                                 const uint8_t *__restrict__ dL, int nL,
                                                 const orbx_keypoint *__restrict__ kR,
                                                 const uint8_t *__restrict__ dR, int nR,
                                                 const uint8_t *__restrict__ pyrL, const uint8_t *__restrict__ pyrR,
                                                 float *__restrict__ uRight, float *__restrict__ depth,
                                                 int *__restrict__ sad, const int *__restrict__ row_begin,
                                                 const uint2 *__restrict__ row_items) {
    // No lane leaves early: the reductions below are DPP operations every lane of the wave takes part in; a keypoint that
    // drops out (`act`) just stops contributing.
    const int l16 = threadIdx.x & 15;
    const int iL = blockIdx.x * ST_KPB + (threadIdx.x >> 4);
    bool act = iL < nL;
    float uL = 0.f, vL = 0.f;
    int levelL = 0;
    if (act) { const orbx_keypoint kp = kL[iL]; uL = kp.x; vL = kp.y; levelL = kp.octave; }
    if (act && l16 == 0) { uRight[iL] = -1.0f; depth[iL] = -1.0f; sad[iL] = -1; }
    const long long row = (long long)vL;
    act = act && row >= 0 && row < sg.nrows0;                 // F6 clamp (reference: out-of-bounds index)
    const float maxD = sg.mbf / sg.mb, minD = 0.f;
    const float minU = uL - maxD, maxU = uL - minD;
    act = act && !(maxU < 0);
    uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
    if (act) { const uint4 *qp = (const uint4 *)(dL + (long long)iL * 32); qa = qp[0]; qb = qp[1]; }
    uint32_t best = 0xffffffffu;
    if (row_begin) {
        // vRowIndices[vL] of the reference (:926-942), built on the device by k_stereo_rows: only the right keypoints whose
        // row band covers this row are visited (their order does not matter: ties go to the smallest index, as the
        // reference's strict '<' over ascending iR resolves them)
        int j = 0, j1 = 0;
        if (act) { j = row_begin[row] + l16; j1 = row_begin[row + 1]; }
        for (; j < j1; j += 16) {
            // a table entry carries what the gate needs (x, index | octave << 16): one coalesced 8-byte load per lane
            // instead of a scattered read of the 28-byte keypoint record
            const uint2 it = row_items[j];
            const int iR = (int)(it.y & 0xffffu), octR = (int)(it.y >> 16);
            const float xR = __uint_as_float(it.x);
            const bool cand = (int)(octR >= levelL - 1) & (int)(octR <= levelL + 1) & (int)(xR >= minU) & (int)(xR <= maxU);
            if (cand) {
                const uint4 *tp = (const uint4 *)(dR + (long long)iR * 32);
                const uint4 ta = tp[0], tb = tp[1];
                const uint32_t d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                                   __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
                if (d < 100u) best = min(best, (d << 16) | (uint32_t)iR);
            }
        }
    } else {
        for (int iR = act ? l16 : nR; iR < nR; iR += 16) {
            const orbx_keypoint kr = kR[iR];
            const float r = 2.0f * sg.scale[kr.octave];
            const int maxr = (int)ceilf(kr.y + r), minr = (int)floorf(kr.y - r);
            const bool cand = (int)(row >= minr) & (int)(row <= maxr) & (int)(kr.octave >= levelL - 1) &
                              (int)(kr.octave <= levelL + 1) & (int)(kr.x >= minU) & (int)(kr.x <= maxU);
            if (cand) {
                const uint4 *tp = (const uint4 *)(dR + (long long)iR * 32);
                const uint4 ta = tp[0], tb = tp[1];
                const uint32_t d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                                   __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
                if (d < 100u) best = min(best, (d << 16) | (uint32_t)iR);   // bestDist starts at TH_HIGH, strict '<'
            }
        }
    }
    best = st_row16_min(best);
    act = act && best != 0xffffffffu && (best >> 16) < 75u;   // thOrbDist = (TH_HIGH + TH_LOW) / 2
    const int bestIdxR = act ? (int)(best & 0xffffu) : 0;
    float uR0 = 0.f;
    if (act) uR0 = kR[bestIdxR].x;
    const float scaleFactor = sg.inv_scale[levelL];
    const float scaleduL = roundf(uL * scaleFactor), scaledvL = roundf(vL * scaleFactor);
    const float scaleduR0 = roundf(uR0 * scaleFactor);
    const int w = 5, Lh = 5;
    const int W = sg.pw[levelL], H = sg.ph[levelL], pitch = sg.pitch[levelL];
    const int y0 = (int)(scaledvL - w), x0 = (int)(scaleduL - w);
    act = act && !(y0 < 0 || y0 + 2 * w + 1 > H || x0 < 0 || x0 + 2 * w + 1 > W);
    const float iniu = scaleduR0 - Lh - w, endu = scaleduR0 + Lh + w + 1;   // fork: minus (src/Frame.cc:1067)
    act = act && !(iniu < 0 || endu >= W);
    // ---- 11 x 11 patch against 11 shifts: SAD_s = sum |(L - cL) - (R[+s] - cR_s)| = sum |(L + cR_s - cL + 511) - (R[+s] + 511)|
    // (both sides non-negative 16-bit values).  Two pixels per lane and round in the 16-bit halves of a register: v_perm
    // picks byte s of both pixels' 12-byte windows of the right image, v_sad_u16 accumulates.  The centre pixel contributes 0 to every shift by construction, so the 7
    // slots beyond the 121 pixels of the 128 a group walks are filled with it.
    uint32_t acc[11];
#pragma unroll
    for (int s = 0; s < 11; ++s) acc[s] = 0u;
    if (act) {
        const uint8_t *IL = pyrL + sg.off[levelL] + (long long)y0 * pitch + x0;
        const uint8_t *IR = pyrR + sg.off[levelL] + (long long)y0 * pitch + ((int)scaleduR0 - w);
        const uint32_t cL = IL[w * pitch + w];
        const orbx_uint3_u cw = *(const orbx_uint3_u *)(IR + w * pitch);   // cR_s = IR[w * pitch + w + (s - Lh)] = byte s
        uint32_t Cs[11];
#pragma unroll
        for (int s = 0; s < 11; ++s) {
            const uint32_t cwd = s < 4 ? cw.x : s < 8 ? cw.y : cw.z;
            Cs[s] = (((cwd >> (8 * (s & 3))) & 0xffu) + 511u - cL) * 0x10001u;   // cR_s - cL + 511 in both halves (256 .. 766)
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int q0 = l16 + 16 * k, q1 = q0 + 64;
            if (q1 >= 121) q1 = 60;                           // the centre pixel: |0 - 0| for every shift
            const int y_0 = q0 / 11, x_0 = q0 - 11 * y_0, y_1 = q1 / 11, x_1 = q1 - 11 * y_1;
            const uint32_t L0 = IL[y_0 * pitch + x_0], L1 = IL[y_1 * pitch + x_1];
            const orbx_uint3_u r0 = *(const orbx_uint3_u *)(IR + y_0 * pitch + x_0 - Lh);   // R bytes x - 5 .. x + 6 of the row
            const orbx_uint3_u r1 = *(const orbx_uint3_u *)(IR + y_1 * pitch + x_1 - Lh);
            const uint32_t A0 = L0 | (L1 << 16);               // + Cs[s] = L + cR_s - cL + 511 in both halves (256 .. 1021)
#pragma unroll
            for (int s = 0; s < 11; ++s) {
                const uint32_t lo = s < 4 ? r0.x : s < 8 ? r0.y : r0.z, hi = s < 4 ? r1.x : s < 8 ? r1.y : r1.z;
                const uint32_t sel = (uint32_t)(s & 3) | (0x0cu << 8) | ((uint32_t)(4 + (s & 3)) << 16) | (0x0cu << 24);
                const uint32_t B = __builtin_amdgcn_perm(hi, lo, sel) + 0x01ff01ffu;   // R + 511 in both halves
                acc[s] = __builtin_amdgcn_sad_u16(A0 + Cs[s], B, acc[s]);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 11; ++s) acc[s] = st_row16_sum(acc[s]);
    if (!act || l16 != 0) return;
    int bestDistS = 0x7fffffff, bestinc = 0;
#pragma unroll
    for (int s = 0; s < 11; ++s)
        if ((int)acc[s] < bestDistS) { bestDistS = (int)acc[s]; bestinc = s - Lh; }
    if (bestinc == -Lh || bestinc == Lh) return;
    float dist1 = 0.f, dist2 = 0.f, dist3 = 0.f;
#pragma unroll
    for (int s = 1; s < 10; ++s)
        if (s - Lh == bestinc) { dist1 = (float)(int)acc[s - 1]; dist2 = (float)(int)acc[s]; dist3 = (float)(int)acc[s + 1]; }
    const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
    if (deltaR < -1 || deltaR > 1) return;
    float bestuR = sg.scale[levelL] * ((float)scaleduR0 + (float)bestinc + deltaR);
    float disparity = uL - bestuR;
    if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) { disparity = (float)0.01; bestuR = (float)((double)uL - 0.01); }
        depth[iL] = sg.mbf / disparity;
        uRight[iL] = bestuR;
        sad[iL] = bestDistS;
    }
}

__global__ __launch_bounds__(256) void k_stereo(OrbxStereoGeom sg, const orbx_keypoint *__restrict__ kL,
                                                const uint8_t *__restrict__ dL, int nL,
                                                const orbx_keypoint *__restrict__ kR,
                                                const uint8_t *__restrict__ dR, int nR,
                                                const uint8_t *__restrict__ pyrL, const uint8_t *__restrict__ pyrR,
                                                float *__restrict__ uRight, float *__restrict__ depth,
                                                int *__restrict__ sad, const int *__restrict__ row_begin,
                                                const uint2 *__restrict__ row_items) {
    orbx_stereo_body(sg, kL, dL, nL, kR, dR, nR, pyrL, pyrR, uRight, depth, sad, row_begin, row_items);
}
// batched: blockIdx.y = stereo pair; keypoints / descriptors / results of pair p at p * cap, pyramids at p * pyr_bytes
__global__ __launch_bounds__(256) void k_stereo_batch(OrbxStereoGeom sg, const orbx_keypoint *__restrict__ kL,
                                                      const uint8_t *__restrict__ dL, const int *__restrict__ nL,
                                                      const orbx_keypoint *__restrict__ kR,
                                                      const uint8_t *__restrict__ dR, const int *__restrict__ nR, int cap,
                                                      const uint8_t *__restrict__ pyrL, const uint8_t *__restrict__ pyrR,
                                                      long long pyr_bytes, float *__restrict__ uRight,
                                                      float *__restrict__ depth, int *__restrict__ sad,
                                                      const int *__restrict__ row_begin, const uint2 *__restrict__ row_items,
                                                      int items_per_pair) {
    const long long p = blockIdx.y;
    orbx_stereo_body(sg, kL + p * cap, dL + p * cap * 32, min(nL[p], cap), kR + p * cap, dR + p * cap * 32, min(nR[p], cap),
                     pyrL + p * pyr_bytes, pyrR + p * pyr_bytes, uRight + p * cap, depth + p * cap, sad + p * cap,
                     row_begin ? row_begin + p * (sg.nrows0 + 1) : nullptr, row_items ? row_items + p * items_per_pair : nullptr);
}
// vRowIndices of Frame::ComputeStereoMatches (src/Frame.cc:926-942) for one stereo pair per workgroup: right keypoint iR is
// listed under every row of [floor(y - r), ceil(y + r)], r = 2 * mvScaleFactors[octave] (rows outside the image are clamped
// away: the reference indexes out of bounds there).  Histogram, scan and fill all in LDS; rows <= ST_MAX_ROWS.
#define ST_MAX_ROWS 4096
__global__ __launch_bounds__(1024) void k_stereo_rows(OrbxStereoGeom sg, const orbx_keypoint *__restrict__ kR,
                                                      const int *__restrict__ nR, int cap, int *__restrict__ row_begin,
                                                      uint2 *__restrict__ row_items, int items_per_pair, int n_direct) {
    __shared__ int s_cnt[ST_MAX_ROWS + 1];
    __shared__ int s_part[1024];
    const long long p = blockIdx.x;
    const int t = threadIdx.x, n = nR ? min(nR[p], cap) : n_direct, rows = sg.nrows0;   // (single pair: the count comes by value)
    const orbx_keypoint *k = kR + p * cap;
    for (int i = t; i <= rows; i += 1024) s_cnt[i] = 0;
    __syncthreads();
    for (int i = t; i < n; i += 1024) {
        const float r = 2.0f * sg.scale[k[i].octave];
        const int maxr = min((int)ceilf(k[i].y + r), rows - 1), minr = max((int)floorf(k[i].y - r), 0);
        for (int y = minr; y <= maxr; ++y) atomicAdd(&s_cnt[y], 1);
    }
    __syncthreads();
    // exclusive scan over the rows: each thread owns a contiguous chunk
    const int chunk = (rows + 1023) / 1024, r0 = t * chunk, r1 = min(r0 + chunk, rows);
    int sum = 0;
    for (int y = r0; y < r1; ++y) sum += s_cnt[y];
    s_part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = t >= o ? s_part[t - o] : 0;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    int run = s_part[t] - sum;
    int *rb = row_begin + p * (rows + 1);
    for (int y = r0; y < r1; ++y) { const int c = s_cnt[y]; s_cnt[y] = run; rb[y] = run; run += c; }
    if (t == 1023) rb[rows] = s_part[1023];
    __syncthreads();
    uint2 *items = row_items + p * items_per_pair;
    for (int i = t; i < n; i += 1024) {
        const float r = 2.0f * sg.scale[k[i].octave];
        const int maxr = min((int)ceilf(k[i].y + r), rows - 1), minr = max((int)floorf(k[i].y - r), 0);
        for (int y = minr; y <= maxr; ++y) {
            const int pos = atomicAdd(&s_cnt[y], 1);
            if (pos < items_per_pair) items[pos] = make_uint2(__float_as_uint(k[i].x), (uint32_t)i | ((uint32_t)k[i].octave << 16));
        }
    }
}
// Median cut of Frame::ComputeStereoMatches (src/Frame.cc:1160-1175) on the device, one workgroup per pair: the reference
// sorts (SAD, index) and drops everything with SAD >= 1.5f * 1.4f * median, median = element size/2 of the sorted list.
// Only the VALUE of that element matters: two 256-bin histogram passes over the 16-bit SAD (<= 121 * 510) select it.
// (the two selections over the 256 bins are block-wide prefix sums -- one wave scan + four wave totals -- not a serial walk by
// one thread: 21 -> 6 us per 64 KITTI pairs)
__global__ __launch_bounds__(256) void k_stereo_cut(const int *__restrict__ nL, int cap, const int *__restrict__ sad,
                                                    float *__restrict__ uRight, float *__restrict__ depth,
                                                    int *__restrict__ nmatches) {
    __shared__ int hist[256];
    __shared__ int wsum[4];
    __shared__ int s_sel, s_rank, s_kept;
    const long long p = blockIdx.x;
    const int n = min(nL[p], cap), t = threadIdx.x, wv = t >> 6;
    const int *sd = sad + p * cap;
    float *ur = uRight + p * cap, *dp = depth + p * cap;
    hist[t] = 0;
    if (t == 0) { s_kept = 0; s_sel = -1; s_rank = 0; }
    __syncthreads();
    for (int i = t; i < n; i += 256) { const int v = sd[i]; if (v >= 0) atomicAdd(&hist[(v >> 8) & 255], 1); }
    __syncthreads();
    // bin that holds the element of rank k = count / 2 (nth_element semantics of :1160-1165), and the rank inside it
    int v = hist[t];
    int incl = orbx_wave_scan(v);
    if ((t & 63) == 63) wsum[wv] = incl;
    __syncthreads();
    const int cnt = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    for (int i = 0; i < wv; ++i) incl += wsum[i];
    if (cnt > 0) {
        const int k = cnt / 2;
        if (incl - v <= k && k < incl) { s_sel = t; s_rank = k - (incl - v); }
    }
    if (t == 0) s_kept = cnt;
    __syncthreads();
    const int hi = s_sel, rank = s_rank;
    if (hi < 0) { if (t == 0) nmatches[p] = 0; return; }
    __syncthreads();
    hist[t] = 0;
    __syncthreads();
    for (int i = t; i < n; i += 256) { const int u = sd[i]; if (u >= 0 && ((u >> 8) & 255) == hi) atomicAdd(&hist[u & 255], 1); }
    __syncthreads();
    v = hist[t];
    incl = orbx_wave_scan(v);
    if ((t & 63) == 63) wsum[wv] = incl;
    __syncthreads();
    for (int i = 0; i < wv; ++i) incl += wsum[i];
    if (incl - v <= rank && rank < incl) s_sel = (hi << 8) | t;
    __syncthreads();
    const float median = (float)s_sel;
    const float thDist = 1.5f * 1.4f * median;
    int dropped = 0;
    for (int i = t; i < n; i += 256) {
        const int u = sd[i];
        if (u >= 0 && (float)u >= thDist) { ur[i] = -1.0f; dp[i] = -1.0f; ++dropped; }
    }
    if (dropped) atomicSub(&s_kept, dropped);
    __syncthreads();
    if (t == 0) nmatches[p] = s_kept;
}

// small helper: zero per-batch counters / status
// ------------------------------------------------------------------------------------------------
// K9: DBoW2 TemplatedVocabulary<FORB>::transform(feature, word_id, weight, nid, levelsup)
// (reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1240-1285, FORB::distance FORB.cpp:81-101; caller
// Frame::ComputeBoW src/Frame.cc:750-765 with levelsup = 4).  16 lanes per descriptor: lane c takes child c of the current
// node (k <= 20 in every loadable vocabulary: one or two rounds), XOR + popcount over the 8 dwords, then a rotate-reduce
// inside the 16-lane DPP row on the key (distance << 16 | position) -- the minimum is the FIRST closest child, as the
// reference's strict '<' scan picks it.  Outputs the leaf node and the node at depth L - levelsup; word id and weight are
// table look-ups the host does in double precision.
// ------------------------------------------------------------------------------------------------
struct DVoc {
    const int *child_begin;       // n_nodes + 1 offsets into child_ids (Node::children in vector order)
    const uint32_t *child_ids;
    const uint8_t *desc;          // n_nodes x 32
    int n_nodes, L;
};
__device__ __forceinline__ uint32_t orbx_row16_min(uint32_t v) {   // all-reduce min inside each row of 16 lanes
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xf, 0xf, false));   // row_ror:1
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xf, 0xf, false));   // row_ror:2
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false));   // row_ror:4
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false));   // row_ror:8
    return v;
}
__global__ __launch_bounds__(256) void k_bow_transform(DVoc voc, const uint8_t *__restrict__ desc,
                                                       const int *__restrict__ counts, long long frame_stride, int levelsup,
                                                       uint32_t *__restrict__ out_leaf, uint32_t *__restrict__ out_nid,
                                                       int out_stride) {
    const int sub = threadIdx.x & 15;
    const int feat = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int f = blockIdx.y;
    const int n = counts[f];
    // inactive groups keep walking a dummy descriptor: the DPP reduction wants whole rows executing together
    const bool live = feat < n;
    const uint4 *q = (const uint4 *)(desc + (long long)f * frame_stride + (long long)(live ? feat : 0) * 32);
    const uint4 qa = n > 0 ? q[0] : make_uint4(0, 0, 0, 0), qb = n > 0 ? q[1] : make_uint4(0, 0, 0, 0);
    const int nid_level = voc.L - levelsup;
    uint32_t node = 0, nid = 0;
    for (int level = 1; level <= 64; ++level) {   // child ids exceed their parent's (checked on the host): terminates
        const int cb = voc.child_begin[node], ce = voc.child_begin[node + 1];
        if (cb == ce) break;                      // leaf
        uint32_t best = 0xffffffffu;
        for (int c0 = cb; c0 < ce; c0 += 16) {
            const int c = c0 + sub;
            uint32_t key = 0xffffffffu;
            if (c < ce) {
                const uint4 *t = (const uint4 *)(voc.desc + (long long)voc.child_ids[c] * 32);
                const uint4 ta = t[0], tb = t[1];
                const uint32_t d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                                   __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
                key = (d << 16) | (uint32_t)(c - cb);
            }
            best = min(best, orbx_row16_min(key));
        }
        node = voc.child_ids[cb + (int)(best & 0xffffu)];
        if (level == nid_level) nid = node;
    }
    if (live && sub == 0) {
        out_leaf[(long long)f * out_stride + feat] = node;
        out_nid[(long long)f * out_stride + feat] = nid;
    }
}

// ------------------------------------------------------------------------------------------------
// K10: Frame::UndistortKeyPoints (reference src/Frame.cc:770-825) = cv::undistortPoints(K, D, R = I, P = K) on the
// keypoint coordinates (OpenCV 3.2 cvUndistortPoints: double precision, five fixed iterations; parity unpinned, see
// oracle/orb_oracle_match.c).  One thread per keypoint, the other 20 bytes of the record are copied.
// ------------------------------------------------------------------------------------------------
struct DUndist { double fx, fy, cx, cy, k[14]; int identity; };
__global__ __launch_bounds__(256) void k_undistort(DUndist u, const orbx_keypoint *__restrict__ kps, const int *__restrict__ counts,
                                                   int fixed_n, int cap, orbx_keypoint *__restrict__ out) {
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n = counts ? min(counts[f], cap) : fixed_n;
    if (i >= n) return;
    orbx_keypoint kp = kps[(long long)f * cap + i];
    if (!u.identity) {
        const double ifx = 1. / u.fx, ify = 1. / u.fy;
        double x = (double)kp.x, y = (double)kp.y;
        const double x0 = x = (x - u.cx) * ifx;
        const double y0 = y = (y - u.cy) * ify;
        for (int j = 0; j < 5; ++j) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((u.k[7] * r2 + u.k[6]) * r2 + u.k[5]) * r2) / (1 + ((u.k[4] * r2 + u.k[1]) * r2 + u.k[0]) * r2);
            const double deltaX = 2 * u.k[2] * x * y + u.k[3] * (r2 + 2 * x * x) + u.k[8] * r2 + u.k[9] * r2 * r2;
            const double deltaY = u.k[2] * (r2 + 2 * y * y) + 2 * u.k[3] * x * y + u.k[10] * r2 + u.k[11] * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        const double xx = u.fx * x + 0.0 * y + u.cx, yy = 0.0 * x + u.fy * y + u.cy;
        const double ww = 1. / (0.0 * x + 0.0 * y + 1.0);
        kp.x = (float)(xx * ww);
        kp.y = (float)(yy * ww);
    }
    out[(long long)f * cap + i] = kp;
}

// ------------------------------------------------------------------------------------------------
// K11: the Frame grid on the device.  k_grid_build = Frame::AssignFeaturesToGrid + PosInGrid (reference src/Frame.cc:432-460,
// 729-745): 64 x 48 buckets, every bucket lists its features in ascending feature index (push_back order).  k_gate =
// Frame::GetFeaturesInArea (:633-717) for one query per wave, fused with DescriptorDistance: the cells are walked in the
// reference's order (ix outer, iy inner -- for one ix the buckets iy0..iy1 are one contiguous range of the CSR item
// array), the radius / level tests (incl. the quirk `bCheckLevels = minLevel > 0 || maxLevel >= 0`, :673) run per lane,
// survivors are ballot-compacted in visiting order and leave as (feature index | Hamming distance << 16).  The policies'
// order-dependent bookkeeping reads these short lists on the host instead of a dense nq x nt distance matrix.
// ------------------------------------------------------------------------------------------------
#define GR_COLS 64
#define GR_ROWS 48
#define GR_CELLS (GR_COLS * GR_ROWS)
__device__ __forceinline__ int gr_cell_of(const DGrid &gp, const orbx_keypoint &kp) {
    const int px = (int)roundf((kp.x - gp.minx) * gp.winv), py = (int)roundf((kp.y - gp.miny) * gp.hinv);   // C round(): half away from zero
    return (px < 0 || px >= GR_COLS || py < 0 || py >= GR_ROWS) ? -1 : px * GR_ROWS + py;
}
__global__ __launch_bounds__(1024) void k_grid_build(DGrid gp, const orbx_keypoint *__restrict__ kps, const int *__restrict__ counts,
                                                     int fixed_n, int cap, int *__restrict__ cell_begin, uint16_t *__restrict__ items) {
    __shared__ int s_cnt[GR_CELLS];
    __shared__ int s_part[1024];
    const long long f = blockIdx.x;
    const int t = threadIdx.x;
    const int n = counts ? min(counts[f], cap) : fixed_n;
    const orbx_keypoint *k = kps + f * cap;
    for (int c = t; c < GR_CELLS; c += 1024) s_cnt[c] = 0;
    __syncthreads();
    for (int i = t; i < n; i += 1024) {
        const int c = gr_cell_of(gp, k[i]);
        if (c >= 0) atomicAdd(&s_cnt[c], 1);
    }
    __syncthreads();
    // exclusive scan over the 3072 buckets: three per thread, Hillis-Steele over the thread sums
    const int c0 = 3 * t;
    const int a0 = s_cnt[c0], a1 = s_cnt[c0 + 1], a2 = s_cnt[c0 + 2];
    const int sum = a0 + a1 + a2;
    s_part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = t >= o ? s_part[t - o] : 0;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    int run = s_part[t] - sum;
    int *cb = cell_begin + f * (GR_CELLS + 1);
    cb[c0] = run; s_cnt[c0] = run; run += a0;
    cb[c0 + 1] = run; s_cnt[c0 + 1] = run; run += a1;
    cb[c0 + 2] = run; s_cnt[c0 + 2] = run;
    if (t == 1023) cb[GR_CELLS] = s_part[1023];
    __syncthreads();
    uint16_t *it = items + f * cap;
    for (int i = t; i < n; i += 1024) {
        const int c = gr_cell_of(gp, k[i]);
        if (c >= 0) it[atomicAdd(&s_cnt[c], 1)] = (uint16_t)i;
    }
    __syncthreads();
    // the atomics filled every bucket in arrival order: restore feature order (buckets hold a handful of entries)
    for (int c = t; c < GR_CELLS; c += 1024) {
        const int b = cb[c], e = s_cnt[c];
        for (int i = b + 1; i < e; ++i) {
            const uint16_t v = it[i];
            int j = i - 1;
            while (j >= b && it[j] > v) { it[j + 1] = it[j]; --j; }
            it[j + 1] = v;
        }
    }
}

// One wave per query, two walks over the query's buckets: the first counts the candidates (radius / level tests only), lane 0
// then reserves that many entries of the output with one returning atomicAdd on `cursor` and records (offset, count) in
// span[query]; the second walk evaluates the Hamming distances and writes the entries in visiting order.  A reservation that
// does not fit `cap` writes nothing: the host reads the total from the cursor, grows the buffer and launches again.
__device__ __forceinline__ bool gr_pass(const DGateQuery &Q, bool check, const orbx_keypoint &kp) {
    bool pass = fabsf(kp.x - Q.x) < Q.r && fabsf(kp.y - Q.y) < Q.r;
    if (check) pass = pass && !(kp.octave < Q.min_level) && !(Q.max_level >= 0 && kp.octave > Q.max_level);
    return pass;
}
__global__ __launch_bounds__(256) void k_gate(DGrid gp, const orbx_keypoint *__restrict__ kps, const uint8_t *__restrict__ desc,
                                              const int *__restrict__ cell_begin, const uint16_t *__restrict__ items,
                                              const DGateQuery *__restrict__ q, const uint8_t *__restrict__ qdesc, int nq,
                                              uint2 *__restrict__ span, uint32_t *__restrict__ cursor,
                                              uint32_t *__restrict__ out_items, uint32_t cap, int fstride) {
    const int lane = threadIdx.x & 63;
    const int qi = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (qi >= nq) return;
    const DGateQuery Q = q[qi];
    // batched calls: target Q.frame's keypoints / descriptors / buckets are fstride records further on (item indices stay
    // relative to their target)
    kps += (long long)Q.frame * fstride;
    desc += (long long)Q.frame * fstride * 32;
    items += (long long)Q.frame * fstride;
    cell_begin += (long long)Q.frame * (GR_CELLS + 1);
    // GetFeaturesInArea's cell range (:643-668); a negative radius switches the query off
    const int x0 = max(0, (int)floorf((Q.x - gp.minx - Q.r) * gp.winv));
    const int x1 = min(GR_COLS - 1, (int)ceilf((Q.x - gp.minx + Q.r) * gp.winv));
    const int y0 = max(0, (int)floorf((Q.y - gp.miny - Q.r) * gp.hinv));
    const int y1 = min(GR_ROWS - 1, (int)ceilf((Q.y - gp.miny + Q.r) * gp.hinv));
    const bool live = Q.r >= 0.f && x0 < GR_COLS && x1 >= 0 && y0 < GR_ROWS && y1 >= 0;
    const bool check = (Q.min_level > 0) || (Q.max_level >= 0);
    int n = 0;
    if (live)
        for (int ix = x0; ix <= x1; ++ix) {
            const int b = cell_begin[ix * GR_ROWS + y0], e = cell_begin[ix * GR_ROWS + y1 + 1];
            for (int j0 = b; j0 < e; j0 += 64) {
                const int j = j0 + lane;
                const bool pass = j < e && gr_pass(Q, check, kps[items[j]]);
                n += __popcll(orbx_ballot(pass));
            }
        }
    uint32_t base = 0;
    if (lane == 0) {
        base = n > 0 ? atomicAdd(cursor, (uint32_t)n) : 0u;
        span[qi] = make_uint2(base, (uint32_t)n);
    }
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (n == 0 || base + (uint32_t)n > cap) return;
    const uint4 *qp = (const uint4 *)(qdesc + (long long)(Q.desc >= 0 ? Q.desc : qi) * 32);
    const uint4 qa = qp[0], qb = qp[1];
    uint32_t w = base;
    for (int ix = x0; ix <= x1; ++ix) {
        const int b = cell_begin[ix * GR_ROWS + y0], e = cell_begin[ix * GR_ROWS + y1 + 1];
        for (int j0 = b; j0 < e; j0 += 64) {
            const int j = j0 + lane;
            bool pass = false;
            int i2 = 0;
            if (j < e) { i2 = items[j]; pass = gr_pass(Q, check, kps[i2]); }
            const unsigned long long m = orbx_ballot(pass);
            if (pass) {
                const uint4 *tp = (const uint4 *)(desc + (long long)i2 * 32);
                const uint4 ta = tp[0], tb = tp[1];
                const uint32_t d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                                   __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
                out_items[w + (uint32_t)orbx_wave_rank(m)] = (uint32_t)i2 | (d << 16);
            }
            w += (uint32_t)__popcll(m);
        }
    }
}

// Hamming distances of the BoW-guided policies: row r pairs descriptor rows[r].q of the first set with the features
// col_idx[col_begin .. col_begin + ncol) of the second set (the features under the same vocabulary node, in the reference's
// visiting order); distances leave as uint16 at out[out_off ..).  One wave per row.
__global__ __launch_bounds__(256) void k_block_dist(const uint8_t *__restrict__ d1, const uint8_t *__restrict__ d2,
                                                    const DDistRow *__restrict__ rows, const uint32_t *__restrict__ col_idx, int nrows,
                                                    uint16_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= nrows) return;
    const DDistRow R = rows[r];
    const uint4 *qp = (const uint4 *)(d1 + (long long)R.q * 32);
    const uint4 qa = qp[0], qb = qp[1];
    for (uint32_t j = lane; j < R.ncol; j += 64) {
        const uint4 *tp = (const uint4 *)(d2 + (long long)col_idx[R.col_begin + j] * 32);
        const uint4 ta = tp[0], tb = tp[1];
        out[R.out_off + j] = (uint16_t)(__popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                                        __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w));
    }
}

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
#include <string>
#include <cstdio>
#include <cstring>
#include <mutex>

hipError_t orbx_upload_pattern() {
    signed char t[64 * 16];
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r)
            for (int k = 0; k < 4; ++k) t[l * 16 + r * 4 + k] = ORBX_PATTERN_I8[(r * 64 + l) * 4 + k];
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_pattern_lane), t, sizeof(t));
    if (e != hipSuccess) return e;
    // IC_Angle weight table: umax of the reference's constructor (src/ORBextractor.cc:866-910) for HALF_PATCH_SIZE = 15
    orbx_params p;
    orbx_default_params(&p);
    OrbxTables tab;
    orbx_build_tables(p, tab);
    static uint32_t w[64][12];
    memset(w, 0, sizeof(w));
    for (int lane = 0; lane < 62; ++lane) {
        const int row = lane >> 1, half = lane & 1, v = row - ORBX_HALF_PATCH;
        const int d = tab.umax[v < 0 ? -v : v];
        for (int k = 0; k < 5; ++k)
            for (int b = 0; b < 4; ++b) {
                const int col = 4 * (1 + 5 * half + k) + b;      // byte column inside the 44-byte LDS patch row
                const int u = col - DS_R;
                if (col < DS_W && u >= -d && u <= d) {
                    w[lane][k] |= (uint32_t)(u + 16) << (8 * b);
                    w[lane][5 + k] |= 1u << (8 * b);
                }
            }
    }
    e = hipMemcpyToSymbol(HIP_SYMBOL(c_orient_w), w, sizeof(w));
    if (e != hipSuccess) return e;
    static uint32_t bb[3][64][4];
    memset(bb, 0, sizeof(bb));
    const int K7[7] = {18, 34, 49, 55, 49, 34, 18};   // the 8-bit Gaussian of k_blur / k_describe
    for (int j = 0; j < 3; ++j)
        for (int lane = 0; lane < 64; ++lane)
            for (int sl = 0; sl < 16; ++sl) {
                const int k = 16 * (lane >> 4) + sl, i = k - (16 * j + (lane & 15));
                if (i >= 0 && i <= 6) bb[j][lane][sl >> 2] |= (uint32_t)K7[i] << (8 * (sl & 3));
            }
    return hipMemcpyToSymbol(HIP_SYMBOL(c_blur_b), bb, sizeof(bb));
}

size_t orbx_quadtree_smem(int ncap, int lds_keys) {
    return 128 + (size_t)ncap * (8 * 4 + 16 + 8 + 6 * 4) + 16 + (size_t)lds_keys * 6 + 16;
}

void orbx_launch_clear(hipStream_t s, int *a, int na, int *b, int nb, int *c, int nc) {
    const int n = na > nb ? (na > nc ? na : nc) : (nb > nc ? nb : nc);
    hipLaunchKernelGGL(k_clear, dim3((n + 255) / 256), dim3(256), 0, s, a, na, b, nb, c, nc);
}
void orbx_launch_pyr_l0(hipStream_t s, const DGeom &g, int B, const uint8_t *imgs, int W, int H, int stride,
                        long long frame_stride, uint8_t *pyr, int *status, int *cand_cursor) {
    const DLevel &L = g.lv[0];
    // interior columns [32, xe): the 20-byte aligned window of every 16-pixel chunk stays inside the source row
    int xe = 32;
    while (xe + 1 <= W) xe += 16;           // chunk at X reads source bytes [X - 22, X + 1): must end inside the row
    const int icols = (xe - 32 + 1023) / 1024;
#if L0_FF
    dim3 grid(B, icols + 1, (L.ph + 4 * L0_ROWS - 1) / (4 * L0_ROWS));
#else
    dim3 grid(icols + 1, (L.ph + 4 * L0_ROWS - 1) / (4 * L0_ROWS), B);
#endif
    hipLaunchKernelGGL(k_pyr_l0, grid, dim3(64, 4), 0, s, g, imgs, W, H, stride, frame_stride, pyr, xe, status, cand_cursor);
}
void orbx_launch_pyr_l0_color(hipStream_t s, const DGeom &g, int B, const uint8_t *imgs, int W, int H, int stride,
                              long long frame_stride, uint8_t *pyr, int nch, int r_off, int b_off, int *status, int *cand_cursor) {
    const DLevel &L = g.lv[0];
    dim3 grid((L.pw + 255) / 256, (L.ph + 3) / 4, B);
    hipLaunchKernelGGL(k_pyr_l0_color, grid, dim3(64, 4), 0, s, g, imgs, W, H, stride, frame_stride, pyr, nch, r_off, b_off, status, cand_cursor);
}
void orbx_launch_pyr_l0_remap(hipStream_t s, const DGeom &g, int B, const uint8_t *imgs, int W, int H, int stride,
                              long long frame_stride, uint8_t *pyr, const uint2 *rect, int *status, int *cand_cursor) {
    const DLevel &L = g.lv[0];
    dim3 grid((L.pw + 255) / 256, (L.ph + 3) / 4, B);
    hipLaunchKernelGGL(k_pyr_l0_remap, grid, dim3(64, 4), 0, s, g, imgs, W, H, stride, frame_stride, pyr, rect, status, cand_cursor);
}
void orbx_launch_pyr_resize(hipStream_t s, const DGeom &g, int B, int level, const OrbxTap *taps, uint8_t *pyr, bool narrow) {
    const DLevel &L = g.lv[level];
    if (narrow) {
        // destination rows per wave: 16 when the launch still has >= 4096 waves, fewer for small batches (the rows of a
        // wave are a serial chain of load -> evaluate -> store steps)
        int rpw = 16;   // (8 / 16 / 32 / 64 rows per wave at 1024 frames: 750 / 708 / 729 / 742 us for the seven launches)
        while (rpw > 2 && (long long)((L.pw + 255) / 256) * ((L.ph + rpw - 1) / rpw) * B < 4096) rpw >>= 1;
#if RR_FF
        dim3 grid(B, (L.pw + 255) / 256, (L.ph + RR_WPB * rpw - 1) / (RR_WPB * rpw));
#else
        dim3 grid((L.pw + 255) / 256, (L.ph + RR_WPB * rpw - 1) / (RR_WPB * rpw), B);
#endif
        hipLaunchKernelGGL(k_pyr_resize_rows, grid, dim3(64, RR_WPB), 0, s, g, level, taps, pyr, rpw);
        return;
    }
    dim3 grid((L.pw + 255) / 256, (L.ph + 4 * RS_ROWS - 1) / (4 * RS_ROWS), B);
    hipLaunchKernelGGL(k_pyr_resize, grid, dim3(64, 4), 0, s, g, level, taps, pyr);
}
void orbx_launch_fast_rows(hipStream_t s, const DGeom &g, int B, const OrbxCell *cells, const OrbxFastGroup *groups,
                           int ngroups, const uint8_t *pyr, uint2 *cand, int *cand_cursor, int *status, int max_ch, int lcap,
                           int dbg_stop) {
    if (ngroups <= 0) return;
    lcap = (max(lcap, 64) + 1) & ~1;
    // LDS per wave decides how many waves a CU holds: the corner list of a group is sized for its usual load, FR_CCAP entries (the
    // average group has ~130 corners), not for the work list's worst case -- a group with more corners than that runs its NMS as the
    // dense rescan of the score map -- and tile / score map are rounded to 16 bytes, not to four rows.  At 640x480: 8 192 bytes with
    // 512 entries = exactly 20 waves per CU, the number the registers allow (round 3, second half: 1.316 / 1.319 -> 1.286 / 1.272 ms
    // against 256 entries = 7 680 bytes, because fewer groups take the rescan; 640 entries = 8 448 bytes = 19 waves: 1.39-1.41 ms).
    const int ccap = min(lcap, FR_CCAP);
    const size_t map_bytes = ((size_t)max_ch * FR_TP + 15) & ~(size_t)15;
    const size_t smem = 2 * map_bytes + (size_t)2 * lcap + 256 + (size_t)2 * ccap;
    // groups per wave: FR_GPW when the launch has waves to spare (the second group's tile is prefetched while the first
    // is processed); one per wave for small batches, where the serial length of a wave is what the caller waits for
    const int gpw = (long long)B * ngroups >= 16384 ? FR_GPW : 1;
    hipLaunchKernelGGL(k_fast_rows, dim3(B, (ngroups + gpw - 1) / gpw), dim3(64), smem, s, g, cells, groups, pyr, cand,
                       cand_cursor, status, max_ch, lcap, ngroups, gpw, dbg_stop, ccap);
}
void orbx_launch_undistort(hipStream_t s, int B, int max_n, int cap, const double *K4, const double *k14, int identity,
                           const orbx_keypoint *kps, const int *counts, orbx_keypoint *out) {
    if (B <= 0 || max_n <= 0) return;
    DUndist u;
    u.fx = K4[0]; u.fy = K4[1]; u.cx = K4[2]; u.cy = K4[3];
    for (int i = 0; i < 14; ++i) u.k[i] = k14[i];
    u.identity = identity;
    hipLaunchKernelGGL(k_undistort, dim3((max_n + 255) / 256, B), dim3(256), 0, s, u, kps, counts, max_n, cap, out);
}
void orbx_launch_bow_transform(hipStream_t s, int B, int max_n, const int *child_begin, const uint32_t *child_ids,
                               const uint8_t *node_desc, int n_nodes, int L, const uint8_t *desc, const int *counts,
                               long long frame_stride, int levelsup, uint32_t *out_leaf, uint32_t *out_nid, int out_stride) {
    if (B <= 0 || max_n <= 0) return;
    DVoc v;
    v.child_begin = child_begin; v.child_ids = child_ids; v.desc = node_desc; v.n_nodes = n_nodes; v.L = L;
    hipLaunchKernelGGL(k_bow_transform, dim3((max_n + 15) / 16, B), dim3(256), 0, s, v, desc, counts, frame_stride, levelsup,
                       out_leaf, out_nid, out_stride);
}
static int g_qt_cus = 0;   // CUs of the current device (orbx_quadtree_prepare)
void orbx_launch_quadtree(hipStream_t s, const DGeom &g, int B, const uint2 *dense, const int *cand_count, uint32_t *lvl_kp, int *lvl_count,
                          int *status, uint16_t *knode_glob, int ncap, int lds_keys, int level_begin, int level_count) {
    if (level_count <= 0) return;
    const size_t smem = orbx_quadtree_smem(ncap, lds_keys);
    // workgroups per CU of this launch decide the workgroup size (see k_quadtree)
    static int forced = -1;
    if (forced < 0) { const char *e = getenv("ORBX_QT_THREADS"); forced = e ? atoi(e) : 0; }
    const long long wgs = (long long)B * level_count, cus = g_qt_cus > 0 ? g_qt_cus : 256;
    int nfeat = 0;   // keys per workgroup scale with the feature quota: 2000 features want 512 threads even at 4 workgroups per CU
    for (int l = 0; l < g.nlevels; ++l) nfeat += g.lv[l].nfeat;   // (1241x376 / 2000, 128 frames: 77 / 63 / 99 us with 256 / 512 / 1024)
    int threads = wgs >= 4 * cus && nfeat <= 1600 ? 256 : wgs >= 2 * cus ? 512 : 1024;
    if (forced == 256 || forced == 512 || forced == 1024) threads = forced;
    const dim3 grid = QT_FF ? dim3(B, level_count) : dim3(level_count, B);
    if (threads == 256)
        hipLaunchKernelGGL(k_quadtree<256>, grid, dim3(256), smem, s, g, dense, cand_count, lvl_kp, lvl_count, status, knode_glob, ncap, lds_keys, level_begin);
    else if (threads == 512)
        hipLaunchKernelGGL(k_quadtree<512>, grid, dim3(512), smem, s, g, dense, cand_count, lvl_kp, lvl_count, status, knode_glob, ncap, lds_keys, level_begin);
    else
        hipLaunchKernelGGL(k_quadtree<1024>, grid, dim3(1024), smem, s, g, dense, cand_count, lvl_kp, lvl_count, status, knode_glob, ncap, lds_keys, level_begin);
}
hipError_t orbx_quadtree_prepare(size_t smem) {
    // The attribute belongs to (function, device), not to a handle: ORB-SLAM2 itself keeps mpIniORBextractor (2 x nFeatures)
    // alive next to mpORBextractorLeft (src/Tracking.cc:171-182), so a later, smaller handle must never lower it.
    static std::mutex mu;
    static size_t cur[64] = {0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (g_qt_cus == 0) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev) == hipSuccess) g_qt_cus = pr.multiProcessorCount; }
    if (dev >= 0 && dev < 64 && smem <= cur[dev]) return hipSuccess;
    e = hipFuncSetAttribute((const void *)k_quadtree<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_quadtree<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_quadtree<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess && dev >= 0 && dev < 64) cur[dev] = smem;
    return e;
}
void orbx_launch_blur(hipStream_t s, const DGeom &g, int B, const uint8_t *pyr, uint8_t *blur) {
    hipLaunchKernelGGL(k_blur, dim3(g.blur_tiles, B), dim3(256), 0, s, g, pyr, blur);
}
void orbx_launch_describe(hipStream_t s, const DGeom &g, int B, const uint8_t *pyr, const uint32_t *lvl_kp,
                          const int *lvl_count, float *lvl_angle, orbx_keypoint *kps, uint8_t *desc,
                          int *counts, int *status, int cap) {
#ifdef ORBX_TIMING_KNOBS
    static int dbg_stop = -1;
    if (dbg_stop < 0) { const char *e = getenv("ORBX_DESC_STOP"); dbg_stop = e ? atoi(e) : 0; }
#else
    const int dbg_stop = 0;
#endif
#if DS_ORDER == 1
    const dim3 grid(8 * ((g.kp_total + DS_WPB - 1) / DS_WPB), (B + 7) / 8);
#else
    const dim3 grid(B, (g.kp_total + DS_WPB - 1) / DS_WPB);
#endif
    if (g.fp_mode == ORBX_FP_GCC_FMA)
        hipLaunchKernelGGL(k_describe<ORBX_FP_GCC_FMA>, grid, dim3(64 * DS_WPB), 0, s, g, pyr, lvl_kp, lvl_count, lvl_angle, kps, desc, counts, status, cap, dbg_stop, B);
    else
        hipLaunchKernelGGL(k_describe<ORBX_FP_STRICT>, grid, dim3(64 * DS_WPB), 0, s, g, pyr, lvl_kp, lvl_count, lvl_angle, kps, desc, counts, status, cap, dbg_stop, B);
}
void orbx_launch_grid_build(hipStream_t s, const DGrid &gp, int nframes, const orbx_keypoint *kps, const int *counts, int fixed_n,
                            int cap, int *cell_begin, uint16_t *items) {
    if (nframes <= 0) return;
    hipLaunchKernelGGL(k_grid_build, dim3(nframes), dim3(1024), 0, s, gp, kps, counts, fixed_n, cap, cell_begin, items);
}
void orbx_launch_gate(hipStream_t s, const DGrid &gp, const orbx_keypoint *kps, const uint8_t *desc, const int *cell_begin,
                      const uint16_t *items, const DGateQuery *q, const uint8_t *qdesc, int nq, uint2 *span, uint32_t *cursor,
                      uint32_t *out_items, uint32_t cap, int fstride) {
    if (nq <= 0) return;
    hipLaunchKernelGGL(k_gate, dim3((nq + 3) / 4), dim3(256), 0, s, gp, kps, desc, cell_begin, items, q, qdesc, nq, span, cursor,
                       out_items, cap, fstride);
}
void orbx_launch_block_dist(hipStream_t s, const uint8_t *d1, const uint8_t *d2, const DDistRow *rows, const uint32_t *col_idx,
                            int nrows, uint16_t *out) {
    if (nrows <= 0) return;
    hipLaunchKernelGGL(k_block_dist, dim3((nrows + 3) / 4), dim3(256), 0, s, d1, d2, rows, col_idx, nrows, out);
}
size_t orbx_match_workspace_bytes(int npairs, int out_stride) { return (size_t)npairs * MT_SPLIT * out_stride * sizeof(uint2); }
void orbx_launch_match(hipStream_t s, int npairs, int max_nq, const uint8_t *q, const int *nq, long long q_stride,
                       const uint8_t *t, const int *nt, long long t_stride, int *best_idx, int *best_dist,
                       int *second_dist, int out_stride, void *workspace, int kernel) {
    // kernel: 0 = matrix pipe, FP4 operands (default); 1 = vector pipe (ORBX_MATCH_KERNEL=valu); 2 = matrix pipe, int8 operands (=i8)
    const bool use_valu = kernel == 1, f4 = kernel == 0;
    if (npairs <= 0 || max_nq <= 0) return;
    if (use_valu) {   // the vector-pipe kernel (orbx_params-free A/B switch ORBX_MATCH_KERNEL=valu, read once per handle)
        const int qblocks = (max_nq + 127) / 128;
        const long long base = (long long)qblocks * npairs;
        const int nsplit = (int)std::min<long long>(MT_SPLIT, std::max<long long>(1, (16384 + base - 1) / base));
        hipLaunchKernelGGL(k_match_valu, dim3(qblocks, nsplit, npairs), dim3(64), 0, s, q, nq, q_stride, t, nt, t_stride,
                           (uint2 *)workspace, out_stride, nsplit);
        hipLaunchKernelGGL(k_match_merge, dim3((max_nq + 255) / 256, npairs), dim3(256), 0, s, nq, (const uint2 *)workspace,
                           best_idx, best_dist, second_dist, out_stride, nsplit);
        return;
    }
    // 512 queries per block (QT = 4, two waves per SIMD: the chip holds 2048 such waves) when the launch fills the chip that
    // way without splitting the train set; otherwise 256 per block and the train split that reaches ~4096 waves
    const long long waves4 = (long long)((max_nq + MT_QPW(4) - 1) / MT_QPW(4)) * npairs;
    if (waves4 >= 2048 && (!f4 || MT_F4_USE4)) {
        hipLaunchKernelGGL(f4 ? k_match_f4<4> : k_match<4>, dim3((max_nq + MT_QPB(4) - 1) / MT_QPB(4), 1, npairs), dim3(64 * MT_WAVES), 0, s, q, nq,
                           q_stride, t, nt, t_stride, (uint2 *)workspace, best_idx, best_dist, second_dist, out_stride, 1);
        return;
    }
    const int qblocks = (max_nq + MT_QPB(2) - 1) / MT_QPB(2);
    const long long base = (long long)((max_nq + MT_QPW(2) - 1) / MT_QPW(2)) * npairs;
    const int nsplit = (int)std::min<long long>(MT_SPLIT, std::max<long long>(1, 4096 / base));
    hipLaunchKernelGGL(f4 ? k_match_f4<2> : k_match<2>, dim3(qblocks, nsplit, npairs), dim3(64 * MT_WAVES), 0, s, q, nq,
                       q_stride, t, nt, t_stride, (uint2 *)workspace, best_idx, best_dist, second_dist, out_stride, nsplit);
    if (nsplit > 1)
        hipLaunchKernelGGL(k_match_merge, dim3((max_nq + 255) / 256, npairs), dim3(256), 0, s, nq, (const uint2 *)workspace,
                           best_idx, best_dist, second_dist, out_stride, nsplit);
}
void orbx_launch_hamming_matrix(hipStream_t s, const uint8_t *q, int nq, const uint8_t *t, int nt, uint16_t *dist) {
    const long long n = (long long)nq * nt;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_hamming_matrix, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, q, nq, t, nt, dist);
}

int orbx_stereo_items_per_pair(const OrbxStereoGeom &sg, int cap) {
    float smax = 1.f;
    for (int l = 0; l < sg.nlevels; ++l) smax = sg.scale[l] > smax ? sg.scale[l] : smax;
    return cap * (2 * (int)ceilf(2.0f * smax) + 3);
}
void orbx_launch_stereo_batch(hipStream_t s, const OrbxStereoGeom &sg, int npairs, int cap, const orbx_keypoint *kL,
                              const uint8_t *dL, const int *nL, const orbx_keypoint *kR, const uint8_t *dR, const int *nR,
                              const uint8_t *pyrL, const uint8_t *pyrR, long long pyr_bytes, float *uRight, float *depth,
                              int *sad, int *nmatches, int *row_begin, uint2 *row_items) {
    if (npairs <= 0 || cap <= 0) return;
    const int ipp = orbx_stereo_items_per_pair(sg, cap);
    const bool table = row_begin && row_items && sg.nrows0 <= ST_MAX_ROWS;
    if (table) hipLaunchKernelGGL(k_stereo_rows, dim3(npairs), dim3(1024), 0, s, sg, kR, nR, cap, row_begin, row_items, ipp, 0);
    hipLaunchKernelGGL(k_stereo_batch, dim3((cap + ST_KPB - 1) / ST_KPB, npairs), dim3(256), 0, s, sg, kL, dL, nL, kR, dR, nR, cap, pyrL, pyrR,
                       pyr_bytes, uRight, depth, sad, table ? row_begin : nullptr, table ? row_items : nullptr, ipp);
    hipLaunchKernelGGL(k_stereo_cut, dim3(npairs), dim3(256), 0, s, nL, cap, sad, uRight, depth, nmatches);
}
void orbx_launch_stereo(hipStream_t s, const OrbxStereoGeom &sg, const orbx_keypoint *kL, const uint8_t *dL, int nL,
                        const orbx_keypoint *kR, const uint8_t *dR, int nR, const uint8_t *pyrL, const uint8_t *pyrR,
                        float *uRight, float *depth, int *sad, int *row_begin, uint2 *row_items) {
    if (nL <= 0) return;
    // the row table of the batched form (vRowIndices) for the one pair too: without it a left keypoint's 16 lanes walk every
    // right keypoint
    const bool table = row_begin && row_items && sg.nrows0 <= ST_MAX_ROWS && nR > 0;
    if (table)
        hipLaunchKernelGGL(k_stereo_rows, dim3(1), dim3(1024), 0, s, sg, kR, (const int *)nullptr, nR, row_begin, row_items,
                           orbx_stereo_items_per_pair(sg, nR), nR);
    hipLaunchKernelGGL(k_stereo, dim3((nL + ST_KPB - 1) / ST_KPB), dim3(256), 0, s, sg, kL, dL, nL, kR, dR, nR, pyrL, pyrR, uRight,
                       depth, sad, table ? row_begin : nullptr, table ? row_items : nullptr);
}

