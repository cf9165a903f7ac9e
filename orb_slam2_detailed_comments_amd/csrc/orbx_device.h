// orbx_device.h -- device-side geometry block passed BY VALUE to every kernel (lives in the kernarg
// segment -> scalar loads), plus small device helpers shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "orbx_internal.h"
#include "orbx_sincos.h"

struct DLevel {
    int pw, ph, pitch, sw, sh;
    int cell_begin, cell_count;
    int qt_w, qt_h, nini;
    float hx;
    int nfeat, kp_cap, kp_begin, cand_cap;
    int tapx, tapy;
    float scale, size;
    int blur_tile_begin, blur_tx;  // flattened blur tile table
    long long off;                 // byte offset in the per-frame pyramid slab
    long long cand_begin;          // record offset in the per-frame candidate table
};

struct DGeom {
    int nlevels, ncells, kp_total, fp_mode;
    int ini_th, min_th, blur_tiles, pad0;
    long long pyr_bytes, cand_total;
    int umax[16];
    DLevel lv[ORBX_MAX_LEVELS];
};

// geometry of the two pyramids ComputeStereoMatches reads (both handles share it: same image size / params)
struct OrbxStereoGeom {
    int nlevels, nrows0;
    float mb, mbf;
    float scale[ORBX_MAX_LEVELS], inv_scale[ORBX_MAX_LEVELS];
    int pw[ORBX_MAX_LEVELS], ph[ORBX_MAX_LEVELS], pitch[ORBX_MAX_LEVELS];
    long long off[ORBX_MAX_LEVELS];
};

// Frame grid (64 x 48 buckets over [minx, maxx) x [miny, maxy)): mfGridElementWidthInv / HeightInv of src/Frame.cc:96-99
struct DGrid { float minx, miny, winv, hinv; };
// one GetFeaturesInArea query (src/Frame.cc:633-717): centre, radius (negative = query switched off), level band
// `frame` = which target of a batched call the query searches (orbx_gate_lists_batch); 0 for the single-target calls
// `desc` = row of the query-descriptor block this query compares with (-1: its own index; batched calls whose problems share
// one point set upload the descriptors once)
struct DGateQuery { float x, y, r; int min_level, max_level; int frame = 0; int desc = -1; };
// one row of a BoW-guided distance block: descriptor q of set 1 against ncol features of set 2
struct DDistRow { uint32_t q, col_begin, ncol, out_off; };

#define ORBX_WAVE 64

__device__ __forceinline__ int orbx_reflect101(int i, int n) {
    // single reflection is enough for |overshoot| < n (borders here are <= 19 px, n >= 20 enforced on host;
    // the loop keeps it exact for any n > 1)
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

// cross-lane moves through DPP (a VALU modifier) instead of __shfl (ds_bpermute occupies the LDS pipe for ~10 ns per
// wave on gfx950 against ~1.8 ns for a DPP move)
#define ORBX_DPP_ROW_SHR(n) (0x110 + (n))
#define ORBX_DPP_WAVE_SHL1 0x130   // lane i <- lane i + 1
#define ORBX_DPP_ROW_BCAST15 0x142
#define ORBX_DPP_ROW_BCAST31 0x143
__device__ __forceinline__ uint32_t orbx_lane_above(uint32_t v) {   // value of lane + 1 (lane 63 gets 0)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ORBX_DPP_WAVE_SHL1, 0xf, 0xf, true);
}
// inclusive prefix sum over the 64 lanes (the canonical GCN DPP scan: shift-adds inside each row of 16, then the row
// broadcasts): 8 DPP adds against 6 ds_bpermute round trips for a __shfl_up ladder
__device__ __forceinline__ int orbx_wave_scan(int v) {
    int t = v + __builtin_amdgcn_update_dpp(0, v, ORBX_DPP_ROW_SHR(1), 0xf, 0xf, true);
    t += __builtin_amdgcn_update_dpp(0, v, ORBX_DPP_ROW_SHR(2), 0xf, 0xf, true);
    t += __builtin_amdgcn_update_dpp(0, v, ORBX_DPP_ROW_SHR(3), 0xf, 0xf, true);
    t += __builtin_amdgcn_update_dpp(0, t, ORBX_DPP_ROW_SHR(4), 0xf, 0xe, true);
    t += __builtin_amdgcn_update_dpp(0, t, ORBX_DPP_ROW_SHR(8), 0xf, 0xc, true);
    t += __builtin_amdgcn_update_dpp(0, t, ORBX_DPP_ROW_BCAST15, 0xa, 0xf, true);
    t += __builtin_amdgcn_update_dpp(0, t, ORBX_DPP_ROW_BCAST31, 0xc, 0xf, true);
    return t;
}
// minimum over the 64 lanes, returned wave-uniform (same ladder; lanes without a source keep the identity)
__device__ __forceinline__ uint32_t orbx_wave_min(uint32_t v) {
    const int I = -1;
    const int iv = (int)v;
    uint32_t t = min(v, (uint32_t)__builtin_amdgcn_update_dpp(I, iv, ORBX_DPP_ROW_SHR(1), 0xf, 0xf, false));
    t = min(t, (uint32_t)__builtin_amdgcn_update_dpp(I, iv, ORBX_DPP_ROW_SHR(2), 0xf, 0xf, false));
    t = min(t, (uint32_t)__builtin_amdgcn_update_dpp(I, iv, ORBX_DPP_ROW_SHR(3), 0xf, 0xf, false));
    t = min(t, (uint32_t)__builtin_amdgcn_update_dpp(I, (int)t, ORBX_DPP_ROW_SHR(4), 0xf, 0xe, false));
    t = min(t, (uint32_t)__builtin_amdgcn_update_dpp(I, (int)t, ORBX_DPP_ROW_SHR(8), 0xf, 0xc, false));
    t = min(t, (uint32_t)__builtin_amdgcn_update_dpp(I, (int)t, ORBX_DPP_ROW_BCAST15, 0xa, 0xf, false));
    t = min(t, (uint32_t)__builtin_amdgcn_update_dpp(I, (int)t, ORBX_DPP_ROW_BCAST31, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)t, 63);
}
// sum over the 64 lanes, returned wave-uniform: shift-adds inside each row of 16, then row broadcasts (the canonical
// GCN reduction), result read from lane 63
__device__ __forceinline__ int orbx_wave_sum(int v) {
    int t = v + __builtin_amdgcn_update_dpp(0, v, ORBX_DPP_ROW_SHR(1), 0xf, 0xf, true);
    t += __builtin_amdgcn_update_dpp(0, v, ORBX_DPP_ROW_SHR(2), 0xf, 0xf, true);
    t += __builtin_amdgcn_update_dpp(0, v, ORBX_DPP_ROW_SHR(3), 0xf, 0xf, true);
    t += __builtin_amdgcn_update_dpp(0, t, ORBX_DPP_ROW_SHR(4), 0xf, 0xe, true);
    t += __builtin_amdgcn_update_dpp(0, t, ORBX_DPP_ROW_SHR(8), 0xf, 0xc, true);
    t += __builtin_amdgcn_update_dpp(0, t, ORBX_DPP_ROW_BCAST15, 0xa, 0xf, true);
    t += __builtin_amdgcn_update_dpp(0, t, ORBX_DPP_ROW_BCAST31, 0xc, 0xf, true);
    return __builtin_amdgcn_readlane(t, 63);
}

// cv::fastAtan2 (degrees), OpenCV 3.2 atanImpl<float>; no contraction (file is built -ffp-contract=off)
__device__ __forceinline__ float orbx_fast_atan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

