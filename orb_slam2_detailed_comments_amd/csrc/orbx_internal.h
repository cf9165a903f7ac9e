// orbx_internal.h -- host-side geometry / workspace description shared by the API and kernels.
#pragma once
#include <cstdint>
#include <vector>
#include "../../include/orbx.h"

#define ORBX_MAX_LEVELS 16
#define ORBX_EDGE 19        // EDGE_THRESHOLD, reference src/ORBextractor.cc:82
#define ORBX_HALF_PATCH 15  // HALF_PATCH_SIZE, :81
#define ORBX_PATCH 31       // PATCH_SIZE, :80
#define ORBX_NPATCH 749     // pixels of the orientation disc defined by umax

// One FAST cell (reference src/ORBextractor.cc:1465-1503): sub-image [y0,y0+ch) x [x0,x0+cw) of the padded level.
struct OrbxCell {
    int16_t x0, y0, cw, ch;  // sub-mat origin and size (padded-level coordinates)
    int16_t offx, offy;      // j*wCell, i*hCell  (added to the sub-mat-local keypoint, :1540-1541)
    int16_t level;
    int16_t idx_in_level;    // row-major ordinal of the cell inside its level (emission order)
    int32_t slot_begin;      // first candidate slot of this cell inside its level's candidate region
    int32_t slot_cap;        // slots reserved: strict 3x3 maxima leave at most one survivor per 2x2 block
};

// One wave of k_fast_rows: `ncell` (1 or 2) horizontally adjacent cells of one cell row whose interiors together span
// at most 64 + ORBX_FAST_XCOLS columns (one lane per interior column in the row walk; the columns beyond the wave are tested
// apart, lanes as rows).  The LDS tile pitch of the kernel bounds it: 64 + XCOLS + 6 ring + 3 alignment bytes <= 76.
#define ORBX_FAST_XCOLS 2
struct OrbxFastGroup {
    int32_t cell0;
    int32_t ncell;
};

// resize tap for one padded destination coordinate (border folded in by reflect-101)
struct OrbxTap {
    int16_t s0, s1;  // source index of the two taps (already clamped)
    int16_t a0, a1;  // 11-bit fixed-point weights
};

struct OrbxLevelGeom {
    int sw, sh;          // un-padded size
    int pw, ph, pitch;   // padded size and row pitch in bytes
    int64_t off;         // byte offset of this level inside one frame's pyramid slab
    int ncols, nrows, wcell, hcell;
    int cell_begin, cell_count;  // range in the per-frame cell table
    int qt_w, qt_h;      // quadtree region (maxBorder - minBorder)
    int nini;            // number of root nodes
    float hx;            // root width
    int nfeat;           // mnFeaturesPerLevel
    int kp_cap;          // max keypoints the quadtree can emit: max(nfeat + 3, 4 * nini)
    int kp_begin;        // slot offset inside the per-frame level-keypoint table
    int cand_cap;        // candidate slots
    int64_t cand_begin;  // offset inside the per-frame candidate table
    int tapx_begin, tapy_begin;  // offsets into the tap table (level > 0)
    bool narrow_taps;    // every aligned group of 4 destination columns reads <= 8 consecutive source bytes (k_pyr_resize_rows)
    float scale;         // mvScaleFactor[level]
    float size;          // (float)(int)(31 * scale)
};

struct OrbxGeom {
    int width = 0, height = 0, nlevels = 0;
    OrbxLevelGeom lv[ORBX_MAX_LEVELS];
    std::vector<OrbxCell> cells;
    std::vector<OrbxFastGroup> fast_groups;
    std::vector<OrbxTap> taps;
    int64_t pyr_bytes = 0;     // per frame
    int64_t cand_total = 0;    // per frame
    int kp_total = 0;          // per frame: sum of kp_cap
    int node_cap = 0;          // max over levels of quadtree node capacity
    int max_cand_cap = 0;
};

struct OrbxTables {
    float scale[ORBX_MAX_LEVELS], inv_scale[ORBX_MAX_LEVELS], sigma2[ORBX_MAX_LEVELS], inv_sigma2[ORBX_MAX_LEVELS];
    int nfeat[ORBX_MAX_LEVELS];
    int umax[ORBX_HALF_PATCH + 1];
};

// host-only math (orbx_geometry.cpp)
void orbx_build_tables(const orbx_params &p, OrbxTables &t);
// returns ORBX_OK / ORBX_BAD_ASPECT / ORBX_UNSUPPORTED
orbx_status orbx_build_geometry(const orbx_params &p, const OrbxTables &t, int width, int height, OrbxGeom &g,
                                const char **why);
