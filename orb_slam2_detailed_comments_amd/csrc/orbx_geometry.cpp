// orbx_geometry.cpp -- host-side tables of the MI355X ORB front-end.
//
// Everything the reference derives once per extractor / per image size on the CPU is derived here,
// once per handle, and uploaded to the device: scale tables and per-level quotas
// (reference src/ORBextractor.cc:776-925), pyramid level sizes (:2097-2107), the FAST cell grid
// (:1438-1503), quadtree root layout (:1060-1090) and the fixed-point bilinear taps of cv::resize
// (OpenCV 3.2 imgwarp.cpp, restated in SURVEY.md Appendix B.2) with the reflect-101 border folded in.
//
// Built with -ffp-contract=off: every float expression below is a single IEEE operation per operator.
#include "orbx_internal.h"
#include <cmath>
#include <algorithm>

static inline int cv_round_f(float v) { return (int)lrintf(v); }      // cvRound: half-to-even
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil_f(float v) { int i = (int)v; return i + (i < v); }
static inline int16_t sat16(int v) { return (int16_t)std::min(32767, std::max(-32768, v)); }
static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

void orbx_build_tables(const orbx_params &p, OrbxTables &t) {
    const int L = p.nlevels;
    const double sf = (double)p.scale_factor;  // the member is a double initialised from the float argument
    t.scale[0] = 1.0f;
    t.sigma2[0] = 1.0f;
    for (int i = 1; i < L; ++i) {
        t.scale[i] = (float)((double)t.scale[i - 1] * sf);
        t.sigma2[i] = t.scale[i] * t.scale[i];
    }
    for (int i = 0; i < L; ++i) {
        t.inv_scale[i] = 1.0f / t.scale[i];
        t.inv_sigma2[i] = 1.0f / t.sigma2[i];
    }
    const float factor = (float)(1.0 / sf);
    float nd = (float)p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)L));
    int sum = 0;
    for (int l = 0; l < L - 1; ++l) {
        t.nfeat[l] = cv_round_f(nd);
        sum += t.nfeat[l];
        nd *= factor;
    }
    t.nfeat[L - 1] = std::max(p.nfeatures - sum, 0);
    // umax: quarter-circle half-widths of the 31x31 orientation patch
    const int vmax = cv_floor_f(ORBX_HALF_PATCH * sqrtf(2.f) / 2 + 1);
    const int vmin = cv_ceil_f(ORBX_HALF_PATCH * sqrtf(2.f) / 2);
    const double hp2 = ORBX_HALF_PATCH * ORBX_HALF_PATCH;
    for (int v = 0; v <= vmax; ++v) t.umax[v] = cv_round_d(std::sqrt(hp2 - v * v));
    for (int v = ORBX_HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (t.umax[v0] == t.umax[v0 + 1]) ++v0;
        t.umax[v] = v0;
        ++v0;
    }
}

// taps for destination coordinates 0..pdst-1 of a padded axis whose centre has `dcentre` samples,
// resized from `ssize` source samples.  horizontal: weights zeroed when clamped; vertical: kept.
static void build_axis_taps(std::vector<OrbxTap> &out, int pdst, int dcentre, int ssize, bool horizontal) {
    const double inv_scale = (double)dcentre / ssize;
    const double scale = 1. / inv_scale;
    for (int P = 0; P < pdst; ++P) {
        const int d = reflect101(P - ORBX_EDGE, dcentre);
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor_f(f);
        f -= s;
        OrbxTap tap;
        if (horizontal) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
            tap.s0 = (int16_t)s;
            tap.s1 = (int16_t)std::min(s + 1, ssize - 1);
        } else {
            tap.s0 = (int16_t)std::min(std::max(s, 0), ssize - 1);
            tap.s1 = (int16_t)std::min(std::max(s + 1, 0), ssize - 1);
        }
        tap.a0 = sat16(cv_round_f((1.f - f) * 2048));
        tap.a1 = sat16(cv_round_f(f * 2048));
        out.push_back(tap);
    }
}

orbx_status orbx_build_geometry(const orbx_params &p, const OrbxTables &t, int width, int height, OrbxGeom &g,
                                const char **why) {
    g = OrbxGeom();
    g.width = width; g.height = height; g.nlevels = p.nlevels;
    int64_t off = 0, cand_off = 0;
    int kp_off = 0;
    for (int l = 0; l < p.nlevels; ++l) {
        OrbxLevelGeom &L = g.lv[l];
        L.sw = cv_round_f((float)width * t.inv_scale[l]);
        L.sh = cv_round_f((float)height * t.inv_scale[l]);
        if (L.sw < 1 || L.sh < 1) { *why = "pyramid level collapses to zero size"; return ORBX_UNSUPPORTED; }
        L.pw = L.sw + 2 * ORBX_EDGE;
        L.ph = L.sh + 2 * ORBX_EDGE;
        L.pitch = (L.pw + 63) & ~63;
        L.off = off;
        off += (int64_t)L.pitch * L.ph;
        off = (off + 255) & ~(int64_t)255;
        L.scale = t.scale[l];
        L.size = (float)(int)(ORBX_PATCH * t.scale[l]);
        L.nfeat = t.nfeat[l];
        // cell grid
        const int minB = ORBX_EDGE - 3;
        const int maxBX = L.pw - ORBX_EDGE + 3, maxBY = L.ph - ORBX_EDGE + 3;
        const float fw = (float)(maxBX - minB), fh = (float)(maxBY - minB);
        L.ncols = (int)(fw / 30.f);
        L.nrows = (int)(fh / 30.f);
        L.cell_begin = (int)g.cells.size();
        int nms_bound = 0;
        L.wcell = L.hcell = 0;
        if (L.ncols > 0 && L.nrows > 0) {
            L.wcell = (int)std::ceil(fw / L.ncols);
            L.hcell = (int)std::ceil(fh / L.nrows);
            // local keypoint coordinates are < cell size - 3 and must fit 6 bits of the emission-order key; wCell < 60 always
            if (L.wcell + 6 > 67 || L.hcell + 6 > 67) { *why = "cell larger than 67 px"; return ORBX_UNSUPPORTED; }
            int ord = 0;
            for (int i = 0; i < L.nrows; ++i) {
                const float iniY = (float)(minB + i * L.hcell);
                float maxY = iniY + L.hcell + 6;
                if (iniY >= maxBY - 3) continue;
                if (maxY > maxBY) maxY = (float)maxBY;
                for (int j = 0; j < L.ncols; ++j) {
                    const float iniX = (float)(minB + j * L.wcell);
                    float maxX = iniX + L.wcell + 6;
                    if (iniX >= maxBX - 6) continue;
                    if (maxX > maxBX) maxX = (float)maxBX;
                    OrbxCell c;
                    c.x0 = (int16_t)iniX; c.y0 = (int16_t)iniY;
                    c.cw = (int16_t)((int)maxX - (int)iniX); c.ch = (int16_t)((int)maxY - (int)iniY);
                    c.offx = (int16_t)(j * L.wcell); c.offy = (int16_t)(i * L.hcell);
                    c.level = (int16_t)l;
                    c.idx_in_level = (int16_t)ord++;
                    if (c.cw >= 7 && c.ch >= 7) {  // cv::FAST yields nothing below 7x7
                        // strict 3x3 maxima: at most one survivor per 2x2 block of the cell interior.  Default = that
                        // exact worst case (cannot overflow); max_cand_per_cell > 0 trades memory for an ORBX_CAPACITY risk
                        int bound = ((c.cw - 6 + 1) / 2) * ((c.ch - 6 + 1) / 2);
                        if (p.max_cand_per_cell > 0) bound = std::min(bound, p.max_cand_per_cell);
                        c.slot_begin = nms_bound;
                        c.slot_cap = bound;
                        nms_bound += bound;
                        g.cells.push_back(c);
                    }
                }
            }
            if (ord > 4095) { *why = "more than 4095 FAST cells in one level"; return ORBX_UNSUPPORTED; }
        }
        L.cell_count = (int)g.cells.size() - L.cell_begin;
        // quadtree region and roots
        L.qt_w = maxBX - minB;
        L.qt_h = maxBY - minB;
        if (L.qt_w >= 4096 || L.qt_h >= 4096) { *why = "level larger than 4095 px"; return ORBX_UNSUPPORTED; }
        L.nini = (int)roundf((float)L.qt_w / L.qt_h);
        if (L.nini <= 0) { *why = "aspect ratio < 0.5: nIni == 0"; return ORBX_BAD_ASPECT; }
        L.hx = (float)L.qt_w / L.nini;
        L.kp_cap = std::max(L.nfeat + 3, 4 * L.nini);
        L.kp_begin = kp_off;
        kp_off += L.kp_cap;
        L.cand_cap = std::max(64, nms_bound);
        L.cand_begin = cand_off;
        cand_off += L.cand_cap;
        g.node_cap = std::max(g.node_cap, (L.kp_cap + 8 + 3) & ~3);   // multiple of 4: LDS arrays stay 16-byte aligned
        g.max_cand_cap = std::max(g.max_cand_cap, L.cand_cap);
        // resize taps (level > 0: source is the PADDED previous level)
        L.tapx_begin = L.tapy_begin = 0;
        L.narrow_taps = false;
        if (l > 0) {
            const OrbxLevelGeom &S = g.lv[l - 1];
            L.tapx_begin = (int)g.taps.size();
            build_axis_taps(g.taps, L.pw, L.sw, S.pw, true);
            L.tapy_begin = (int)g.taps.size();
            build_axis_taps(g.taps, L.ph, L.sh, S.ph, false);
            L.narrow_taps = true;
            for (int X = 0; X < L.pw; X += 4) {
                int lo = 0x7fff, hi = 0;
                for (int i = 0; i < 4; ++i) {
                    const OrbxTap &t = g.taps[L.tapx_begin + std::min(X + i, L.pw - 1)];
                    lo = std::min<int>(lo, t.s0); hi = std::max<int>(hi, t.s0);
                    // the kernel reads the second tap as "the next byte": true unless clamped, where its weight is 0
                    if (t.s1 != t.s0 + 1 && t.a1 != 0) L.narrow_taps = false;
                }
                if (hi + 2 - lo > 8) L.narrow_taps = false;
            }
        }
    }
    // FAST wave groups: pair a cell with its right-hand neighbour when both interiors fit the 64 lanes of a wave plus the
    // ORBX_FAST_XCOLS columns k_fast_rows tests outside its row walk (two 33-column cells: 66)
    for (size_t i = 0; i < g.cells.size();) {
        const OrbxCell &a = g.cells[i];
        OrbxFastGroup grp;
        grp.cell0 = (int32_t)i; grp.ncell = 1;
        if (i + 1 < g.cells.size()) {
            const OrbxCell &b = g.cells[i + 1];
            const bool adjacent = b.level == a.level && b.y0 == a.y0 && b.ch == a.ch && b.x0 == a.x0 + a.cw - 6 &&
                                  b.idx_in_level == a.idx_in_level + 1;
            if (adjacent && (a.cw - 6) + (b.cw - 6) <= 64 + ORBX_FAST_XCOLS) grp.ncell = 2;
        }
        g.fast_groups.push_back(grp);
        i += (size_t)grp.ncell;
    }
    g.pyr_bytes = off;
    g.cand_total = cand_off;
    g.kp_total = kp_off;
    return ORBX_OK;
}
