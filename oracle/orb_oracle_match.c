/* orb_oracle_match.c -- CPU restatement of the matcher policies on the hot path (see orb_oracle.h).
 * TEST INFRASTRUCTURE ONLY.
 *
 *   Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea   src/Frame.cc:432-460, 729-745, 633-717
 *   ORBmatcher::SearchForInitialization                           src/ORBmatcher.cc:570-712
 *   Frame::ComputeStereoMatches                                   src/Frame.cc:880-1176
 *
 * The Frame / KeyFrame / MapPoint classes cannot be linked here (they need OpenCV, Eigen, DBoW2, g2o), so the
 * functions take the fields they read as plain arrays.  Fork deviations are followed:
 * `factor = HISTO_LENGTH / 360.0f`, `iniu = scaleduR0 - L - w`.
 */
#define _GNU_SOURCE
#include "orb_oracle.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define FRAME_GRID_ROWS 48 /* include/Frame.h:54 */
#define FRAME_GRID_COLS 64 /* include/Frame.h:59 */
#define TH_HIGH 100
#define TH_LOW 50
#define HISTO_LENGTH 30

/* ------------------------------------------------------------------ grid */
struct orc_grid {
    float minx, miny, winv, hinv;
    int n;
    const orc_keypoint *kps;
    int *cell_begin; /* [COLS*ROWS+1] */
    int *items;      /* feature indices, cell-major, insertion order */
};

/* Frame::AssignFeaturesToGrid src/Frame.cc:432-460 with PosInGrid :729-745 */
orc_grid *orc_grid_build(const orc_keypoint *kps, int n, float minx, float maxx, float miny, float maxy) {
    orc_grid *g = (orc_grid *)calloc(1, sizeof(*g));
    g->minx = minx; g->miny = miny;
    g->winv = (float)FRAME_GRID_COLS / (maxx - minx); /* src/Frame.cc:405-407 */
    g->hinv = (float)FRAME_GRID_ROWS / (maxy - miny);
    g->n = n; g->kps = kps;
    const int nc = FRAME_GRID_COLS * FRAME_GRID_ROWS;
    int *cell = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    g->cell_begin = (int *)calloc(nc + 1, sizeof(int));
    g->items = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        int px = (int)roundf((kps[i].x - minx) * g->winv);
        int py = (int)roundf((kps[i].y - miny) * g->hinv);
        if (px < 0 || px >= FRAME_GRID_COLS || py < 0 || py >= FRAME_GRID_ROWS) cell[i] = -1;
        else { cell[i] = px * FRAME_GRID_ROWS + py; g->cell_begin[cell[i] + 1]++; }
    }
    for (int c = 0; c < nc; ++c) g->cell_begin[c + 1] += g->cell_begin[c];
    int *fill = (int *)calloc(nc, sizeof(int));
    for (int i = 0; i < n; ++i)
        if (cell[i] >= 0) g->items[g->cell_begin[cell[i]] + fill[cell[i]]++] = i;
    free(fill); free(cell);
    return g;
}
void orc_grid_free(orc_grid *g) {
    if (!g) return;
    free(g->cell_begin); free(g->items); free(g);
}

/* Frame::GetFeaturesInArea src/Frame.cc:633-717 (including the bCheckLevels quirk :673) */
int orc_grid_query(const orc_grid *g, float x, float y, float r, int minLevel, int maxLevel, int *out, int cap) {
    int n = 0;
    int nMinCellX = (int)floorf((x - g->minx - r) * g->winv);
    if (nMinCellX < 0) nMinCellX = 0;
    if (nMinCellX >= FRAME_GRID_COLS) return 0;
    int nMaxCellX = (int)ceilf((x - g->minx + r) * g->winv);
    if (nMaxCellX > FRAME_GRID_COLS - 1) nMaxCellX = FRAME_GRID_COLS - 1;
    if (nMaxCellX < 0) return 0;
    int nMinCellY = (int)floorf((y - g->miny - r) * g->hinv);
    if (nMinCellY < 0) nMinCellY = 0;
    if (nMinCellY >= FRAME_GRID_ROWS) return 0;
    int nMaxCellY = (int)ceilf((y - g->miny + r) * g->hinv);
    if (nMaxCellY > FRAME_GRID_ROWS - 1) nMaxCellY = FRAME_GRID_ROWS - 1;
    if (nMaxCellY < 0) return 0;
    const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int c = ix * FRAME_GRID_ROWS + iy;
            for (int j = g->cell_begin[c]; j < g->cell_begin[c + 1]; ++j) {
                const orc_keypoint *kp = &g->kps[g->items[j]];
                if (bCheckLevels) {
                    if (kp->octave < minLevel) continue;
                    if (maxLevel >= 0 && kp->octave > maxLevel) continue;
                }
                const float distx = kp->x - x, disty = kp->y - y;
                if (fabsf(distx) < r && fabsf(disty) < r) {
                    if (n < cap) out[n] = g->items[j];
                    ++n;
                }
            }
        }
    return n;
}

/* ------------------------------------------------------------------ SearchForInitialization */
/* src/ORBmatcher.cc:570-712.  prev_matched[2*n1] is vbPrevMatched (in/out), matches12[n1] out. */
int orc_search_for_initialization(const orc_keypoint *k1, const uint8_t *d1, int n1, const orc_keypoint *k2,
                                  const uint8_t *d2, int n2, float minx, float maxx, float miny, float maxy,
                                  float *prev_matched, int window, float nnratio, int check_ori, int *matches12) {
    int nmatches = 0;
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    int *hist = (int *)malloc(sizeof(int) * HISTO_LENGTH * (n1 > 0 ? n1 : 1));
    int hn[HISTO_LENGTH] = {0};
    const float factor = HISTO_LENGTH / 360.0f; /* fork: src/ORBmatcher.cc:583 */
    int *matchedDist = (int *)malloc(sizeof(int) * (n2 > 0 ? n2 : 1));
    int *matches21 = (int *)malloc(sizeof(int) * (n2 > 0 ? n2 : 1));
    int *cands = (int *)malloc(sizeof(int) * (n2 > 0 ? n2 : 1));
    for (int i = 0; i < n2; ++i) { matchedDist[i] = INT_MAX; matches21[i] = -1; }
    orc_grid *g = orc_grid_build(k2, n2, minx, maxx, miny, maxy);
    for (int i1 = 0; i1 < n1; ++i1) {
        const int level1 = k1[i1].octave;
        if (level1 > 0) continue;
        const int nc = orc_grid_query(g, prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)window, level1, level1, cands, n2);
        if (nc == 0) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int c = 0; c < nc; ++c) {
            const int i2 = cands[c];
            const int dist = orc_descriptor_distance(d1 + (size_t)i1 * 32, d2 + (size_t)i2 * 32);
            if (matchedDist[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; nmatches--; }
                matches12[i1] = bestIdx2;
                matches21[bestIdx2] = i1;
                matchedDist[bestIdx2] = bestDist;
                nmatches++;
                if (check_ori) {
                    float rot = k1[i1].angle - k2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    hist[bin * n1 + hn[bin]++] = i1;
                }
            }
        }
    }
    if (check_ori) {
        int ind1, ind2, ind3;
        orc_three_maxima(hn, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < hn[i]; j++) {
                const int idx1 = hist[i * n1 + j];
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
            }
        }
    }
    for (int i1 = 0; i1 < n1; ++i1)
        if (matches12[i1] >= 0) {
            prev_matched[2 * i1] = k2[matches12[i1]].x;
            prev_matched[2 * i1 + 1] = k2[matches12[i1]].y;
        }
    orc_grid_free(g);
    free(hist); free(matchedDist); free(matches21); free(cands);
    return nmatches;
}

/* ------------------------------------------------------------------ ComputeStereoMatches */
typedef struct { int dist, idx; } distidx;
static int distidx_cmp(const void *a, const void *b) {
    const distidx *x = (const distidx *)a, *y = (const distidx *)b;
    if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

/* src/Frame.cc:880-1176.  pyrL/pyrR: the padded pyramid levels of the two extractors (fork: mvImagePyramid holds
 * the padded images), pw/ph their sizes (row stride = pw).  F6: the reference indexes vRowIndices[yi] without a
 * bound check (rows reach past mvImagePyramid[0].rows in fork mode) -- the restatement clamps, which is identical
 * wherever the reference is well-defined.  An empty vDistIdx (reference: undefined behaviour) returns 0 matches. */
int orc_stereo_matches(const orc_keypoint *kL, const uint8_t *dL, int nL, const orc_keypoint *kR, const uint8_t *dR,
                       int nR, int nlevels, const float *scale, const float *inv_scale, const uint8_t *const *pyrL,
                       const uint8_t *const *pyrR, const int *pw, const int *ph, float mb, float mbf, float *uRight,
                       float *depth) {
    for (int i = 0; i < nL; ++i) { uRight[i] = -1.0f; depth[i] = -1.0f; }
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = ph[0];
    /* row table: lists of right keypoints per image row, in ascending iR order */
    int *rcount = (int *)calloc(nRows + 1, sizeof(int));
    for (int iR = 0; iR < nR; ++iR) {
        const float kpY = kR[iR].y;
        const float r = 2.0f * scale[kR[iR].octave];
        const int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
        for (int yi = minr; yi <= maxr; ++yi)
#ifdef ORC_F6_UNCLAMPED
            /* sanitizer demonstration only (tests/test_sanitizers.py, oracle/f6_demo.c): the reference's index expression
             * vRowIndices[yi] (src/Frame.cc:934-941) on a table sized by mvImagePyramid[0].rows (:910), no bound check */
            rcount[yi + 1]++;
#else
            if (yi >= 0 && yi < nRows) rcount[yi + 1]++;
#endif
    }
    for (int y = 0; y < nRows; ++y) rcount[y + 1] += rcount[y];
    int *ritems = (int *)malloc(sizeof(int) * (rcount[nRows] > 0 ? rcount[nRows] : 1));
    int *rfill = (int *)calloc(nRows, sizeof(int));
    for (int iR = 0; iR < nR; ++iR) {
        const float kpY = kR[iR].y;
        const float r = 2.0f * scale[kR[iR].octave];
        const int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
        for (int yi = minr; yi <= maxr; ++yi)
            if (yi >= 0 && yi < nRows) ritems[rcount[yi] + rfill[yi]++] = iR;
    }
    free(rfill);
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    distidx *vDistIdx = (distidx *)malloc(sizeof(distidx) * (nL > 0 ? nL : 1));
    int nd = 0;
    for (int iL = 0; iL < nL; ++iL) {
        const orc_keypoint *kpL = &kL[iL];
        const int levelL = kpL->octave;
        const float vL = kpL->y, uL = kpL->x;
        const long row = (long)vL;
        if (row < 0 || row >= nRows) continue; /* F6 clamp */
        const int cb = rcount[row], ce = rcount[row + 1];
        if (cb == ce) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH;
        int bestIdxR = 0;
        for (int c = cb; c < ce; ++c) {
            const int iR = ritems[c];
            const orc_keypoint *kpR = &kR[iR];
            if (kpR->octave < levelL - 1 || kpR->octave > levelL + 1) continue;
            const float uR = kpR->x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_descriptor_distance(dL + (size_t)iL * 32, dR + (size_t)iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = kR[bestIdxR].x;
            const float scaleFactor = inv_scale[kpL->octave];
            const float scaleduL = roundf(kpL->x * scaleFactor);
            const float scaledvL = roundf(kpL->y * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5, L = 5;
            const int lv = kpL->octave;
            const int W = pw[lv], H = ph[lv];
            const int y0 = (int)(scaledvL - w), x0 = (int)(scaleduL - w);
            /* cv::Mat::rowRange/colRange would assert outside the image: treated as "no match" */
            if (y0 < 0 || y0 + 2 * w + 1 > H || x0 < 0 || x0 + 2 * w + 1 > W) continue;
            const uint8_t *IL = pyrL[lv] + (size_t)y0 * W + x0;
            const float cL = (float)IL[w * W + w];
            int bestDistS = INT_MAX, bestincR = 0;
            float vDists[2 * 5 + 1];
            const float iniu = scaleduR0 - L - w; /* fork: minus (src/Frame.cc:1067) */
            const float endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= W) continue;
            for (int incR = -L; incR <= +L; incR++) {
                const uint8_t *IR = pyrR[lv] + (size_t)y0 * W + ((int)scaleduR0 + incR - w);
                const float cR = (float)IR[w * W + w];
                double acc = 0; /* cv::norm(NORM_L1) accumulates in double; operands are integer valued => exact */
                for (int yy = 0; yy < 2 * w + 1; ++yy)
                    for (int xx = 0; xx < 2 * w + 1; ++xx) {
                        const float a = (float)IL[yy * W + xx] - cL, b = (float)IR[yy * W + xx] - cR;
                        acc += fabsf(a - b);
                    }
                const float dist = (float)acc;
                if (dist < bestDistS) { bestDistS = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = scale[kpL->octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
                depth[iL] = mbf / disparity;
                uRight[iL] = bestuR;
                vDistIdx[nd].dist = bestDistS; vDistIdx[nd].idx = iL; nd++;
            }
        }
    }
    int nmatch = nd;
    if (nd > 0) {
        qsort(vDistIdx, nd, sizeof(distidx), distidx_cmp);
        const float median = (float)vDistIdx[nd / 2].dist;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nd - 1; i >= 0; i--) {
            if (vDistIdx[i].dist < thDist) break;
            uRight[vDistIdx[i].idx] = -1; depth[vDistIdx[i].idx] = -1; nmatch--;
        }
    }
    free(vDistIdx); free(rcount); free(ritems);
    return nmatch;
}

/* ------------------------------------------------------------------ SearchByProjection(Frame&, const Frame&, th, bMono) */
/* src/ORBmatcher.cc:1702-1871 (caller src/Tracking.cc:1430,1445).  MapPoint state is passed as arrays:
 *   last frame : has_mp[i] (pMP && !mvbOutlier[i]), world position xw[3*i], representative descriptor mpdesc[32*i]
 *                (pMP->GetDescriptor()), obs[i] = pMP->Observations(), keypoints kl (octave, angle of mvKeysUn)
 *   current    : keypoints kc, descriptors dc, u_right[] (mvuRight), Tcw, intrinsics, bounds; its mvpMapPoints are
 *                all NULL on entry (Tracking fills them with NULL before the call, src/Tracking.cc:1420) and the
 *                result is matched_last[i2] = index of the last-frame point assigned to current feature i2, or -1.
 * OpenCV boundary (parity unpinned): Rcw * x3Dw + tcw is cv::gemm's 3x3 * 3x1 float special case:
 *   t = a0*b0 + a1*b1 + a2*b2 in float (left to right), result = (float)((double)t + (double)tcw_i).
 * fp_mode: `fx*xc*invzc + cx` and `u - mbf*invzc` are contracted by g++ -O3 -march=native (reference flags) into
 *   fma(fx*xc, invzc, cx) and fma(-mbf, invzc, u); STRICT evaluates them with separate roundings. */
static float gemm3(const float *a, const float *b, float c) {
    float t = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    return (float)((double)t * 1.0 + (double)c * 1.0);
}

int orc_search_by_projection_ff(const orc_keypoint *kc, const uint8_t *dc, const float *u_right, int nc,
                                const float *Tcw /*4x4 row major*/, float fx, float fy, float cx, float cy, float minx,
                                float maxx, float miny, float maxy, float mb, float mbf, const float *scale_factors,
                                const orc_keypoint *kl, int nl, const uint8_t *has_mp, const float *xw,
                                const uint8_t *mpdesc, const int *obs, const float *Tlw, float th, int bMono,
                                int check_ori, int fp_mode, int *matched_last) {
    int nmatches = 0;
    for (int i = 0; i < nc; ++i) matched_last[i] = -1;
    int *hist = (int *)malloc(sizeof(int) * HISTO_LENGTH * (nl > 0 ? nl : 1));
    int hn[HISTO_LENGTH] = {0};
    const float factor = HISTO_LENGTH / 360.0f; /* fork: :1713 */
    float Rcw[9], tcw[3], Rlw[9], tlw[3], twc[3], tlc[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) { Rcw[3 * r + c] = Tcw[4 * r + c]; Rlw[3 * r + c] = Tlw[4 * r + c]; }
        tcw[r] = Tcw[4 * r + 3]; tlw[r] = Tlw[4 * r + 3];
    }
    for (int r = 0; r < 3; ++r) { /* twc = -Rcw.t() * tcw : gemm with alpha = -1 */
        float t = Rcw[0 + r] * tcw[0] + Rcw[3 + r] * tcw[1] + Rcw[6 + r] * tcw[2];
        twc[r] = (float)((double)t * -1.0);
    }
    for (int r = 0; r < 3; ++r) tlc[r] = gemm3(&Rlw[3 * r], twc, tlw[r]);
    const int bForward = tlc[2] > mb && !bMono;
    const int bBackward = -tlc[2] > mb && !bMono;
    orc_grid *g = orc_grid_build(kc, nc, minx, maxx, miny, maxy);
    int *cands = (int *)malloc(sizeof(int) * (nc > 0 ? nc : 1));
    for (int i = 0; i < nl; i++) {
        if (!has_mp[i]) continue;
        float x3Dc[3];
        for (int r = 0; r < 3; ++r) x3Dc[r] = gemm3(&Rcw[3 * r], &xw[3 * i], tcw[r]);
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = (float)(1.0 / x3Dc[2]);
        if (invzc < 0) continue;
        float u, v;
        if (fp_mode == ORC_FP_GCC_FMA) { u = fmaf(fx * xc, invzc, cx); v = fmaf(fy * yc, invzc, cy); }
        else { u = fx * xc * invzc + cx; v = fy * yc * invzc + cy; }
        if (u < minx || u > maxx) continue;
        if (v < miny || v > maxy) continue;
        const int nLastOctave = kl[i].octave;
        const float radius = th * scale_factors[nLastOctave];
        int ncand;
        if (bForward) ncand = orc_grid_query(g, u, v, radius, nLastOctave, -1, cands, nc);
        else if (bBackward) ncand = orc_grid_query(g, u, v, radius, 0, nLastOctave, cands, nc);
        else ncand = orc_grid_query(g, u, v, radius, nLastOctave - 1, nLastOctave + 1, cands, nc);
        if (ncand == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < ncand; ++c) {
            const int i2 = cands[c];
            if (matched_last[i2] >= 0 && obs[matched_last[i2]] > 0) continue;
            if (u_right[i2] > 0) {
                const float ur = fp_mode == ORC_FP_GCC_FMA ? fmaf(-mbf, invzc, u) : u - mbf * invzc;
                const float er = fabsf(ur - u_right[i2]);
                if (er > radius) continue;
            }
            const int dist = orc_descriptor_distance(mpdesc + (size_t)i * 32, dc + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            matched_last[bestIdx2] = i;
            nmatches++;
            if (check_ori) {
                float rot = kl[i].angle - kc[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                hist[bin * nl + hn[bin]++] = bestIdx2;
            }
        }
    }
    if (check_ori) {
        int ind1, ind2, ind3;
        orc_three_maxima(hn, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < hn[i]; j++) { matched_last[hist[i * nl + j]] = -1; nmatches--; }
    }
    orc_grid_free(g);
    free(hist); free(cands);
    return nmatches;
}

/* ------------------------------------------------------------------ SearchByProjection(Frame&, vector<MapPoint*>&, th) */
/* src/ORBmatcher.cc:69-184 (+ RadiusByViewingCos :187-194); caller Tracking::SearchLocalPoints src/Tracking.cc:1953.
 * MapPoint fields as arrays (filled by Frame::isInFrustum in the reference): in_view[i] = mbTrackInView && !isBad(),
 * proj[3*i] = (mTrackProjX, mTrackProjY, mTrackProjXR), level[i] = mnTrackScaleLevel, view_cos[i] = mTrackViewCos,
 * mpdesc = GetDescriptor(), mp_obs[i] = Observations().  Frame: keypoints, descriptors, u_right, and frame_obs[idx] =
 * Observations() of the MapPoint already attached to feature idx (-1: none).  Output assigned[idx] = index of the
 * MapPoint newly attached to feature idx, or -1. */
int orc_search_by_projection_mp(const orc_keypoint *kf, const uint8_t *df, const float *u_right, const int *frame_obs,
                                int nf, float minx, float maxx, float miny, float maxy, const float *scale_factors,
                                int nmp, const uint8_t *in_view, const float *proj, const int *level,
                                const float *view_cos, const uint8_t *mpdesc, const int *mp_obs, float th, float nnratio,
                                int *assigned) {
    int nmatches = 0;
    int *occ = (int *)malloc(sizeof(int) * (nf > 0 ? nf : 1)); /* Observations() of whatever sits at feature idx now */
    for (int i = 0; i < nf; ++i) { assigned[i] = -1; occ[i] = frame_obs[i]; }
    const int bFactor = th != 1.0;
    orc_grid *g = orc_grid_build(kf, nf, minx, maxx, miny, maxy);
    int *cands = (int *)malloc(sizeof(int) * (nf > 0 ? nf : 1));
    for (int iMP = 0; iMP < nmp; iMP++) {
        if (!in_view[iMP]) continue;
        const int nPredictedLevel = level[iMP];
        float r = view_cos[iMP] > 0.998 ? 2.5f : 4.0f;
        if (bFactor) r *= th;
        const int nc = orc_grid_query(g, proj[3 * iMP], proj[3 * iMP + 1], r * scale_factors[nPredictedLevel],
                                      nPredictedLevel - 1, nPredictedLevel, cands, nf);
        if (nc == 0) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < nc; ++c) {
            const int idx = cands[c];
            if (occ[idx] > 0) continue; /* F.mvpMapPoints[idx] && Observations() > 0 */
            if (u_right[idx] > 0) {
                const float er = fabsf(proj[3 * iMP + 2] - u_right[idx]);
                if (er > r * scale_factors[nPredictedLevel]) continue;
            }
            const int dist = orc_descriptor_distance(mpdesc + (size_t)iMP * 32, df + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kf[idx].octave; bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = kf[idx].octave; bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            assigned[bestIdx] = iMP;
            occ[bestIdx] = mp_obs[iMP];
            nmatches++;
        }
    }
    orc_grid_free(g);
    free(occ); free(cands);
    return nmatches;
}

/* ------------------------------------------------------------------------------------------------
 * BoW-guided policies.  DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>, ascending node id) is passed
 * flattened: node ids (ascending), nn + 1 offsets, feature indices in the vector's own order.  Only matching node
 * ids are visited; map::lower_bound of the reference's merge walk is the same as stepping through the sorted ids.
 * ---------------------------------------------------------------------------------------------- */
static void bow_rot_bin_push(int **hist, int *hn, float a1, float a2, int value) {
    float rot = a1 - a2;                       /* src/ORBmatcher.cc:340-351 */
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * (HISTO_LENGTH / 360.0f));
    if (bin == HISTO_LENGTH) bin = 0;
    if (bin >= 0 && bin < HISTO_LENGTH) hist[bin][hn[bin]++] = value;
}

/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (src/ORBmatcher.cc:248-410).
 * matched_kf[nf]: index of the keyframe feature whose MapPoint was attached to frame feature i, -1 = NULL. */
int orc_search_by_bow_kf_frame(const orc_keypoint *kkf, const uint8_t *dkf, int nkf, const uint8_t *kf_has_mp,
                               int nn_kf, const uint32_t *node_kf, const int *beg_kf, const uint32_t *idx_kf,
                               const orc_keypoint *kf_, const uint8_t *df, int nf, int nn_f, const uint32_t *node_f,
                               const int *beg_f, const uint32_t *idx_f, float nnratio, int check_ori, int *matched_kf) {
    (void)nkf;
    int nmatches = 0;
    int *hist[HISTO_LENGTH], hn[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; ++i) { hist[i] = (int *)malloc(sizeof(int) * (size_t)(nf + 1)); hn[i] = 0; }
    for (int i = 0; i < nf; ++i) matched_kf[i] = -1;
    int a = 0, b = 0;
    while (a < nn_kf && b < nn_f) {
        if (node_kf[a] == node_f[b]) {
            for (int ik = beg_kf[a]; ik < beg_kf[a + 1]; ++ik) {
                const unsigned realIdxKF = idx_kf[ik];
                if (!kf_has_mp[realIdxKF]) continue;                    /* !pMP || pMP->isBad()  (:285-291) */
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int jf = beg_f[b]; jf < beg_f[b + 1]; ++jf) {
                    const unsigned realIdxF = idx_f[jf];
                    if (matched_kf[realIdxF] >= 0) continue;             /* :305 */
                    const int dist = orc_descriptor_distance(dkf + 32 * (size_t)realIdxKF, df + 32 * (size_t)realIdxF);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= TH_LOW && (float)bestDist1 < nnratio * (float)bestDist2) {   /* :328-331 */
                    matched_kf[bestIdxF] = (int)realIdxKF;
                    if (check_ori) bow_rot_bin_push(hist, hn, kkf[realIdxKF].angle, kf_[bestIdxF].angle, bestIdxF);
                    nmatches++;
                }
            }
            a++; b++;
        } else if (node_kf[a] < node_f[b]) {
            while (a < nn_kf && node_kf[a] < node_f[b]) a++;            /* lower_bound */
        } else {
            while (b < nn_f && node_f[b] < node_kf[a]) b++;
        }
    }
    if (check_ori) {
        int sizes[HISTO_LENGTH], i1, i2, i3;
        for (int i = 0; i < HISTO_LENGTH; ++i) sizes[i] = hn[i];
        orc_three_maxima(sizes, HISTO_LENGTH, &i1, &i2, &i3);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int j = 0; j < hn[i]; ++j) { matched_kf[hist[i][j]] = -1; nmatches--; }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; ++i) free(hist[i]);
    return nmatches;
}

/* ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (src/ORBmatcher.cc:722-866): loop-closure candidates.
 * matches12[n1]: index of the KF2 feature whose MapPoint is vpMatches12[i1], -1 = NULL. */
int orc_search_by_bow_kf_kf(const orc_keypoint *k1, const uint8_t *d1, int n1, const uint8_t *has_mp1, int nn1,
                            const uint32_t *node1, const int *beg1, const uint32_t *idx1v, const orc_keypoint *k2,
                            const uint8_t *d2, int n2, const uint8_t *has_mp2, int nn2, const uint32_t *node2,
                            const int *beg2, const uint32_t *idx2v, float nnratio, int check_ori, int *matches12) {
    int nmatches = 0;
    int *hist[HISTO_LENGTH], hn[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; ++i) { hist[i] = (int *)malloc(sizeof(int) * (size_t)(n1 + 1)); hn[i] = 0; }
    uint8_t *matched2 = (uint8_t *)calloc((size_t)n2 + 1, 1);
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    int a = 0, b = 0;
    while (a < nn1 && b < nn2) {
        if (node1[a] == node2[b]) {
            for (int p = beg1[a]; p < beg1[a + 1]; ++p) {
                const unsigned i1 = idx1v[p];
                if (!has_mp1[i1]) continue;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int q = beg2[b]; q < beg2[b + 1]; ++q) {
                    const unsigned i2 = idx2v[q];
                    if (matched2[i2] || !has_mp2[i2]) continue;          /* :779-785 */
                    const int dist = orc_descriptor_distance(d1 + 32 * (size_t)i1, d2 + 32 * (size_t)i2);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = (int)i2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < TH_LOW && (float)bestDist1 < nnratio * (float)bestDist2) {    /* strict '<' here (:802) */
                    matches12[i1] = bestIdx2;
                    matched2[bestIdx2] = 1;
                    if (check_ori) bow_rot_bin_push(hist, hn, k1[i1].angle, k2[bestIdx2].angle, (int)i1);
                    nmatches++;
                }
            }
            a++; b++;
        } else if (node1[a] < node2[b]) {
            while (a < nn1 && node1[a] < node2[b]) a++;
        } else {
            while (b < nn2 && node2[b] < node1[a]) b++;
        }
    }
    if (check_ori) {
        int sizes[HISTO_LENGTH], i1, i2, i3;
        for (int i = 0; i < HISTO_LENGTH; ++i) sizes[i] = hn[i];
        orc_three_maxima(sizes, HISTO_LENGTH, &i1, &i2, &i3);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int j = 0; j < hn[i]; ++j) { matches12[hist[i][j]] = -1; nmatches--; }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; ++i) free(hist[i]);
    free(matched2);
    return nmatches;
}

/* ORBmatcher::CheckDistEpipolarLine (src/ORBmatcher.cc:206-233).  F12 row major 3x3.  The comparison is in double:
 * 3.84 is a double literal. */
static int orc_check_dist_epipolar(const orc_keypoint *kp1, const orc_keypoint *kp2, const float *F12, float sigma2,
                                   int fp_mode) {
    float a, b, c, num, den;
    if (fp_mode == ORC_FP_GCC_FMA) {   /* g++ -O3 -march=native: the first product of each sum is fused */
        a = fmaf(kp1->x, F12[0], kp1->y * F12[3]) + F12[6];
        b = fmaf(kp1->x, F12[1], kp1->y * F12[4]) + F12[7];
        c = fmaf(kp1->x, F12[2], kp1->y * F12[5]) + F12[8];
        num = fmaf(a, kp2->x, b * kp2->y) + c;
        den = fmaf(a, a, b * b);
    } else {
        a = kp1->x * F12[0] + kp1->y * F12[3] + F12[6];
        b = kp1->x * F12[1] + kp1->y * F12[4] + F12[7];
        c = kp1->x * F12[2] + kp1->y * F12[5] + F12[8];
        num = a * kp2->x + b * kp2->y + c;
        den = a * a + b * b;
    }
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return (double)dsqr < 3.84 * (double)sigma2;
}

/* ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:879-1087; fork: vbMatched2 bookkeeping :1022,1069).
 * (ex, ey) = epipole of KF1's centre in KF2 (:892-898, computed by the caller with its pose matrices).
 * matches12[n1]: matched KF2 feature or -1; the reference's vMatchedPairs is (i, matches12[i]) for ascending i. */
int orc_search_for_triangulation(const orc_keypoint *k1, const uint8_t *d1, int n1, const uint8_t *has_mp1,
                                 const float *ur1, int nn1, const uint32_t *node1, const int *beg1,
                                 const uint32_t *idx1v, const orc_keypoint *k2, const uint8_t *d2, int n2,
                                 const uint8_t *has_mp2, const float *ur2, int nn2, const uint32_t *node2,
                                 const int *beg2, const uint32_t *idx2v, const float *F12, float ex, float ey,
                                 const float *scale_factors2, const float *level_sigma2_2, int only_stereo,
                                 int check_ori, int fp_mode, int *matches12) {
    int nmatches = 0;
    int *hist[HISTO_LENGTH], hn[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; ++i) { hist[i] = (int *)malloc(sizeof(int) * (size_t)(n1 + 1)); hn[i] = 0; }
    uint8_t *matched2 = (uint8_t *)calloc((size_t)n2 + 1, 1);
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    int a = 0, b = 0;
    while (a < nn1 && b < nn2) {
        if (node1[a] == node2[b]) {
            for (int p = beg1[a]; p < beg1[a + 1]; ++p) {
                const unsigned i1 = idx1v[p];
                if (has_mp1[i1]) continue;                               /* already triangulated (:929-932) */
                const int stereo1 = ur1[i1] >= 0;
                if (only_stereo && !stereo1) continue;
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int q = beg2[b]; q < beg2[b + 1]; ++q) {
                    const unsigned i2 = idx2v[q];
                    if (matched2[i2] || has_mp2[i2]) continue;
                    const int stereo2 = ur2[i2] >= 0;
                    if (only_stereo && !stereo2) continue;
                    const int dist = orc_descriptor_distance(d1 + 32 * (size_t)i1, d2 + 32 * (size_t)i2);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const orc_keypoint *kp2 = &k2[i2];
                    if (!stereo1 && !stereo2) {
                        const float distex = ex - kp2->x, distey = ey - kp2->y;
                        const float d2e = fp_mode == ORC_FP_GCC_FMA ? fmaf(distex, distex, distey * distey)
                                                                    : distex * distex + distey * distey;
                        if (d2e < 100 * scale_factors2[kp2->octave]) continue;
                    }
                    if (orc_check_dist_epipolar(&k1[i1], kp2, F12, level_sigma2_2[kp2->octave], fp_mode)) {
                        bestIdx2 = (int)i2;
                        bestDist = dist;
                    }
                }
                if (bestIdx2 >= 0) {
                    matches12[i1] = bestIdx2;
                    matched2[bestIdx2] = 1;
                    nmatches++;
                    if (check_ori) bow_rot_bin_push(hist, hn, k1[i1].angle, k2[bestIdx2].angle, (int)i1);
                }
            }
            a++; b++;
        } else if (node1[a] < node2[b]) {
            while (a < nn1 && node1[a] < node2[b]) a++;
        } else {
            while (b < nn2 && node2[b] < node1[a]) b++;
        }
    }
    if (check_ori) {
        int sizes[HISTO_LENGTH], i1, i2, i3;
        for (int i = 0; i < HISTO_LENGTH; ++i) sizes[i] = hn[i];
        orc_three_maxima(sizes, HISTO_LENGTH, &i1, &i2, &i3);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int j = 0; j < hn[i]; ++j) {
                matched2[matches12[hist[i][j]]] = 0;
                matches12[hist[i][j]] = -1;
                nmatches--;
            }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; ++i) free(hist[i]);
    free(matched2);
    return nmatches;
}

/* ------------------------------------------------------------------------------------------------
 * Projection-guided policies of the back-end.  The pose algebra in front of them (cv::Mat products, cv::norm,
 * MapPoint::PredictScale) is OpenCV / libm code that stays in the reference's own translation unit; these functions
 * start where the reference has, for every MapPoint, the flag "passed every geometric test", the projection (u, v[, ur]),
 * the predicted level and the representative descriptor.
 * ---------------------------------------------------------------------------------------------- */

/* ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) src/ORBmatcher.cc:1100-1280, from :1168 (radius) to the
 * selection of bestIdx (:1180-1245).  best_idx[i] = feature of the keyframe the MapPoint is fused into (the caller then
 * runs Replace / AddObservation / AddMapPoint in order, which does not feed back into the selection), -1 = none.
 * Returns nFused. */
int orc_fuse(const orc_keypoint *kk, const uint8_t *dk, const float *ur_k, int nk, float minx, float maxx, float miny,
             float maxy, const float *scale_factors, const float *inv_level_sigma2, int np, const uint8_t *valid,
             const float *uv, const float *ur, const int *level, const uint8_t *desc, float th, int fp_mode,
             int *best_idx) {
    orc_grid *g = orc_grid_build(kk, nk, minx, maxx, miny, maxy);
    int *cands = (int *)malloc(sizeof(int) * (size_t)(nk + 1));
    int nFused = 0;
    for (int i = 0; i < np; ++i) {
        best_idx[i] = -1;
        if (!valid[i]) continue;
        const float u = uv[2 * i], v = uv[2 * i + 1];
        const int nPredictedLevel = level[i];
        const float radius = th * scale_factors[nPredictedLevel];
        const int nc = orc_grid_query(g, u, v, radius, -1, -1, cands, nk);   /* KeyFrame::GetFeaturesInArea: no level filter */
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nc; ++c) {
            const int idx = cands[c];
            const orc_keypoint *kp = &kk[idx];
            const int kpLevel = kp->octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (ur_k[idx] >= 0) {                                          /* stereo observation: 3-dof chi2 (:1198-1212) */
                const float ex = u - kp->x, ey = v - kp->y, er = ur[i] - ur_k[idx];
                const float e2 = fp_mode == ORC_FP_GCC_FMA ? fmaf(er, er, fmaf(ex, ex, ey * ey)) : ex * ex + ey * ey + er * er;
                if ((double)(e2 * inv_level_sigma2[kpLevel]) > 7.8) continue;
            } else {
                const float ex = u - kp->x, ey = v - kp->y;
                const float e2 = fp_mode == ORC_FP_GCC_FMA ? fmaf(ex, ex, ey * ey) : ex * ex + ey * ey;
                if ((double)(e2 * inv_level_sigma2[kpLevel]) > 5.99) continue;
            }
            const int dist = orc_descriptor_distance(desc + 32 * (size_t)i, dk + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) { best_idx[i] = bestIdx; nFused++; }
    }
    free(cands);
    orc_grid_free(g);
    return nFused;
}

/* Shared by Fuse(KF, Scw, ...) :1282-1430 and SearchByProjection(KF, Scw, vpPoints, vpMatched, th) :415-560:
 * level band [pred - 1, pred], no chi2 gate, bestDist <= TH_LOW.  `taken` (may be NULL) = vpMatched[idx] != NULL,
 * updated as the reference updates vpMatched (the order of the points matters). */
static int orc_project_and_pick(const orc_keypoint *kk, const uint8_t *dk, int nk, float minx, float maxx, float miny,
                                float maxy, const float *scale_factors, int np, const uint8_t *valid, const float *uv,
                                const int *level, const uint8_t *desc, float th, int init_best, int max_dist,
                                uint8_t *taken, int *best_idx) {
    orc_grid *g = orc_grid_build(kk, nk, minx, maxx, miny, maxy);
    int *cands = (int *)malloc(sizeof(int) * (size_t)(nk + 1));
    int n = 0;
    for (int i = 0; i < np; ++i) {
        best_idx[i] = -1;
        if (!valid[i]) continue;
        const int nPredictedLevel = level[i];
        const float radius = th * scale_factors[nPredictedLevel];
        const int nc = orc_grid_query(g, uv[2 * i], uv[2 * i + 1], radius, -1, -1, cands, nk);
        int bestDist = init_best, bestIdx = -1;
        for (int c = 0; c < nc; ++c) {
            const int idx = cands[c];
            if (taken && taken[idx]) continue;
            const int kpLevel = kk[idx].octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            const int dist = orc_descriptor_distance(desc + 32 * (size_t)i, dk + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= max_dist) {
            best_idx[i] = bestIdx;
            if (taken) taken[bestIdx] = 1;
            n++;
        }
    }
    free(cands);
    orc_grid_free(g);
    return n;
}

/* ORBmatcher::Fuse(KeyFrame*, cv::Mat Scw, vpPoints, th, vpReplacePoint) :1282-1430 */
int orc_fuse_sim3(const orc_keypoint *kk, const uint8_t *dk, int nk, float minx, float maxx, float miny, float maxy,
                  const float *scale_factors, int np, const uint8_t *valid, const float *uv, const int *level,
                  const uint8_t *desc, float th, int *best_idx) {
    return orc_project_and_pick(kk, dk, nk, minx, maxx, miny, maxy, scale_factors, np, valid, uv, level, desc, th, INT_MAX,
                                TH_LOW, NULL, best_idx);
}

/* ORBmatcher::SearchByProjection(KeyFrame*, cv::Mat Scw, vpPoints, vpMatched, int th) :415-560; matched[nk] in/out */
int orc_search_by_projection_sim3(const orc_keypoint *kk, const uint8_t *dk, int nk, float minx, float maxx, float miny,
                                  float maxy, const float *scale_factors, int np, const uint8_t *valid, const float *uv,
                                  const int *level, const uint8_t *desc, int th, uint8_t *matched, int *best_idx) {
    return orc_project_and_pick(kk, dk, nk, minx, maxx, miny, maxy, scale_factors, np, valid, uv, level, desc, (float)th, 256,
                                TH_LOW, matched, best_idx);
}

/* ORBmatcher::SearchBySim3 :1433-1690 from the two projection loops on: points of KF1 projected into KF2 (valid1 already
 * includes "has a good MapPoint and !vbAlreadyMatched1"), points of KF2 into KF1, TH_HIGH, then the mutual-agreement
 * pass.  matches12[n1] = KF2 feature whose MapPoint becomes vpMatches12[i1] (only NEW matches), -1 otherwise. */
int orc_search_by_sim3(const orc_keypoint *k1, const uint8_t *d1, int n1, const float *bounds1, const float *sf1,
                       const orc_keypoint *k2, const uint8_t *d2, int n2, const float *bounds2, const float *sf2,
                       const uint8_t *valid1, const float *uv1in2, const int *level1in2, const uint8_t *mpdesc1,
                       const uint8_t *valid2, const float *uv2in1, const int *level2in1, const uint8_t *mpdesc2, float th,
                       int *matches12) {
    int *m1 = (int *)malloc(sizeof(int) * (size_t)(n1 + 1)), *m2 = (int *)malloc(sizeof(int) * (size_t)(n2 + 1));
    orc_project_and_pick(k2, d2, n2, bounds2[0], bounds2[1], bounds2[2], bounds2[3], sf2, n1, valid1, uv1in2, level1in2,
                         mpdesc1, th, INT_MAX, TH_HIGH, NULL, m1);
    orc_project_and_pick(k1, d1, n1, bounds1[0], bounds1[1], bounds1[2], bounds1[3], sf1, n2, valid2, uv2in1, level2in1,
                         mpdesc2, th, INT_MAX, TH_HIGH, NULL, m2);
    int nFound = 0;
    for (int i1 = 0; i1 < n1; ++i1) {
        matches12[i1] = -1;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && m2[idx2] == i1) { matches12[i1] = idx2; nFound++; }
    }
    free(m1); free(m2);
    return nFound;
}

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist) :1873-2020 (relocalisation).
 * Points = pKF's MapPoints (valid: good, not already found, projection inside the image, distance in range); cur_has_mp[nc]
 * = CurrentFrame.mvpMapPoints[i2] != NULL, updated; matched_point[nc] = index of the point attached to feature i2, -1. */
int orc_search_by_projection_kf(const orc_keypoint *kc, const uint8_t *dc, int nc, float minx, float maxx, float miny,
                                float maxy, const float *scale_factors, int np, const uint8_t *valid, const float *uv,
                                const int *level, const uint8_t *desc, const float *kf_angle, float th, int orb_dist,
                                int check_ori, uint8_t *cur_has_mp, int *matched_point) {
    orc_grid *g = orc_grid_build(kc, nc, minx, maxx, miny, maxy);
    int *cands = (int *)malloc(sizeof(int) * (size_t)(nc + 1));
    int *hist[HISTO_LENGTH], hn[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; ++i) { hist[i] = (int *)malloc(sizeof(int) * (size_t)(nc + 1)); hn[i] = 0; }
    for (int i = 0; i < nc; ++i) matched_point[i] = -1;
    int nmatches = 0;
    for (int i = 0; i < np; ++i) {
        if (!valid[i]) continue;
        const int nPredictedLevel = level[i];
        const float radius = th * scale_factors[nPredictedLevel];
        const int ncand = orc_grid_query(g, uv[2 * i], uv[2 * i + 1], radius, nPredictedLevel - 1, nPredictedLevel + 1, cands, nc);
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < ncand; ++c) {
            const int i2 = cands[c];
            if (cur_has_mp[i2]) continue;
            const int dist = orc_descriptor_distance(desc + 32 * (size_t)i, dc + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= orb_dist) {
            cur_has_mp[bestIdx2] = 1;
            matched_point[bestIdx2] = i;
            nmatches++;
            if (check_ori) bow_rot_bin_push(hist, hn, kf_angle[i], kc[bestIdx2].angle, bestIdx2);
        }
    }
    if (check_ori) {
        int sizes[HISTO_LENGTH], i1, i2, i3;
        for (int i = 0; i < HISTO_LENGTH; ++i) sizes[i] = hn[i];
        orc_three_maxima(sizes, HISTO_LENGTH, &i1, &i2, &i3);
        for (int i = 0; i < HISTO_LENGTH; ++i) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int j = 0; j < hn[i]; ++j) { cur_has_mp[hist[i][j]] = 0; matched_point[hist[i][j]] = -1; nmatches--; }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; ++i) free(hist[i]);
    free(cands);
    orc_grid_free(g);
    return nmatches;
}

/* ------------------------------------------------------------------------------------------------
 * DBoW2: TemplatedVocabulary<FORB>::transform (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1240-1285 per feature,
 * :1136-1216 for the BowVector / FeatureVector; BowVector.cpp:40-95; FORB::distance FORB.cpp:81-101).
 * The tree is passed flattened (CSR of Node::children).  When the descent ends above level L - levelsup the reference
 * leaves *nid unassigned; 0 is returned here.
 * ---------------------------------------------------------------------------------------------- */
void orc_bow_transform(int n_nodes, int L, const int *child_begin, const uint32_t *child_ids, const uint8_t *node_desc,
                       const double *node_weight, const uint32_t *node_word, const uint8_t *desc, int n, int levelsup,
                       uint32_t *word_id, double *weight, uint32_t *node_id) {
    (void)n_nodes;
    const int nid_level = L - levelsup;
    for (int i = 0; i < n; ++i) {
        uint32_t final_id = 0, nid = 0;
        int current_level = 0;
        while (child_begin[final_id] != child_begin[final_id + 1]) {           /* do { } while (!isLeaf) on a non-empty tree */
            ++current_level;
            const int cb = child_begin[final_id], ce = child_begin[final_id + 1];
            uint32_t best = child_ids[cb];
            double best_d = (double)orc_descriptor_distance(desc + 32 * (size_t)i, node_desc + 32 * (size_t)best);
            for (int c = cb + 1; c < ce; ++c) {
                const double d = (double)orc_descriptor_distance(desc + 32 * (size_t)i, node_desc + 32 * (size_t)child_ids[c]);
                if (d < best_d) { best_d = d; best = child_ids[c]; }
            }
            final_id = best;
            if (current_level == nid_level) nid = final_id;
        }
        word_id[i] = node_word[final_id];
        weight[i] = node_weight[final_id];
        node_id[i] = nid;
    }
}

/* literal std::map emulation with sorted arrays + insertion, in feature order */
int orc_bow_vectors(int weighting, int scoring, const uint32_t *word_id, const double *weight, const uint32_t *node_id,
                    int n, uint32_t *bow_word, double *bow_value, int *n_bow, uint32_t *fv_node, int *fv_begin,
                    uint32_t *fv_index, int *n_fv_nodes) {
    int nb = 0, nn = 0;
    int *cnt = (int *)calloc((size_t)n + 1, sizeof(int));
    uint32_t *tmp_idx = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n + 1) * (size_t)(n > 0 ? 1 : 1));
    /* FeatureVector: per node a growing list; emulate with (node, feature) pairs kept sorted by node, stable */
    uint32_t *pn = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n + 1)), *pf = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n + 1));
    int np = 0;
    for (int i = 0; i < n; ++i) {
        if (!(weight[i] > 0)) continue;
        /* v.addWeight / addIfNotExist */
        int lo = 0;
        while (lo < nb && bow_word[lo] < word_id[i]) ++lo;                       /* lower_bound */
        if (lo < nb && bow_word[lo] == word_id[i]) {
            if (weighting == 0 || weighting == 1) bow_value[lo] += weight[i];
        } else {
            for (int j = nb; j > lo; --j) { bow_word[j] = bow_word[j - 1]; bow_value[j] = bow_value[j - 1]; }
            bow_word[lo] = word_id[i]; bow_value[lo] = weight[i]; ++nb;
        }
        /* fv.addFeature(nid, i_feature) */
        int p = np;
        while (p > 0 && pn[p - 1] > node_id[i]) { pn[p] = pn[p - 1]; pf[p] = pf[p - 1]; --p; }
        pn[p] = node_id[i]; pf[p] = (uint32_t)i; ++np;
    }
    const int norm = scoring == 1 ? 2 : scoring == 5 ? 0 : 1;                     /* ScoringObject::mustNormalize */
    if ((weighting == 0 || weighting == 1) && nb > 0 && norm == 0) {
        const double nd = (double)nb;
        for (int i = 0; i < nb; ++i) bow_value[i] /= nd;
    }
    if (norm) {
        double s = 0.0;
        if (norm == 1) for (int i = 0; i < nb; ++i) s += fabs(bow_value[i]);
        else { for (int i = 0; i < nb; ++i) s += bow_value[i] * bow_value[i]; s = sqrt(s); }
        if (s > 0.0) for (int i = 0; i < nb; ++i) bow_value[i] /= s;
    }
    fv_begin[0] = 0;
    for (int i = 0; i < np; ++i) {
        if (i == 0 || pn[i] != pn[i - 1]) { fv_node[nn] = pn[i]; ++nn; fv_begin[nn] = fv_begin[nn - 1]; }
        fv_index[fv_begin[nn]++] = pf[i];
    }
    *n_bow = nb; *n_fv_nodes = nn;
    free(cnt); free(tmp_idx); free(pn); free(pf);
    return nb;
}

/* ------------------------------------------------------------------------------------------------
 * Frame::UndistortKeyPoints (src/Frame.cc:770-825) / ComputeImageBounds (:830-865): cv::undistortPoints(src, dst, K, D,
 * R = empty, P = K) on float points.  OpenCV boundary, PARITY UNPINNED: restated from OpenCV 3.2 imgproc/undistort.cpp
 * (cvUndistortPoints): everything in double, five fixed iterations of the inverse distortion model, then P * (x, y, 1):
 *   x0 = x = (u - cx) / fx  (as  (u - cx) * (1. / fx)),  y likewise
 *   repeat 5:  r2 = x*x + y*y
 *              icdist = (1 + ((k7*r2 + k6)*r2 + k5)*r2) / (1 + ((k4*r2 + k1)*r2 + k0)*r2)
 *              dx = 2*k2*x*y + k3*(r2 + 2*x*x) + k8*r2 + k9*r2*r2 ;  dy = k2*(r2 + 2*y*y) + 2*k3*x*y + k10*r2 + k11*r2*r2
 *              x = (x0 - dx)*icdist ; y = (y0 - dy)*icdist
 *   xx = P00*x + P01*y + P02 ; yy = P10*x + P11*y + P12 ; ww = 1. / (P20*x + P21*y + P22) ; out = (float)(xx*ww), (float)(yy*ww)
 * k = (k1, k2, p1, p2, k3, 0...) as ORB-SLAM2 passes them (mDistCoef has 4 or 5 entries); the tilt model (k12, k13) is the
 * identity for zero tilt.  K: fx, fy, cx, cy (float, converted to double as cvConvert does).
 * ---------------------------------------------------------------------------------------------- */
void orc_undistort_points(const float *pts, int n, float fx, float fy, float cx, float cy, const float *dist, int ndist,
                          float *out) {
    double k[14] = {0};
    for (int i = 0; i < ndist && i < 14; ++i) k[i] = (double)dist[i];
    const double dfx = (double)fx, dfy = (double)fy, dcx = (double)cx, dcy = (double)cy;
    const double ifx = 1. / dfx, ify = 1. / dfy;
    for (int i = 0; i < n; ++i) {
        double x = (double)pts[2 * i], y = (double)pts[2 * i + 1];
        double x0 = x = (x - dcx) * ifx;
        double y0 = y = (y - dcy) * ify;
        for (int j = 0; j < 5; ++j) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        const double xx = dfx * x + 0.0 * y + dcx, yy = 0.0 * x + dfy * y + dcy;
        const double ww = 1. / (0.0 * x + 0.0 * y + 1.0);
        out[2 * i] = (float)(xx * ww);
        out[2 * i + 1] = (float)(yy * ww);
    }
}
/* mvKeysUn from mvKeys (identity copy when dist[0] == 0, :772-776) */
void orc_undistort_keypoints(const orc_keypoint *k, int n, float fx, float fy, float cx, float cy, const float *dist,
                             int ndist, orc_keypoint *out) {
    for (int i = 0; i < n; ++i) out[i] = k[i];
    if (ndist < 1 || dist[0] == 0.0f) return;
    for (int i = 0; i < n; ++i) {
        float p[2] = {k[i].x, k[i].y}, q[2];
        orc_undistort_points(p, 1, fx, fy, cx, cy, dist, ndist, q);
        out[i].x = q[0]; out[i].y = q[1];
    }
}
/* ComputeImageBounds: (mnMinX, mnMaxX, mnMinY, mnMaxY) */
void orc_image_bounds(int cols, int rows, float fx, float fy, float cx, float cy, const float *dist, int ndist, float *b4) {
    if (ndist >= 1 && dist[0] != 0.0f) {
        float p[8] = {0.f, 0.f, (float)cols, 0.f, 0.f, (float)rows, (float)cols, (float)rows}, q[8];
        orc_undistort_points(p, 4, fx, fy, cx, cy, dist, ndist, q);
        b4[0] = q[0] < q[4] ? q[0] : q[4];     /* min(x of top-left, x of bottom-left) */
        b4[1] = q[2] > q[6] ? q[2] : q[6];
        b4[2] = q[1] < q[3] ? q[1] : q[3];
        b4[3] = q[5] > q[7] ? q[5] : q[7];
    } else {
        b4[0] = 0.0f; b4[1] = (float)cols; b4[2] = 0.0f; b4[3] = (float)rows;
    }
}

/* ------------------------------------------------------------------------------------------------
 * cv::remap(src, dst, map1 CV_32FC1, map2 CV_32FC1, INTER_LINEAR, BORDER_CONSTANT 0) for 8-bit single-channel images
 * (reference Examples/Stereo/stereo_euroc.cc:183-194: rectification of both EuRoC images before the extractor).
 * OpenCV boundary, PARITY UNPINNED: restated from OpenCV 3.2 imgproc/imgwarp.cpp:
 *   sx = cvRound(map1 * 32), sy = cvRound(map2 * 32)            (INTER_TAB_SIZE = 32; round half to even)
 *   ix = saturate_cast<short>(sx >> 5), iy likewise, fx = sx & 31, fy = sy & 31
 *   weights = BilinearTab_i[fy * 32 + fx] : saturate_cast<short>((1 - fy/32)(1 - fx/32) * 32768) etc. -- exact integers
 *             (32 - fx)(32 - fy) * 32 ..., except the entry fx = fy = 0: 32768 saturates to 32767 and initInterTab2D's
 *             residue correction puts the missing 1 on the [1][1] tap: {32767, 0, 0, 1}
 *   dst = (w00*S[iy][ix] + w01*S[iy][ix+1] + w10*S[iy+1][ix] + w11*S[iy+1][ix+1] + (1 << 14)) >> 15, taps outside the
 *         source read the border value 0.
 * ---------------------------------------------------------------------------------------------- */
static short orc_sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
void orc_remap_linear_u8(const uint8_t *src, int sw, int sh, int sstride, const float *mapx, const float *mapy, int dw,
                         int dh, uint8_t *dst, int dstride) {
    static short tab[32 * 32][4];
    static int init = 0;
    if (!init) {
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                const float vy[2] = {1.f - i * (1.f / 32), i * (1.f / 32)}, vx[2] = {1.f - j * (1.f / 32), j * (1.f / 32)};
                short *t = tab[i * 32 + j];
                int isum = 0;
                for (int k1 = 0; k1 < 2; ++k1)
                    for (int k2 = 0; k2 < 2; ++k2) {
                        const float v = vy[k1] * vx[k2] * 32768.f;
                        t[k1 * 2 + k2] = orc_sat_short((int)lrintf(v));
                        isum += t[k1 * 2 + k2];
                    }
                if (isum != 32768) t[3] = (short)(t[3] - (isum - 32768));   /* only (0, 0): {32767, 0, 0, 1} */
            }
        init = 1;
    }
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            const int sx = (int)lrintf(mapx[(size_t)y * dw + x] * 32.f), sy = (int)lrintf(mapy[(size_t)y * dw + x] * 32.f);
            const int ix = orc_sat_short(sx >> 5), iy = orc_sat_short(sy >> 5);
            const short *w = tab[(sy & 31) * 32 + (sx & 31)];
            int acc = 0;
            for (int k1 = 0; k1 < 2; ++k1)
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int xx = ix + k2, yy = iy + k1;
                    const int p = (xx >= 0 && xx < sw && yy >= 0 && yy < sh) ? src[(size_t)yy * sstride + xx] : 0;
                    acc += w[k1 * 2 + k2] * p;
                }
            const int v = (acc + (1 << 14)) >> 15;
            dst[(size_t)y * dstride + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
}
