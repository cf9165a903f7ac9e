/* orb_oracle.c -- CPU restatement of the reference ORB extractor path (see orb_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into / called from the product library.
 * PARITY UNPINNED at the OpenCV boundary (see orb_oracle.h header).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).  Implicit FMA
 * contraction is OFF for the whole file; the only contracted expressions of the reference
 * as built by its own flags (-O3 -march=native, CMakeLists.txt:10-11) are the two tap
 * coordinates of GET_VALUE (src/ORBextractor.cc:207-209) and they are written with an
 * explicit fmaf() under fp_mode == ORC_FP_GCC_FMA.
 */
#define _GNU_SOURCE
#include "orb_oracle.h"
#include "../include/orbx_pattern_data.h"
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define PATCH_SIZE 31       /* src/ORBextractor.cc:80 */
#define HALF_PATCH_SIZE 15  /* :81 */
#define EDGE_THRESHOLD 19   /* :82 */
#define MAX_LEVELS 32

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* ---------------------------------------------------------------- OpenCV scalar helpers */
/* cvRound: round-half-to-even (cvtss2si / lrint) */
int orc_cv_round_f(float v) { return (int)lrintf(v); }
static int cv_round_d(double v) { return (int)lrint(v); }
static int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
static int cv_ceil_f(float v) { int i = (int)v; return i + (i < v); }
static short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
static uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

/* cv::fastAtan2 (OpenCV 3.2 core/mathfuncs: atanImpl<float>), degrees */
float orc_fast_atan2(float y, float x) {
    static const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ---------------------------------------------------------------- copyMakeBorder REFLECT_101 */
static int reflect101(int i, int n) {
    /* cv::borderInterpolate(BORDER_REFLECT_101): gfedcb|abcdefgh|gfedcba */
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

void orc_border_reflect101(const uint8_t *src, int w, int h, int sstride, uint8_t *dst,
                           int dstride, int border) {
    int dw = w + 2 * border, dh = h + 2 * border;
    for (int y = 0; y < dh; ++y) {
        const uint8_t *srow = src + (size_t)reflect101(y - border, h) * sstride;
        uint8_t *drow = dst + (size_t)y * dstride;
        for (int x = 0; x < dw; ++x) drow[x] = srow[reflect101(x - border, w)];
    }
}

/* ---------------------------------------------------------------- cv::resize INTER_LINEAR 8UC1 */
/* OpenCV 3.2 imgproc/imgwarp.cpp: resize() -> resizeGeneric_ with HResizeLinear<uchar,int,short,2048>
 * and VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>.  */
void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw,
                          int dh, int dstride) {
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * dw);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * dw);
    int *T0 = (int *)malloc(sizeof(int) * dw), *T1 = (int *)malloc(sizeof(int) * dw);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_short(orc_cv_round_f((1.f - fx) * 2048));
        ialpha[2 * dx + 1] = sat_short(orc_cv_round_f(fx * 2048));
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        short b0 = sat_short(orc_cv_round_f((1.f - fy) * 2048));
        short b1 = sat_short(orc_cv_round_f(fy * 2048));
        int sy0 = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);           /* clip(sy, 0, sh) */
        int sy1 = sy + 1 < 0 ? 0 : (sy + 1 < sh ? sy + 1 : sh - 1);
        const uint8_t *S0 = src + (size_t)sy0 * sstride, *S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; ++dx) {
            int sx = xofs[dx];
            int sx1 = sx + 1 < sw ? sx + 1 : sw - 1; /* dx>=xmax: weight of 2nd tap is 0 */
            int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
            T0[dx] = S0[sx] * a0 + S0[sx1] * a1;
            T1[dx] = S1[sx] * a0 + S1[sx1] * a1;
        }
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; ++dx)
            D[dx] = (uint8_t)((((b0 * (T0[dx] >> 4)) >> 16) + ((b1 * (T1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(T0); free(T1);
}

/* ---------------------------------------------------------------- cv::FAST 9_16 + NMS */
static const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* OpenCV 3.2 features2d/fast_score.cpp cornerScore<16> (scalar branch) */
static int corner_score16(const uint8_t *ptr, const int *pixel, int threshold) {
    enum { K = 8, N = K * 3 + 1 };
    int k, v = ptr[0];
    short d[N];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        if (d[k + 3] < a) a = d[k + 3];
        if (a <= a0) continue;
        for (int j = 4; j <= 8; ++j) if (d[k + j] < a) a = d[k + j];
        int m = a < d[k] ? a : d[k];
        if (m > a0) a0 = m;
        m = a < d[k + 9] ? a : d[k + 9];
        if (m > a0) a0 = m;
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int j = 3; j <= 5; ++j) if (d[k + j] > b) b = d[k + j];
        if (b >= b0) continue;
        for (int j = 6; j <= 8; ++j) if (d[k + j] > b) b = d[k + j];
        int m = b > d[k] ? b : d[k];
        if (m < b0) b0 = m;
        m = b > d[k + 9] ? b : d[k + 9];
        if (m < b0) b0 = m;
    }
    return -b0 - 1;
}

int orc_fast9_16(const uint8_t *img, int w, int h, int stride, int threshold, orc_keypoint *out,
                 int cap) {
    enum { K = 8, N = 25 };
    int pixel[25];
    int n = 0;
    for (int k = 0; k < 16; ++k) pixel[k] = RING_DX[k] + RING_DY[k] * stride;
    for (int k = 16; k < 25; ++k) pixel[k] = pixel[k - 16];
    threshold = threshold < 0 ? 0 : threshold > 255 ? 255 : threshold;
    if (w < 7 || h < 7) return 0;
    uint8_t *buf = (uint8_t *)calloc((size_t)3 * w, 1);
    int *cp = (int *)malloc(sizeof(int) * 3 * (w + 1));
    uint8_t *sb[3] = {buf, buf + w, buf + 2 * w};
    int *cpb[3] = {cp + 1, cp + 1 + (w + 1), cp + 1 + 2 * (w + 1)};
    for (int i = 3; i < h - 2; ++i) {
        const uint8_t *ptr = img + (size_t)i * stride + 3;
        uint8_t *curr = sb[(i - 3) % 3];
        int *cornerpos = cpb[(i - 3) % 3];
        memset(curr, 0, w);
        int ncorners = 0;
        if (i < h - 3) {
            for (int j = 3; j < w - 3; ++j, ++ptr) {
                int v = ptr[0];
                int vt_lo = v - threshold, vt_hi = v + threshold;
                /* run-length test over the 25-long wrapped ring, as FAST_t<16> does */
                int count = 0, hit = 0;
                for (int k = 0; k < N; ++k) {
                    if (ptr[pixel[k]] < vt_lo) { if (++count > K) { hit = 1; break; } }
                    else count = 0;
                }
                if (!hit) {
                    count = 0;
                    for (int k = 0; k < N; ++k) {
                        if (ptr[pixel[k]] > vt_hi) { if (++count > K) { hit = 1; break; } }
                        else count = 0;
                    }
                }
                if (hit) {
                    cornerpos[ncorners++] = j;
                    curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t *prev = sb[(i - 4 + 3) % 3];
        const uint8_t *pprev = sb[(i - 5 + 3) % 3];
        cornerpos = cpb[(i - 4 + 3) % 3];
        ncorners = cornerpos[-1];
        for (int k = 0; k < ncorners; ++k) {
            int j = cornerpos[k];
            int score = prev[j];
            if (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] &&
                score > pprev[j] && score > pprev[j + 1] && score > curr[j - 1] &&
                score > curr[j] && score > curr[j + 1]) {
                if (n < cap) {
                    orc_keypoint kp = {(float)j, (float)(i - 1), 7.f, -1.f, (float)score, 0, -1};
                    out[n] = kp;
                }
                ++n;
            }
        }
    }
    free(buf); free(cp);
    return n;
}

/* ---------------------------------------------------------------- cv::GaussianBlur 7x7 sigma 2, 8U */
/* OpenCV 3.2 imgproc/smooth.cpp+filter.cpp: float kernel -> 8-bit fixed point {18,34,49,55,49,34,18};
 * row pass int32 (RowFilter<uchar,int>), column pass SymmColumnFilter<FixedPtCastEx<int,uchar>> whose
 * SSE2 vector op (SymmColumnVec_32s8u) handles x < (w & ~3) in FLOAT (kernel/2^16, cvtps round-to-
 * nearest-even) and leaves the last (w & 3) columns to the integer scalar tail ((s + 2^15) >> 16). */
static void gauss_kernel_fixed(int k[7]) {
    float cf[7];
    double sum = 0, scale2X = -0.5 / (2.0 * 2.0);
    for (int i = 0; i < 7; ++i) {
        double x = i - 3.0;
        double t = exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < 7; ++i) {
        cf[i] = (float)(cf[i] * sum);
        k[i] = orc_cv_round_f(cf[i] * 256.f);
    }
}

void orc_gaussian_blur7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride) {
    int k[7];
    gauss_kernel_fixed(k);
    int *rows = (int *)malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; ++y) {
        const uint8_t *s = src + (size_t)y * sstride;
        int *r = rows + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            int acc = 0;
            for (int i = 0; i < 7; ++i) acc += k[i] * s[reflect101(x + i - 3, w)];
            r[x] = acc;
        }
    }
    const float kf0 = k[3] / 65536.f, kf1 = k[2] / 65536.f, kf2 = k[1] / 65536.f, kf3 = k[0] / 65536.f;
    int wv = w & ~3;
    for (int y = 0; y < h; ++y) {
        const int *r0 = rows + (size_t)y * w;
        const int *rp1 = rows + (size_t)reflect101(y + 1, h) * w, *rm1 = rows + (size_t)reflect101(y - 1, h) * w;
        const int *rp2 = rows + (size_t)reflect101(y + 2, h) * w, *rm2 = rows + (size_t)reflect101(y - 2, h) * w;
        const int *rp3 = rows + (size_t)reflect101(y + 3, h) * w, *rm3 = rows + (size_t)reflect101(y - 3, h) * w;
        uint8_t *d = dst + (size_t)y * dstride;
        for (int x = 0; x < wv; ++x) {
            float s0 = (float)r0[x] * kf0 + 0.f;
            s0 = s0 + (float)(rp1[x] + rm1[x]) * kf1;
            s0 = s0 + (float)(rp2[x] + rm2[x]) * kf2;
            s0 = s0 + (float)(rp3[x] + rm3[x]) * kf3;
            d[x] = sat_u8((int)lrintf(s0));
        }
        for (int x = wv; x < w; ++x) {
            int s0 = k[3] * r0[x] + k[2] * (rp1[x] + rm1[x]) + k[1] * (rp2[x] + rm2[x]) +
                     k[0] * (rp3[x] + rm3[x]);
            d[x] = sat_u8((s0 + (1 << 15)) >> 16);
        }
    }
    free(rows);
}

/* ---------------------------------------------------------------- quadtree (literal list semantics) */
typedef struct qnode {
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy; /* include/ORBextractor.h:45-77 */
    int *keys; int nkeys;                        /* vKeys: indices into the candidate array, in order */
    int noMore;
    long seq;                                    /* creation order == address order under a bump allocator (F3) */
    struct qnode *prev, *next;
} qnode;

typedef struct qlist { qnode *head, *tail; int size; long next_seq; } qlist;

static qnode *qnode_new(qlist *L, int keycap) {
    qnode *n = (qnode *)calloc(1, sizeof(qnode));
    n->keys = (int *)malloc(sizeof(int) * (keycap > 0 ? keycap : 1));
    n->seq = L->next_seq++;
    return n;
}
static void qlist_push_front(qlist *L, qnode *n) {
    n->prev = NULL; n->next = L->head;
    if (L->head) L->head->prev = n; else L->tail = n;
    L->head = n; L->size++;
}
static void qlist_push_back(qlist *L, qnode *n) {
    n->next = NULL; n->prev = L->tail;
    if (L->tail) L->tail->next = n; else L->head = n;
    L->tail = n; L->size++;
}
static qnode *qlist_erase(qlist *L, qnode *n) { /* returns next */
    qnode *nx = n->next;
    if (n->prev) n->prev->next = n->next; else L->head = n->next;
    if (n->next) n->next->prev = n->prev; else L->tail = n->prev;
    L->size--;
    free(n->keys); free(n);
    return nx;
}

/* ExtractorNode::DivideNode  src/ORBextractor.cc:963-1035 */
static void divide_node(qlist *L, const qnode *p, const orc_keypoint *K, qnode *c[4]) {
    const int halfX = (int)ceilf((float)(p->URx - p->ULx) / 2);
    const int halfY = (int)ceilf((float)(p->BRy - p->ULy) / 2);
    for (int i = 0; i < 4; ++i) c[i] = qnode_new(L, p->nkeys);
    qnode *n1 = c[0], *n2 = c[1], *n3 = c[2], *n4 = c[3];
    n1->ULx = p->ULx; n1->ULy = p->ULy;
    n1->URx = p->ULx + halfX; n1->URy = p->ULy;
    n1->BLx = p->ULx; n1->BLy = p->ULy + halfY;
    n1->BRx = p->ULx + halfX; n1->BRy = p->ULy + halfY;
    n2->ULx = n1->URx; n2->ULy = n1->URy;
    n2->URx = p->URx; n2->URy = p->URy;
    n2->BLx = n1->BRx; n2->BLy = n1->BRy;
    n2->BRx = p->URx; n2->BRy = p->ULy + halfY;
    n3->ULx = n1->BLx; n3->ULy = n1->BLy;
    n3->URx = n1->BRx; n3->URy = n1->BRy;
    n3->BLx = p->BLx; n3->BLy = p->BLy;
    n3->BRx = n1->BRx; n3->BRy = p->BLy;
    n4->ULx = n3->URx; n4->ULy = n3->URy;
    n4->URx = n2->BRx; n4->URy = n2->BRy;
    n4->BLx = n3->BRx; n4->BLy = n3->BRy;
    n4->BRx = p->BRx; n4->BRy = p->BRy;
    for (int i = 0; i < p->nkeys; ++i) {
        const orc_keypoint *kp = &K[p->keys[i]];
        qnode *t;
        if (kp->x < n1->URx) t = (kp->y < n1->BRy) ? n1 : n3;
        else t = (kp->y < n1->BRy) ? n2 : n4;
        t->keys[t->nkeys++] = p->keys[i];
    }
    for (int i = 0; i < 4; ++i) if (c[i]->nkeys == 1) c[i]->noMore = 1;
}

typedef struct { int n; qnode *p; } szptr;
static int szptr_cmp(const void *a, const void *b) {
    const szptr *x = (const szptr *)a, *y = (const szptr *)b;
    if (x->n != y->n) return x->n < y->n ? -1 : 1;
    /* pair<int,ExtractorNode*> compares the pointer next: address order == creation order (spec, F3) */
    if (x->p->seq != y->p->seq) return x->p->seq < y->p->seq ? -1 : 1;
    return 0;
}

/* push children like src/ORBextractor.cc:1177-1248 / :1313-1355 */
static void push_children(qlist *L, qnode *c[4], szptr *vec, int *nvec, int *nToExpand) {
    for (int i = 0; i < 4; ++i) {
        if (c[i]->nkeys > 0) {
            qlist_push_front(L, c[i]);
            if (c[i]->nkeys > 1) {
                if (nToExpand) (*nToExpand)++;
                vec[*nvec].n = c[i]->nkeys; vec[*nvec].p = c[i]; (*nvec)++;
            }
        } else { free(c[i]->keys); free(c[i]); }
    }
}

/* ORBextractor::DistributeOctTree  src/ORBextractor.cc:1050-1417 */
int orc_distribute_octtree(const orc_keypoint *K, int nkeys, int minX, int maxX, int minY, int maxY,
                           int N, int *out_idx, int cap) {
    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
    if (nIni <= 0) return -3; /* reference divides by zero here (:1060-1063) */
    const float hX = (float)(maxX - minX) / nIni;
    qlist L = {0};
    qnode **ini = (qnode **)malloc(sizeof(qnode *) * nIni);
    for (int i = 0; i < nIni; ++i) {
        qnode *ni = qnode_new(&L, nkeys);
        ni->ULx = (int)(hX * (float)i); ni->ULy = 0;
        ni->URx = (int)(hX * (float)(i + 1)); ni->URy = 0;
        ni->BLx = ni->ULx; ni->BLy = maxY - minY;
        ni->BRx = ni->URx; ni->BRy = maxY - minY;
        qlist_push_back(&L, ni);
        ini[i] = ni;
    }
    for (int i = 0; i < nkeys; ++i) {
        int b = (int)(K[i].x / hX);
        if (b >= nIni) b = nIni - 1; /* reference would index out of bounds; cannot happen for valid input */
        ini[b]->keys[ini[b]->nkeys++] = i;
    }
    free(ini);
    for (qnode *it = L.head; it;) {
        if (it->nkeys == 1) { it->noMore = 1; it = it->next; }
        else if (it->nkeys == 0) it = qlist_erase(&L, it);
        else it = it->next;
    }
    int bFinish = 0;
    int veccap = 4 * (nkeys + 4 * nIni) + 16;
    szptr *vec = (szptr *)malloc(sizeof(szptr) * veccap);
    szptr *pvec = (szptr *)malloc(sizeof(szptr) * veccap);
    int nvec = 0;
    while (!bFinish) {
        int prevSize = L.size;
        int nToExpand = 0;
        nvec = 0;
        for (qnode *it = L.head; it;) {
            if (it->noMore) { it = it->next; continue; }
            qnode *c[4];
            divide_node(&L, it, K, c);
            push_children(&L, c, vec, &nvec, &nToExpand);
            it = qlist_erase(&L, it);
        }
        if (L.size >= N || L.size == prevSize) {
            bFinish = 1;
        } else if (L.size + nToExpand * 3 > N) {
            while (!bFinish) {
                prevSize = L.size;
                int npv = nvec;
                memcpy(pvec, vec, sizeof(szptr) * npv);
                nvec = 0;
                qsort(pvec, npv, sizeof(szptr), szptr_cmp);
                for (int j = npv - 1; j >= 0; --j) {
                    qnode *c[4];
                    divide_node(&L, pvec[j].p, K, c);
                    push_children(&L, c, vec, &nvec, NULL);
                    qlist_erase(&L, pvec[j].p);
                    if (L.size >= N) break;
                }
                if (L.size >= N || L.size == prevSize) bFinish = 1;
            }
        }
    }
    int n = 0;
    for (qnode *it = L.head; it; it = it->next) {
        int best = it->keys[0];
        float maxResponse = K[best].response;
        for (int k = 1; k < it->nkeys; ++k)
            if (K[it->keys[k]].response > maxResponse) { best = it->keys[k]; maxResponse = K[best].response; }
        if (n < cap) out_idx[n] = best;
        ++n;
    }
    for (qnode *it = L.head; it;) it = qlist_erase(&L, it);
    free(vec); free(pvec);
    return n;
}

/* ---------------------------------------------------------------- orientation */
static int g_umax[HALF_PATCH_SIZE + 1];
static int g_umax_ready = 0;
static void build_umax(int *umax) { /* src/ORBextractor.cc:866-910 */
    int v, v0;
    int vmax = cv_floor_f(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = cv_ceil_f(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

/* IC_Angle  src/ORBextractor.cc:104-161 */
float orc_ic_angle(const uint8_t *img, int stride, int x, int y) {
    if (!g_umax_ready) { build_umax(g_umax); g_umax_ready = 1; }
    int m_01 = 0, m_10 = 0;
    const uint8_t *center = img + (size_t)y * stride + x;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0, d = g_umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* ---------------------------------------------------------------- descriptor */
/* computeOrbDescriptor  src/ORBextractor.cc:177-254 */
void orc_descriptor(const uint8_t *blur, int stride, int x, int y, float angle_deg, int fp_mode,
                    uint8_t *desc) {
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = angle_deg * factorPI;
    float a = cosf(angle), b = sinf(angle);
    const uint8_t *center = blur + (size_t)y * stride + x;
    const signed char *pat = ORBX_PATTERN_I8;
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int j = 0; j < 8; ++j) {
            int t[2];
            for (int s = 0; s < 2; ++s) {
                float px = (float)pat[4 * j + 2 * s], py = (float)pat[4 * j + 2 * s + 1];
                int iy, ix;
                if (fp_mode == ORC_FP_GCC_FMA) {
                    /* g++ -O3 -march=native: first product fused, second rounded */
                    iy = orc_cv_round_f(fmaf(px, b, py * a));
                    ix = orc_cv_round_f(fmaf(px, a, -(py * b)));
                } else {
                    iy = orc_cv_round_f(px * b + py * a);
                    ix = orc_cv_round_f(px * a - py * b);
                }
                t[s] = center[iy * stride + ix];
            }
            val |= (t[0] < t[1]) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ---------------------------------------------------------------- extractor object */
struct orc_extractor {
    int nfeatures, nlevels, iniTh, minTh, fp_mode;
    double scaleFactor;
    float sf[MAX_LEVELS], inv[MAX_LEVELS], sigma2[MAX_LEVELS], invsigma2[MAX_LEVELS];
    int nPerLevel[MAX_LEVELS];
    int umax[HALF_PATCH_SIZE + 1];
    /* last-frame state */
    uint8_t *pyr[MAX_LEVELS], *blur[MAX_LEVELS];
    int pw[MAX_LEVELS], ph[MAX_LEVELS];
    orc_keypoint *cand[MAX_LEVELS]; int ncand[MAX_LEVELS];
    orc_keypoint *kps[MAX_LEVELS]; int nkps[MAX_LEVELS];
    double t[6];
};

orc_extractor *orc_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th,
                          int fp_mode) {
    if (nlevels < 1 || nlevels > MAX_LEVELS) return NULL;
    orc_extractor *e = (orc_extractor *)calloc(1, sizeof(*e));
    e->nfeatures = nfeatures; e->nlevels = nlevels; e->iniTh = ini_th; e->minTh = min_th;
    e->fp_mode = fp_mode;
    e->scaleFactor = scale_factor; /* member is a double (include/ORBextractor.h:234) */
    e->sf[0] = 1.0f; e->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; ++i) {
        e->sf[i] = (float)(e->sf[i - 1] * e->scaleFactor);
        e->sigma2[i] = e->sf[i] * e->sf[i];
    }
    for (int i = 0; i < nlevels; ++i) {
        e->inv[i] = 1.0f / e->sf[i];
        e->invsigma2[i] = 1.0f / e->sigma2[i];
    }
    float factor = (float)(1.0f / e->scaleFactor);
    float nd = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) {
        e->nPerLevel[l] = orc_cv_round_f(nd);
        sum += e->nPerLevel[l];
        nd *= factor;
    }
    e->nPerLevel[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    build_umax(e->umax);
    return e;
}

static void free_frame_state(orc_extractor *e) {
    for (int l = 0; l < e->nlevels; ++l) {
        free(e->pyr[l]); e->pyr[l] = NULL;
        free(e->blur[l]); e->blur[l] = NULL;
        free(e->cand[l]); e->cand[l] = NULL;
        free(e->kps[l]); e->kps[l] = NULL;
        e->ncand[l] = e->nkps[l] = 0;
    }
}

void orc_destroy(orc_extractor *e) {
    if (!e) return;
    free_frame_state(e);
    free(e);
}

void orc_get_tables(const orc_extractor *e, float *scale, float *inv_scale, float *sigma2,
                    float *inv_sigma2, int *fpl, int *umax16) {
    for (int i = 0; i < e->nlevels; ++i) {
        if (scale) scale[i] = e->sf[i];
        if (inv_scale) inv_scale[i] = e->inv[i];
        if (sigma2) sigma2[i] = e->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = e->invsigma2[i];
        if (fpl) fpl[i] = e->nPerLevel[i];
    }
    if (umax16) for (int i = 0; i <= HALF_PATCH_SIZE; ++i) umax16[i] = e->umax[i];
}

/* ComputePyramid  src/ORBextractor.cc:2093-2168 (fork: pyramid entries are the padded images) */
static void compute_pyramid(orc_extractor *e, const uint8_t *img, int w, int h, int stride) {
    int cols = w, rows = h; /* image.cols/rows never change inside the loop */
    for (int level = 0; level < e->nlevels; ++level) {
        float scale = e->inv[level];
        int sw = orc_cv_round_f((float)cols * scale), sh = orc_cv_round_f((float)rows * scale);
        int W = sw + EDGE_THRESHOLD * 2, H = sh + EDGE_THRESHOLD * 2;
        uint8_t *temp = (uint8_t *)malloc((size_t)W * H);
        if (level != 0) {
            uint8_t *centre = temp + (size_t)EDGE_THRESHOLD * W + EDGE_THRESHOLD;
            /* source is mvImagePyramid[level-1] == the previous PADDED image (:2166) */
            orc_resize_linear_u8(e->pyr[level - 1], e->pw[level - 1], e->ph[level - 1],
                                 e->pw[level - 1], centre, sw, sh, W);
            /* copyMakeBorder(centre -> temp, REFLECT_101 | ISOLATED): in place */
            uint8_t *tmpc = (uint8_t *)malloc((size_t)sw * sh);
            for (int y = 0; y < sh; ++y) memcpy(tmpc + (size_t)y * sw, centre + (size_t)y * W, sw);
            orc_border_reflect101(tmpc, sw, sh, sw, temp, W, EDGE_THRESHOLD);
            free(tmpc);
        } else {
            orc_border_reflect101(img, w, h, stride, temp, W, EDGE_THRESHOLD);
        }
        e->pyr[level] = temp; e->pw[level] = W; e->ph[level] = H;
    }
}

/* ComputeKeyPointsOctTree  src/ORBextractor.cc:1424-1601 */
static int compute_keypoints(orc_extractor *e) {
    const float W = 30;
    for (int level = 0; level < e->nlevels; ++level) {
        const int cols = e->pw[level], rows = e->ph[level];
        const uint8_t *im = e->pyr[level];
        const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
        const int maxBorderX = cols - EDGE_THRESHOLD + 3, maxBorderY = rows - EDGE_THRESHOLD + 3;
        const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        if (nCols <= 0 || nRows <= 0) { e->ncand[level] = 0; e->nkps[level] = 0; continue; }
        const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
        double t0 = now_s();
        int capc = 1024, nc = 0;
        orc_keypoint *cand = (orc_keypoint *)malloc(sizeof(orc_keypoint) * capc);
        int cellcap = (wCell + 6) * (hCell + 6);
        orc_keypoint *cell = (orc_keypoint *)malloc(sizeof(orc_keypoint) * cellcap);
        for (int i = 0; i < nRows; ++i) {
            const float iniY = (float)(minBorderY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; ++j) {
                const float iniX = (float)(minBorderX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = (float)maxBorderX;
                int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;
                const uint8_t *sub = im + (size_t)y0 * cols + x0;
                int n = orc_fast9_16(sub, x1 - x0, y1 - y0, cols, e->iniTh, cell, cellcap);
                if (n == 0) n = orc_fast9_16(sub, x1 - x0, y1 - y0, cols, e->minTh, cell, cellcap);
                for (int k = 0; k < n; ++k) {
                    cell[k].x += j * wCell;
                    cell[k].y += i * hCell;
                    if (nc == capc) { capc *= 2; cand = (orc_keypoint *)realloc(cand, sizeof(orc_keypoint) * capc); }
                    cand[nc++] = cell[k];
                }
            }
        }
        free(cell);
        e->cand[level] = cand; e->ncand[level] = nc;
        double t1 = now_s();
        e->t[1] += t1 - t0;
        int N = e->nPerLevel[level];
        int cap = nc + 8;
        int *idx = (int *)malloc(sizeof(int) * cap);
        int nk = orc_distribute_octtree(cand, nc, minBorderX, maxBorderX, minBorderY, maxBorderY, N, idx, cap);
        if (nk < 0) { free(idx); return nk; }
        orc_keypoint *kps = (orc_keypoint *)malloc(sizeof(orc_keypoint) * (nk > 0 ? nk : 1));
        const int scaledPatchSize = (int)(PATCH_SIZE * e->sf[level]);
        for (int i = 0; i < nk; ++i) {
            kps[i] = cand[idx[i]];
            kps[i].x += minBorderX;
            kps[i].y += minBorderY;
            kps[i].octave = level;
            kps[i].size = (float)scaledPatchSize;
        }
        free(idx);
        e->kps[level] = kps; e->nkps[level] = nk;
        e->t[2] += now_s() - t1;
    }
    double t0 = now_s();
    for (int level = 0; level < e->nlevels; ++level)
        for (int i = 0; i < e->nkps[level]; ++i) {
            orc_keypoint *kp = &e->kps[level][i];
            kp->angle = orc_ic_angle(e->pyr[level], e->pw[level], orc_cv_round_f(kp->x), orc_cv_round_f(kp->y));
        }
    e->t[3] += now_s() - t0;
    return 0;
}

/* operator()  src/ORBextractor.cc:1961-2084 */
int orc_extract(orc_extractor *e, const uint8_t *img, int w, int h, int stride, orc_keypoint *kps,
                uint8_t *desc, int cap) {
    if (!img || w <= 0 || h <= 0) return -1; /* _image.empty() -> silent return */
    free_frame_state(e);
    double t0 = now_s();
    compute_pyramid(e, img, w, h, stride);
    e->t[0] += now_s() - t0;
    int rc = compute_keypoints(e);
    if (rc < 0) return rc;
    int total = 0;
    for (int l = 0; l < e->nlevels; ++l) total += e->nkps[l];
    if (kps && total > cap) return -4;
    int offset = 0;
    for (int level = 0; level < e->nlevels; ++level) {
        int n = e->nkps[level];
        if (n == 0) continue;
        int W = e->pw[level], H = e->ph[level];
        t0 = now_s();
        e->blur[level] = (uint8_t *)malloc((size_t)W * H);
        orc_gaussian_blur7(e->pyr[level], W, H, W, e->blur[level], W);
        double t1 = now_s();
        e->t[4] += t1 - t0;
        for (int i = 0; i < n; ++i) {
            orc_keypoint *kp = &e->kps[level][i];
            uint8_t d[32];
            orc_descriptor(e->blur[level], W, orc_cv_round_f(kp->x), orc_cv_round_f(kp->y), kp->angle,
                           e->fp_mode, d);
            if (desc) memcpy(desc + (size_t)(offset + i) * 32, d, 32);
        }
        e->t[5] += now_s() - t1;
        if (kps) {
            for (int i = 0; i < n; ++i) {
                orc_keypoint kp = e->kps[level][i];
                if (level != 0) {
                    float scale = e->sf[level];
                    kp.x *= scale; kp.y *= scale;
                }
                kps[offset + i] = kp;
            }
        }
        offset += n;
    }
    return total;
}

int orc_level_dims(const orc_extractor *e, int level, int *w, int *h) {
    if (level < 0 || level >= e->nlevels || !e->pyr[level]) return -1;
    *w = e->pw[level]; *h = e->ph[level];
    return 0;
}
const uint8_t *orc_level_image(const orc_extractor *e, int level) { return e->pyr[level]; }
const uint8_t *orc_level_blur(const orc_extractor *e, int level) { return e->blur[level]; }
int orc_level_candidates(const orc_extractor *e, int level, orc_keypoint *out, int cap) {
    int n = e->ncand[level];
    if (out) memcpy(out, e->cand[level], sizeof(orc_keypoint) * (n < cap ? n : cap));
    return n;
}
int orc_level_keypoints(const orc_extractor *e, int level, orc_keypoint *out, int cap) {
    int n = e->nkps[level];
    if (out) memcpy(out, e->kps[level], sizeof(orc_keypoint) * (n < cap ? n : cap));
    return n;
}
void orc_get_stage_times(const orc_extractor *e, double *t6) { memcpy(t6, e->t, sizeof(double) * 6); }

/* ---------------------------------------------------------------- matcher primitives */
/* ORBmatcher::DescriptorDistance  src/ORBmatcher.cc:2073-2093 (SWAR popcount over 8 x 32 bit) */
int orc_descriptor_distance(const uint8_t *a, const uint8_t *b) {
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4); memcpy(&pb, b + 4 * i, 4);
        unsigned int v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

/* best / second-best bookkeeping exactly as every ORBmatcher search loop does it
 * (e.g. src/ORBmatcher.cc:627-640): strict '<' so the first minimum wins. */
void orc_match_bruteforce(const uint8_t *q, int nq, const uint8_t *t, int nt, int *best_idx,
                          int *best_dist, int *second_dist) {
    for (int i = 0; i < nq; ++i) {
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx = -1;
        for (int j = 0; j < nt; ++j) {
            int dist = orc_descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx = j; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        best_idx[i] = bestIdx; best_dist[i] = bestDist; second_dist[i] = bestDist2;
    }
}

/* ORBmatcher::ComputeThreeMaxima  src/ORBmatcher.cc:2026-2068 */
void orc_three_maxima(const int *hs, int L, int *ind1, int *ind2, int *ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    *ind1 = *ind2 = *ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = hs[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > max3) { max3 = s; *ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

/* cv::cvtColor(.., CV_RGB2GRAY / CV_BGR2GRAY / CV_RGBA2GRAY / CV_BGRA2GRAY), 8-bit (OpenCV 3.2 RGB2Gray<uchar>:
 * coefficients R2Y = 4899, G2Y = 9617, B2Y = 1868, yuv_shift = 14, rounding term 1 << 13); the conversion
 * Tracking::GrabImage* applies before the extractor (src/Tracking.cc:245-271, 302-320, 372-385). */
void orc_cvt_gray(const uint8_t *src, int w, int h, int sstride, int nch, int rgb_order, uint8_t *dst, int dstride) {
    const int ro = rgb_order ? 0 : 2, bo = rgb_order ? 2 : 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const uint8_t *p = src + (size_t)y * sstride + (size_t)x * nch;
            dst[(size_t)y * dstride + x] = (uint8_t)((p[ro] * 4899 + p[1] * 9617 + p[bo] * 1868 + (1 << 13)) >> 14);
        }
}
