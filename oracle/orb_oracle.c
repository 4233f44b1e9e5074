/*
 * orb_oracle.c -- scalar CPU restatement of the ORB-SLAM2 front-end feature
 * path.  TEST INFRASTRUCTURE ONLY (see orb_oracle.h; PARITY UNPINNED).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see Makefile).
 * All reference citations are relative to /root/reference/src/ORB_SLAM2/.
 */
#include "orb_oracle.h"

#include <float.h>
#include <stddef.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* scalar helpers                                                            */
/* ------------------------------------------------------------------------ */

/* cvRound: round-half-to-even (OpenCV uses cvtss2si / lrint). */
int orc_cvround_f(float v) { return (int)lrintf(v); }
static int cvround_d(double v) { return (int)lrint(v); }

/* cv::fastAtan2, OpenCV 3.x generic path: degree result in [0,360), 7th-order
 * odd polynomial, plain float mul/add (called at src/ORBextractor.cc:129). */
float orc_fast_atan2(float y, float x)
{
    const float rad2deg = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * rad2deg;
    const float p3 = -0.3258083974640975f * rad2deg;
    const float p5 = 0.1555786518463281f * rad2deg;
    const float p7 = -0.04432655554792128f * rad2deg;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
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

/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1710-1727): 8 words of
 * xor + SWAR popcount. */
int orc_hamming256(const uint8_t *a, const uint8_t *b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t wa, wb;
        memcpy(&wa, a + 4 * i, 4);
        memcpy(&wb, b + 4 * i, 4);
        uint32_t v = wa ^ wb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0x0F0F0F0Fu) * 0x01010101u) >> 24);
    }
    return dist;
}

/* ------------------------------------------------------------------------ */
/* extractor tables (src/ORBextractor.cc:456-577)                            */
/* ------------------------------------------------------------------------ */
int orc_tables(const orc_params *p, float *scale, float *inv_scale, float *sigma2,
               float *inv_sigma2, int *quota, int *umax16)
{
    int L = p->nlevels;
    if (L < 1 || L > ORC_MAX_LEVELS) return -1;
    float sc[ORC_MAX_LEVELS], s2[ORC_MAX_LEVELS];
    sc[0] = 1.0f;
    s2[0] = 1.0f;
    for (int i = 1; i < L; i++) { /* :476-482 */
        sc[i] = sc[i - 1] * p->scaleFactor;
        s2[i] = sc[i] * sc[i];
    }
    for (int i = 0; i < L; i++) { /* :487-491 */
        if (scale) scale[i] = sc[i];
        if (sigma2) sigma2[i] = s2[i];
        if (inv_scale) inv_scale[i] = 1.0f / sc[i];
        if (inv_sigma2) inv_sigma2[i] = 1.0f / s2[i];
    }
    if (quota) { /* :498-512 */
        float factor = 1.0f / p->scaleFactor;
        float nDesired = (float)p->nfeatures * (1 - factor) /
                         (1 - (float)pow((double)factor, (double)L));
        int sum = 0;
        for (int l = 0; l < L - 1; l++) {
            quota[l] = orc_cvround_f(nDesired);
            sum += quota[l];
            nDesired *= factor;
        }
        quota[L - 1] = p->nfeatures - sum > 0 ? p->nfeatures - sum : 0;
    }
    if (umax16) { /* :528-576: quarter circle of radius 15, made symmetric */
        const int R = 15;
        int vmax = (int)floorf((float)R * sqrtf(2.f) / 2 + 1);
        int vmin = (int)ceilf((float)R * sqrtf(2.f) / 2);
        for (int v = 0; v <= vmax; v++) umax16[v] = cvround_d(sqrt((double)(R * R) - v * v));
        for (int v = R, v0 = 0; v >= vmin; --v) {
            while (umax16[v0] == umax16[v0 + 1]) ++v0;
            umax16[v] = v0;
            ++v0;
        }
    }
    return 0;
}

/* Level size (src/ORBextractor.cc:1534-1536). */
void orc_level_size(const orc_params *p, int w, int h, int level, int *lw, int *lh)
{
    float sc = 1.0f;
    for (int i = 1; i <= level; i++) sc = sc * p->scaleFactor;
    float inv = 1.0f / sc;
    *lw = orc_cvround_f((float)w * inv);
    *lh = orc_cvround_f((float)h * inv);
}

/* ------------------------------------------------------------------------ */
/* cv::resize INTER_LINEAR on 8UC1 (called at src/ORBextractor.cc:1550).     */
/* OpenCV 3.2 fixed-point path: 11-bit coefficient tables, horizontal pass   */
/* into int32, vertical pass with the (>>4, >>16, +2, >>2) cast.             */
/* ------------------------------------------------------------------------ */
static short sat_short(float v)
{
    int i = orc_cvround_f(v);
    return (short)(i < -32768 ? -32768 : i > 32767 ? 32767 : i);
}

int orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst,
                         int dw, int dh, int dstride)
{
    if (sw < 1 || sh < 1 || dw < 1 || dh < 1) return -1;
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * (size_t)dw);
    int *row0 = (int *)malloc(sizeof(int) * (size_t)dw);
    int *row1 = (int *)malloc(sizeof(int) * (size_t)dw);
    if (!xofs || !ialpha || !row0 || !row1) {
        free(xofs); free(ialpha); free(row0); free(row1);
        return -2;
    }
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_short((1.f - fx) * 2048);
        ialpha[2 * dx + 1] = sat_short(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        short b0 = sat_short((1.f - fy) * 2048), b1 = sat_short(fy * 2048);
        int sy0 = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
        int sy1 = sy + 1 < 0 ? 0 : (sy + 1 >= sh ? sh - 1 : sy + 1);
        const uint8_t *S0 = src + (size_t)sy0 * sstride, *S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            int sx1 = sx + 1 < sw ? sx + 1 : sw - 1; /* weight is 0 there */
            int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
            row0[dx] = S0[sx] * a0 + S0[sx1] * a1;
            row1[dx] = S1[sx] * a0 + S1[sx1] * a1;
        }
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++)
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(row0); free(row1);
    return 0;
}

/* cv::copyMakeBorder(..., BORDER_REFLECT_101) in place around an interior that
 * already sits inside a larger allocation (src/ORBextractor.cc:1567-1586). */
static int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

void orc_border_reflect101(uint8_t *interior, int w, int h, int stride, int border)
{
    for (int y = 0; y < h; y++) {
        uint8_t *row = interior + (ptrdiff_t)y * stride;
        for (int i = 1; i <= border; i++) {
            row[-i] = row[reflect101(-i, w)];
            row[w - 1 + i] = row[reflect101(w - 1 + i, w)];
        }
    }
    for (int i = 1; i <= border; i++) {
        memcpy(interior + (ptrdiff_t)(-i) * stride - border,
               interior + (ptrdiff_t)reflect101(-i, h) * stride - border, (size_t)(w + 2 * border));
        memcpy(interior + (ptrdiff_t)(h - 1 + i) * stride - border,
               interior + (ptrdiff_t)reflect101(h - 1 + i, h) * stride - border,
               (size_t)(w + 2 * border));
    }
}

/* cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on 8U (called at
 * src/ORBextractor.cc:1502): separable, 8-bit fixed-point kernel, 32-bit
 * accumulation, result (sum + 2^15) >> 16 saturated. */
void orc_gauss7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride,
                int variant)
{
    static const int KA[7] = {18, 34, 49, 55, 49, 34, 18};
    static const int KB[7] = {18, 34, 48, 56, 48, 34, 18};
    const int *K = variant ? KB : KA;
    int *tmp = (int *)malloc(sizeof(int) * (size_t)w * (size_t)h);
    if (!tmp) return;
    for (int y = 0; y < h; y++) {
        const uint8_t *S = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int k = -3; k <= 3; k++) s += K[k + 3] * S[reflect101(x + k, w)];
            tmp[(size_t)y * w + x] = s;
        }
    }
    for (int y = 0; y < h; y++) {
        uint8_t *D = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int k = -3; k <= 3; k++) s += K[k + 3] * tmp[(size_t)reflect101(y + k, h) * w + x];
            s = (s + 32768) >> 16;
            D[x] = (uint8_t)(s > 255 ? 255 : s);
        }
    }
    free(tmp);
}

/* ------------------------------------------------------------------------ */
/* cv::FAST TYPE_9_16 with non-max suppression (called at                     */
/* src/ORBextractor.cc:1078-1088).                                            */
/* ------------------------------------------------------------------------ */
static const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* Is the pixel a corner at threshold t: 9 contiguous ring pixels all darker
 * than v-t or all brighter than v+t. */
static int fast_is_corner(const uint8_t *p, int stride, int t)
{
    int v = p[0];
    int ring[25];
    /* Quick reject: any 9-long arc contains one pixel of every opposite pair
     * (k, k+8), so each pair must have a member outside [v-t, v+t] of the same
     * polarity.  Pure pruning; the run test below decides. */
    {
        const int lo = v - t, hi = v + t;
#define ORC_CLS(k) (p[RING_DY[k] * stride + RING_DX[k]] < lo ? 1 : (p[RING_DY[k] * stride + RING_DX[k]] > hi ? 2 : 0))
        int d = ORC_CLS(0) | ORC_CLS(8);
        if (!d) return 0;
        d &= ORC_CLS(4) | ORC_CLS(12);
        if (!d) return 0;
        d &= ORC_CLS(2) | ORC_CLS(10);
        d &= ORC_CLS(6) | ORC_CLS(14);
        if (!d) return 0;
#undef ORC_CLS
    }
    for (int k = 0; k < 16; k++) ring[k] = p[RING_DY[k] * stride + RING_DX[k]];
    for (int k = 16; k < 25; k++) ring[k] = ring[k - 16];
    int run_d = 0, run_b = 0;
    for (int k = 0; k < 25; k++) {
        if (ring[k] < v - t) { if (++run_d > 8) return 1; } else run_d = 0;
        if (ring[k] > v + t) { if (++run_b > 8) return 1; } else run_b = 0;
    }
    return 0;
}

/* cornerScore<16>: max(t, best arc) - 1, where best arc = max over the 16
 * 9-long arcs of the minimum signed contrast (either polarity). */
static int fast_arc_strength(const uint8_t *p, int stride)
{
    int v = p[0];
    int d[25];
    for (int k = 0; k < 16; k++) d[k] = v - p[RING_DY[k] * stride + RING_DX[k]];
    for (int k = 16; k < 25; k++) d[k] = d[k - 16];
    int best = -255;
    for (int s = 0; s < 16; s++) {
        int mn = 255, mx = -255;
        for (int k = s; k < s + 9; k++) {
            if (d[k] < mn) mn = d[k];
            if (d[k] > mx) mx = d[k];
        }
        if (mn > best) best = mn;   /* all darker by at least mn */
        if (-mx > best) best = -mx; /* all brighter by at least -mx */
    }
    return best;
}

int orc_fast_score_pixel(const uint8_t *p, int stride)
{
    int a = fast_arc_strength(p, stride) - 1;
    return a < 0 ? 0 : a;
}

int orc_fast9(const uint8_t *img, int w, int h, int stride, int threshold, int nonmax,
              int *xs, int *ys, int *scores, int cap)
{
    int t = threshold < 0 ? 0 : threshold > 255 ? 255 : threshold;
    if (w < 7 || h < 7) return 0;
    uint8_t *sc = (uint8_t *)calloc((size_t)w * (size_t)h, 1);
    uint8_t *is = (uint8_t *)calloc((size_t)w * (size_t)h, 1);
    if (!sc || !is) { free(sc); free(is); return -2; }
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const uint8_t *p = img + (size_t)y * stride + x;
            if (fast_is_corner(p, stride, t)) {
                int a = fast_arc_strength(p, stride);
                int s = (a > t ? a : t) - 1;
                is[(size_t)y * w + x] = 1;
                sc[(size_t)y * w + x] = (uint8_t)s;
            }
        }
    int n = 0;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            size_t o = (size_t)y * w + x;
            if (!is[o]) continue;
            int s = sc[o];
            int keep = 1;
            if (nonmax) {
                for (int dy = -1; dy <= 1 && keep; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        if (!dx && !dy) continue;
                        if (!(s > sc[o + (ptrdiff_t)dy * w + dx])) { keep = 0; break; }
                    }
            }
            if (keep) {
                if (n < cap) { xs[n] = x; ys[n] = y; scores[n] = s; }
                n++;
            }
        }
    free(sc); free(is);
    return n;
}

/* Per-level candidates (src/ORBextractor.cc:1018-1107). */
int orc_level_candidates(const uint8_t *plane, int w, int h, int stride, int iniTh,
                         int minTh, float *xs, float *ys, float *resp, int cap)
{
    const float W = 30;
    const int minBX = ORC_EDGE - 3, minBY = minBX;
    const int maxBX = w - ORC_EDGE + 3, maxBY = h - ORC_EDGE + 3;
    const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols < 1 || nRows < 1) return -1;
    const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
    int tcap = (wCell + 6) * (hCell + 6);
    int *tx = (int *)malloc(sizeof(int) * 3 * (size_t)tcap);
    if (!tx) return -2;
    int *ty = tx + tcap, *ts = ty + tcap;
    int n = 0;
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBY - 3) continue;
        if (maxY > maxBY) maxY = (float)maxBY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBX - 6) continue;
            if (maxX > maxBX) maxX = (float)maxBX;
            int x0 = (int)iniX, y0 = (int)iniY, cw = (int)maxX - x0, ch = (int)maxY - y0;
            const uint8_t *sub = plane + (ptrdiff_t)y0 * stride + x0;
            int c = orc_fast9(sub, cw, ch, stride, iniTh, 1, tx, ty, ts, tcap);
            if (c == 0) c = orc_fast9(sub, cw, ch, stride, minTh, 1, tx, ty, ts, tcap);
            if (c < 0) { free(tx); return c; }
            for (int k = 0; k < c; k++) {
                if (n < cap) {
                    xs[n] = (float)tx[k] + (float)(j * wCell);
                    ys[n] = (float)ty[k] + (float)(i * hCell);
                    resp[n] = (float)ts[k];
                }
                n++;
            }
        }
    }
    free(tx);
    return n;
}

/* ------------------------------------------------------------------------ */
/* DistributeOctTree (src/ORBextractor.cc:606-674, 688-1006)                 */
/* ------------------------------------------------------------------------ */
typedef struct {
    int ulx, uly, urx, bry; /* UL.x, UL.y, UR.x, BR.y (enough to rebuild all corners) */
    int *pts;               /* indices of member points, stable order */
    int npts;
    int nomore;
    int seq;        /* creation sequence (tie-break for the size sort) */
    int prev, next; /* list links (pool indices) */
    int alive;
} qnode;

typedef struct {
    qnode *pool;
    int used, cap;
    int head, tail, size;
    int seq;
} qlist;

static int ql_alloc(qlist *L)
{
    if (L->used == L->cap) {
        int nc = L->cap * 2;
        qnode *np = (qnode *)realloc(L->pool, sizeof(qnode) * (size_t)nc);
        if (!np) return -1;
        L->pool = np;
        L->cap = nc;
    }
    qnode *q = &L->pool[L->used];
    memset(q, 0, sizeof(*q));
    q->prev = q->next = -1;
    q->seq = L->seq++;
    q->alive = 1;
    return L->used++;
}
static void ql_push_front(qlist *L, int id)
{
    qnode *q = &L->pool[id];
    q->prev = -1;
    q->next = L->head;
    if (L->head >= 0) L->pool[L->head].prev = id;
    L->head = id;
    if (L->tail < 0) L->tail = id;
    L->size++;
}
static void ql_push_back(qlist *L, int id)
{
    qnode *q = &L->pool[id];
    q->next = -1;
    q->prev = L->tail;
    if (L->tail >= 0) L->pool[L->tail].next = id;
    L->tail = id;
    if (L->head < 0) L->head = id;
    L->size++;
}
static int ql_erase(qlist *L, int id) /* returns next */
{
    qnode *q = &L->pool[id];
    int nx = q->next;
    if (q->prev >= 0) L->pool[q->prev].next = q->next; else L->head = q->next;
    if (q->next >= 0) L->pool[q->next].prev = q->prev; else L->tail = q->prev;
    q->alive = 0;
    free(q->pts);
    q->pts = NULL;
    L->size--;
    return nx;
}

/* ExtractorNode::DivideNode (:606-674).  Creates up to four children (only the
 * non-empty ones are pushed to the list front, in order n1..n4, as the callers
 * at :817-862 and :916-952 do).  Appends the children holding >1 point to
 * expandable[] and returns their number, or <0. */
static int q_divide(qlist *L, int id, const float *xs, const float *ys, int *expandable,
                    int *nexp)
{
    qnode P = L->pool[id];
    const int halfX = (int)ceilf((float)(P.urx - P.ulx) / 2);
    const int halfY = (int)ceilf((float)(P.bry - P.uly) / 2);
    const int midx = P.ulx + halfX, midy = P.uly + halfY;
    int cnt[4] = {0, 0, 0, 0};
    int *quad = (int *)malloc(sizeof(int) * (size_t)(P.npts > 0 ? P.npts : 1));
    if (!quad) return -2;
    for (int i = 0; i < P.npts; i++) {
        float x = xs[P.pts[i]], y = ys[P.pts[i]];
        int qd;
        if (x < (float)midx) qd = (y < (float)midy) ? 0 : 2;
        else qd = (y < (float)midy) ? 1 : 3;
        quad[i] = qd;
        cnt[qd]++;
    }
    for (int c = 0; c < 4; c++) {
        if (cnt[c] == 0) continue;
        int cid = ql_alloc(L);
        if (cid < 0) { free(quad); return -2; }
        qnode *C = &L->pool[cid];
        P = L->pool[id]; /* pool may have moved */
        C->ulx = (c & 1) ? midx : P.ulx;
        C->urx = (c & 1) ? P.urx : midx;
        C->uly = (c & 2) ? midy : P.uly;
        C->bry = (c & 2) ? P.bry : midy;
        C->pts = (int *)malloc(sizeof(int) * (size_t)cnt[c]);
        if (!C->pts) { free(quad); return -2; }
        C->npts = 0;
        for (int i = 0; i < P.npts; i++)
            if (quad[i] == c) C->pts[C->npts++] = P.pts[i];
        C->nomore = (C->npts == 1);
        ql_push_front(L, cid);
        if (C->npts > 1) expandable[(*nexp)++] = cid;
    }
    free(quad);
    return 0;
}

typedef struct { int count, seq, id; } size_entry;
/* Order among nodes of equal size in the partial passes.  The reference sorts pair<int, ExtractorNode*>
 * (src/ORBextractor.cc:906), i.e. by heap address, which no restatement can know.  Rule 0 (the one the GPU path
 * implements): the node created last is split first.  Rule 1: the node created first is split first -- only used to
 * measure how much of the output depends on the order (orc_set_tie_rule). */
static int g_tie_rule = 0;
void orc_set_tie_rule(int rule) { g_tie_rule = rule; }
static int size_entry_cmp(const void *a, const void *b)
{
    const size_entry *x = (const size_entry *)a, *y = (const size_entry *)b;
    if (x->count != y->count) return x->count < y->count ? -1 : 1;
    const int c = x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0);
    return g_tie_rule ? -c : c;
}

int orc_distribute_octree(const float *xs, const float *ys, const float *resp, int n,
                          int minX, int maxX, int minY, int maxY, int N, int *out_idx,
                          int cap, int *tie_straddle)
{
    if (tie_straddle) *tie_straddle = 0;
    if (n <= 0) return 0;
    /* :696-699 */
    const int nIni = (int)roundf((float)(maxX - minX) / (float)(maxY - minY));
    if (nIni < 1) return -1; /* the reference divides by zero here */
    const float hX = (float)(maxX - minX) / (float)nIni;

    qlist L;
    L.cap = 4 * (N > n ? n : N) + 4 * nIni + 64;
    L.pool = (qnode *)malloc(sizeof(qnode) * (size_t)L.cap);
    L.used = 0; L.head = L.tail = -1; L.size = 0; L.seq = 0;
    int *expandable = (int *)malloc(sizeof(int) * (size_t)(4 * n + 16));
    int *prev_exp = (int *)malloc(sizeof(int) * (size_t)(4 * n + 16));
    size_entry *order = (size_entry *)malloc(sizeof(size_entry) * (size_t)(4 * n + 16));
    int rc = 0;
    if (!L.pool || !expandable || !prev_exp || !order) { rc = -2; goto done; }

    /* roots :712-730 */
    for (int i = 0; i < nIni; i++) {
        int id = ql_alloc(&L);
        if (id < 0) { rc = -2; goto done; }
        qnode *q = &L.pool[id];
        q->ulx = (int)(hX * (float)i);
        q->urx = (int)(hX * (float)(i + 1));
        q->uly = 0;
        q->bry = maxY - minY;
        q->pts = (int *)malloc(sizeof(int) * (size_t)n);
        if (!q->pts) { rc = -2; goto done; }
        ql_push_back(&L, id);
    }
    /* :734-741 (root pool indices are 0..nIni-1) */
    for (int i = 0; i < n; i++) {
        int r = (int)(xs[i] / hX);
        if (r < 0 || r >= nIni) { rc = -3; goto done; }
        qnode *q = &L.pool[r];
        q->pts[q->npts++] = i;
    }
    /* :745-763 */
    for (int id = L.head; id >= 0;) {
        qnode *q = &L.pool[id];
        if (q->npts == 1) { q->nomore = 1; id = q->next; }
        else if (q->npts == 0) id = ql_erase(&L, id);
        else id = q->next;
    }

    int finish = 0;
    while (!finish) { /* :778-970 */
        int prevSize = L.size;
        int nexp = 0;
        for (int id = L.head; id >= 0;) {
            if (L.pool[id].nomore) { id = L.pool[id].next; continue; }
            int r = q_divide(&L, id, xs, ys, expandable, &nexp);
            if (r < 0) { rc = r; goto done; }
            id = ql_erase(&L, id);
        }
        int nToExpand = nexp;
        if (L.size >= N || L.size == prevSize) {
            finish = 1;
        } else if (L.size + nToExpand * 3 > N) {
            while (!finish) { /* :893-968 */
                prevSize = L.size;
                int nprev = nexp;
                memcpy(prev_exp, expandable, sizeof(int) * (size_t)nprev);
                nexp = 0;
                for (int k = 0; k < nprev; k++) {
                    order[k].count = L.pool[prev_exp[k]].npts;
                    order[k].seq = L.pool[prev_exp[k]].seq;
                    order[k].id = prev_exp[k];
                }
                qsort(order, (size_t)nprev, sizeof(size_entry), size_entry_cmp);
                for (int j = nprev - 1; j >= 0; j--) {
                    int r = q_divide(&L, order[j].id, xs, ys, expandable, &nexp);
                    if (r < 0) { rc = r; goto done; }
                    ql_erase(&L, order[j].id);
                    if (L.size >= N) {
                        /* the reference's pointer tie-break would matter only if an
                         * unprocessed entry has the same count as the last processed */
                        if (tie_straddle && j > 0 && order[j - 1].count == order[j].count)
                            *tie_straddle = 1;
                        break;
                    }
                }
                if (L.size >= N || L.size == prevSize) finish = 1;
            }
        }
    }

    /* :975-1003: best response per node, list order front -> back */
    {
        int m = 0;
        for (int id = L.head; id >= 0; id = L.pool[id].next) {
            qnode *q = &L.pool[id];
            int best = q->pts[0];
            float mr = resp[best];
            for (int k = 1; k < q->npts; k++)
                if (resp[q->pts[k]] > mr) { best = q->pts[k]; mr = resp[best]; }
            if (m < cap) out_idx[m] = best;
            m++;
        }
        rc = m;
    }
done:
    if (L.pool) {
        for (int i = 0; i < L.used; i++) free(L.pool[i].pts);
        free(L.pool);
    }
    free(expandable); free(prev_exp); free(order);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* IC_Angle (src/ORBextractor.cc:85-130)                                     */
/* ------------------------------------------------------------------------ */
float orc_ic_angle(const uint8_t *plane, int stride, int cx, int cy, const int *umax16)
{
    const uint8_t *c = plane + (ptrdiff_t)cy * stride + cx;
    int m01 = 0, m10 = 0;
    for (int u = -15; u <= 15; u++) m10 += u * c[u];
    for (int v = 1; v <= 15; v++) {
        int vsum = 0, d = umax16[v];
        for (int u = -d; u <= d; u++) {
            int vp = c[u + v * stride], vm = c[u - v * stride];
            vsum += vp - vm;
            m10 += u * (vp + vm);
        }
        m01 += v * vsum;
    }
    return orc_fast_atan2((float)m01, (float)m10);
}

/* ------------------------------------------------------------------------ */
/* computeOrbDescriptor (src/ORBextractor.cc:142-191)                        */
/* ------------------------------------------------------------------------ */
static const int8_t BRIEF_PATTERN[1024] = {
#include "brief_pattern.inc"
};

void orc_brief(const uint8_t *blurred, int stride, int cx, int cy, float angle_deg,
               uint8_t *desc32)
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = angle_deg * factorPI;
    float a = cosf(angle), b = sinf(angle);
    const uint8_t *c = blurred + (ptrdiff_t)cy * stride + cx;
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            const int8_t *p = BRIEF_PATTERN + (i * 8 + k) * 4;
            float x0 = (float)p[0], y0 = (float)p[1], x1 = (float)p[2], y1 = (float)p[3];
            int t0 = c[orc_cvround_f(x0 * b + y0 * a) * stride + orc_cvround_f(x0 * a - y0 * b)];
            int t1 = c[orc_cvround_f(x1 * b + y1 * a) * stride + orc_cvround_f(x1 * a - y1 * b)];
            val |= (t0 < t1) << k;
        }
        desc32[i] = (uint8_t)val;
    }
}

/* ------------------------------------------------------------------------ */
/* The extractor                                                              */
/* ------------------------------------------------------------------------ */
struct orc_extractor {
    orc_params p;
    float scale[ORC_MAX_LEVELS], inv_scale[ORC_MAX_LEVELS];
    int quota[ORC_MAX_LEVELS];
    int umax[16];
    uint8_t *plane[ORC_MAX_LEVELS]; /* allocation incl. halo */
    uint8_t *blur[ORC_MAX_LEVELS];
    int lw[ORC_MAX_LEVELS], lh[ORC_MAX_LEVELS];
    int count[ORC_MAX_LEVELS];
    int tie_straddles;
};

orc_extractor *orc_extractor_create(const orc_params *p)
{
    orc_extractor *e = (orc_extractor *)calloc(1, sizeof(*e));
    if (!e) return NULL;
    e->p = *p;
    if (orc_tables(p, e->scale, e->inv_scale, NULL, NULL, e->quota, e->umax) != 0) {
        free(e);
        return NULL;
    }
    return e;
}

static void extractor_free_planes(orc_extractor *e)
{
    for (int l = 0; l < ORC_MAX_LEVELS; l++) {
        free(e->plane[l]); e->plane[l] = NULL;
        free(e->blur[l]); e->blur[l] = NULL;
        e->count[l] = 0;
    }
}

void orc_extractor_destroy(orc_extractor *e)
{
    if (!e) return;
    extractor_free_planes(e);
    free(e);
}

static uint8_t *interior(orc_extractor *e, int l)
{
    int stride = e->lw[l] + 2 * ORC_EDGE;
    return e->plane[l] + (size_t)ORC_EDGE * stride + ORC_EDGE;
}

int orc_extractor_level(orc_extractor *e, int level, int blurred, const uint8_t **ptr, int *w,
                        int *h, int *stride)
{
    if (level < 0 || level >= e->p.nlevels || !e->plane[level]) return -1;
    *w = e->lw[level];
    *h = e->lh[level];
    if (blurred) {
        if (!e->blur[level]) return -2;
        *ptr = e->blur[level];
        *stride = e->lw[level];
    } else {
        *ptr = interior(e, level);
        *stride = e->lw[level] + 2 * ORC_EDGE;
    }
    return 0;
}
int orc_extractor_level_count(orc_extractor *e, int level) { return e->count[level]; }
int orc_extractor_tie_straddles(orc_extractor *e) { return e->tie_straddles; }

int orc_extract(orc_extractor *e, const uint8_t *img, int w, int h, int stride,
                orc_keypoint *kps, uint8_t *desc, int cap)
{
    const int L = e->p.nlevels;
    if (!img || w <= 0 || h <= 0) return 0; /* :1459-1460 */
    extractor_free_planes(e);
    e->tie_straddles = 0;

    /* ComputePyramid :1528-1592 */
    for (int l = 0; l < L; l++) {
        float sc = e->inv_scale[l];
        e->lw[l] = orc_cvround_f((float)w * sc);
        e->lh[l] = orc_cvround_f((float)h * sc);
        if (e->lw[l] < 1 || e->lh[l] < 1) return -1;
        int st = e->lw[l] + 2 * ORC_EDGE;
        e->plane[l] = (uint8_t *)malloc((size_t)st * (size_t)(e->lh[l] + 2 * ORC_EDGE));
        if (!e->plane[l]) return -2;
        uint8_t *in = interior(e, l);
        if (l == 0) {
            for (int y = 0; y < h; y++) memcpy(in + (size_t)y * st, img + (size_t)y * stride, (size_t)w);
        } else {
            int pst = e->lw[l - 1] + 2 * ORC_EDGE;
            int r = orc_resize_linear_u8(interior(e, l - 1), e->lw[l - 1], e->lh[l - 1], pst, in,
                                         e->lw[l], e->lh[l], st);
            if (r) return r;
        }
        orc_border_reflect101(in, e->lw[l], e->lh[l], st, ORC_EDGE);
    }

    /* ComputeKeyPointsOctTree :1009-1142 */
    int total = 0;
    orc_keypoint *lvl_kps[ORC_MAX_LEVELS];
    memset(lvl_kps, 0, sizeof(lvl_kps));
    int rc = 0;
    for (int l = 0; l < L && rc >= 0; l++) {
        int st = e->lw[l] + 2 * ORC_EDGE;
        const uint8_t *in = interior(e, l);
        int ccap = (e->lw[l] * e->lh[l]) / 4 + 1024;
        float *cx = (float *)malloc(sizeof(float) * 3 * (size_t)ccap);
        /* a subdivision round at most quadruples the node list, and it starts from up to 128 roots: the result of
         * DistributeOctTree has fewer than 4 * max(N, roots) nodes */
        const int scap = 4 * (e->quota[l] + 128) + 64;
        int *sel = (int *)malloc(sizeof(int) * (size_t)scap);
        if (!cx || !sel) { free(cx); free(sel); rc = -2; break; }
        float *cy = cx + ccap, *cr = cy + ccap;
        int nc = orc_level_candidates(in, e->lw[l], e->lh[l], st, e->p.iniThFAST, e->p.minThFAST,
                                      cx, cy, cr, ccap);
        if (nc < 0 || nc > ccap) { free(cx); free(sel); rc = nc < 0 ? nc : -4; break; }
        int ts = 0;
        int ns = orc_distribute_octree(cx, cy, cr, nc, ORC_EDGE - 3, e->lw[l] - ORC_EDGE + 3,
                                       ORC_EDGE - 3, e->lh[l] - ORC_EDGE + 3, e->quota[l], sel,
                                       scap, &ts);
        if (ns < 0 || ns > scap) { free(cx); free(sel); rc = ns < 0 ? ns : -4; break; }
        e->tie_straddles += ts;
        lvl_kps[l] = (orc_keypoint *)malloc(sizeof(orc_keypoint) * (size_t)(ns > 0 ? ns : 1));
        const int scaledPatch = (int)(31 * e->scale[l]); /* :1120 */
        for (int i = 0; i < ns; i++) { /* :1126-1135 */
            orc_keypoint *k = &lvl_kps[l][i];
            k->x = cx[sel[i]] + (float)(ORC_EDGE - 3);
            k->y = cy[sel[i]] + (float)(ORC_EDGE - 3);
            k->size = (float)scaledPatch;
            k->response = cr[sel[i]];
            k->octave = l;
            k->class_id = -1;
            /* computeOrientation :586-595 */
            k->angle = orc_ic_angle(in, st, orc_cvround_f(k->x), orc_cvround_f(k->y), e->umax);
        }
        e->count[l] = ns;
        total += ns;
        free(cx); free(sel);
    }
    if (rc < 0) {
        for (int l = 0; l < L; l++) free(lvl_kps[l]);
        return rc;
    }

    /* descriptors :1491-1520 */
    int off = 0;
    for (int l = 0; l < L; l++) {
        int ns = e->count[l];
        if (ns == 0) { free(lvl_kps[l]); continue; }
        int st = e->lw[l] + 2 * ORC_EDGE;
        e->blur[l] = (uint8_t *)malloc((size_t)e->lw[l] * (size_t)e->lh[l]);
        if (!e->blur[l]) { rc = -2; free(lvl_kps[l]); continue; }
        orc_gauss7(interior(e, l), e->lw[l], e->lh[l], st, e->blur[l], e->lw[l], e->p.gauss_variant);
        for (int i = 0; i < ns; i++) {
            orc_keypoint k = lvl_kps[l][i];
            if (off + i < cap) {
                orc_brief(e->blur[l], e->lw[l], orc_cvround_f(k.x), orc_cvround_f(k.y), k.angle,
                          desc + (size_t)(off + i) * 32);
                if (l != 0) { k.x *= e->scale[l]; k.y *= e->scale[l]; }
                kps[off + i] = k;
            }
        }
        off += ns;
        free(lvl_kps[l]);
    }
    if (rc < 0) return rc;
    return total;
}

/* ------------------------------------------------------------------------ */
/* Matching                                                                   */
/* ------------------------------------------------------------------------ */

/* Brute-force scan pattern (src/ORBmatcher.cc:205-226): candidates in index
 * order, strict '<' so the lowest index wins ties; best and second-best
 * distance. */
void orc_bruteforce(const uint8_t *Q, int nq, const uint8_t *T, int nt, int32_t *best_idx,
                    int32_t *best_d, int32_t *second_d)
{
    for (int i = 0; i < nq; i++) {
        int b1 = 256, b2 = 256, bi = -1;
        for (int j = 0; j < nt; j++) {
            int d = orc_hamming256(Q + (size_t)i * 32, T + (size_t)j * 32);
            if (d < b1) { b2 = b1; b1 = d; bi = j; }
            else if (d < b2) b2 = d;
        }
        best_idx[i] = bi; best_d[i] = b1; second_d[i] = b2;
    }
}

/* Grid geometry: geom6 = {build origin x, y, query origin x, y, inverse cell width, inverse cell height}.
 * A Frame bins its keypoints and computes the cell range of a query with one origin (mnMinX, mnMinY) and
 * mfGridElementWidthInv = 64 / (mnMaxX - mnMinX) (src/Frame.cc:101-102).  A KeyFrame keeps the Frame's grid
 * and inverse cell sizes but queries with its own int-truncated mnMinX / mnMinY (include/KeyFrame.h:185,
 * src/KeyFrame.cc:37-52, 585-624). */
void orc_geom_from_bounds(const float *b, float *geom6)
{
    geom6[0] = geom6[2] = b[0];
    geom6[1] = geom6[3] = b[1];
    geom6[4] = (float)ORC_GRID_COLS / (b[2] - b[0]);
    geom6[5] = (float)ORC_GRID_ROWS / (b[3] - b[1]);
}

/* Frame::PosInGrid + AssignFeaturesToGrid (src/Frame.cc:270-293, 485-500). */
static int pos_in_grid(const orc_keypoint *kp, const float *g, int *px, int *py)
{
    *px = (int)roundf((kp->x - g[0]) * g[4]);
    *py = (int)roundf((kp->y - g[1]) * g[5]);
    if (*px < 0 || *px >= ORC_GRID_COLS || *py < 0 || *py >= ORC_GRID_ROWS) return 0;
    return 1;
}

void orc_grid_build_geom(const orc_keypoint *kps, int n, const float *geom6, int32_t *cell_start,
                         int32_t *cell_items)
{
    const int NC = ORC_GRID_COLS * ORC_GRID_ROWS;
    int32_t *cnt = (int32_t *)calloc((size_t)NC + 1, sizeof(int32_t));
    for (int i = 0; i < n; i++) {
        int px, py;
        if (pos_in_grid(&kps[i], geom6, &px, &py)) cnt[px * ORC_GRID_ROWS + py]++;
    }
    int32_t acc = 0;
    for (int c = 0; c < NC; c++) { cell_start[c] = acc; acc += cnt[c]; cnt[c] = 0; }
    cell_start[NC] = acc;
    for (int i = 0; i < n; i++) {
        int px, py;
        if (pos_in_grid(&kps[i], geom6, &px, &py)) {
            int c = px * ORC_GRID_ROWS + py;
            cell_items[cell_start[c] + cnt[c]++] = i;
        }
    }
    free(cnt);
}

void orc_grid_build(const orc_keypoint *kps, int n, const float *bounds4, int32_t *cell_start,
                    int32_t *cell_items)
{
    float g[6];
    orc_geom_from_bounds(bounds4, g);
    orc_grid_build_geom(kps, n, g, cell_start, cell_items);
}

/* Frame::GetFeaturesInArea (src/Frame.cc:397-474); KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:585-624) with
 * minLevel = maxLevel = -1. */
int orc_features_in_area_geom(const orc_keypoint *kps, const float *g, const int32_t *cell_start,
                              const int32_t *cell_items, float x, float y, float r, int minLevel,
                              int maxLevel, int32_t *out, int cap)
{
    const float invW = g[4], invH = g[5];
    int n = 0;
    int c0 = (int)floorf((x - g[2] - r) * invW);
    if (c0 < 0) c0 = 0;
    if (c0 >= ORC_GRID_COLS) return 0;
    int c1 = (int)ceilf((x - g[2] + r) * invW);
    if (c1 > ORC_GRID_COLS - 1) c1 = ORC_GRID_COLS - 1;
    if (c1 < 0) return 0;
    int r0 = (int)floorf((y - g[3] - r) * invH);
    if (r0 < 0) r0 = 0;
    if (r0 >= ORC_GRID_ROWS) return 0;
    int r1 = (int)ceilf((y - g[3] + r) * invH);
    if (r1 > ORC_GRID_ROWS - 1) r1 = ORC_GRID_ROWS - 1;
    if (r1 < 0) return 0;
    const int check = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = c0; ix <= c1; ix++)
        for (int iy = r0; iy <= r1; iy++) {
            int c = ix * ORC_GRID_ROWS + iy;
            for (int k = cell_start[c]; k < cell_start[c + 1]; k++) {
                const orc_keypoint *kp = &kps[cell_items[k]];
                if (check) {
                    if (kp->octave < minLevel) continue;
                    if (maxLevel >= 0 && kp->octave > maxLevel) continue;
                }
                float dx = kp->x - x, dy = kp->y - y;
                if (fabsf(dx) < r && fabsf(dy) < r) {
                    if (n < cap) out[n] = cell_items[k];
                    n++;
                }
            }
        }
    return n;
}

int orc_features_in_area(const orc_keypoint *kps, const float *b, const int32_t *cell_start,
                         const int32_t *cell_items, float x, float y, float r, int minLevel,
                         int maxLevel, int32_t *out, int cap)
{
    float g[6];
    orc_geom_from_bounds(b, g);
    return orc_features_in_area_geom(kps, g, cell_start, cell_items, x, y, r, minLevel, maxLevel, out, cap);
}

/* ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:1661-1704). */
void orc_three_maxima(const int *sizes, int L, int *ind1, int *ind2, int *ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    *ind1 = *ind2 = *ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            *ind3 = *ind2; *ind2 = *ind1; *ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            *ind3 = *ind2; *ind2 = i;
        } else if (s > max3) {
            max3 = s; *ind3 = i;
        }
    }
    if ((float)max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

static int rot_bin(float a1, float a2)
{
    const float factor = 1.0f / ORC_HISTO_LENGTH; /* the reference's 1/30 quirk */
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == ORC_HISTO_LENGTH) bin = 0;
    return bin;
}

typedef struct { int *v; int n, cap; } ivec;
static void ivec_push(ivec *a, int x)
{
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 64;
        a->v = (int *)realloc(a->v, sizeof(int) * (size_t)a->cap);
    }
    a->v[a->n++] = x;
}

/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:422-571). */
int orc_search_for_initialization(const orc_keypoint *kps1, const uint8_t *desc1, int n1,
                                  const orc_keypoint *kps2, const uint8_t *desc2, int n2,
                                  const float *bounds4, float *prev_xy, int32_t *matches12,
                                  int windowSize, float nnratio, int checkOri)
{
    int nmatches = 0;
    int32_t *cs = (int32_t *)malloc(sizeof(int32_t) * (ORC_GRID_COLS * ORC_GRID_ROWS + 1));
    int32_t *ci = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 > 0 ? n2 : 1));
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 > 0 ? n2 : 1));
    int *mdist = (int *)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
    int *m21 = (int *)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
    ivec hist[ORC_HISTO_LENGTH];
    memset(hist, 0, sizeof(hist));
    orc_grid_build(kps2, n2, bounds4, cs, ci);
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    for (int i = 0; i < n2; i++) { mdist[i] = INT_MAX; m21[i] = -1; }

    for (int i1 = 0; i1 < n1; i1++) {
        int level1 = kps1[i1].octave;
        if (level1 > 0) continue;
        int nc = orc_features_in_area(kps2, bounds4, cs, ci, prev_xy[2 * i1], prev_xy[2 * i1 + 1],
                                      (float)windowSize, level1, level1, cand, n2);
        if (nc == 0) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int k = 0; k < nc; k++) {
            int i2 = cand[k];
            int dist = orc_hamming256(desc1 + (size_t)i1 * 32, desc2 + (size_t)i2 * 32);
            if (mdist[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= ORC_TH_LOW) {
            if ((float)bestDist < (float)bestDist2 * nnratio) {
                if (m21[bestIdx2] >= 0) { matches12[m21[bestIdx2]] = -1; nmatches--; }
                matches12[i1] = bestIdx2;
                m21[bestIdx2] = i1;
                mdist[bestIdx2] = bestDist;
                nmatches++;
                if (checkOri) ivec_push(&hist[rot_bin(kps1[i1].angle, kps2[bestIdx2].angle)], i1);
            }
        }
    }
    if (checkOri) {
        int sizes[ORC_HISTO_LENGTH], i1_, i2_, i3_;
        for (int i = 0; i < ORC_HISTO_LENGTH; i++) sizes[i] = hist[i].n;
        orc_three_maxima(sizes, ORC_HISTO_LENGTH, &i1_, &i2_, &i3_);
        for (int i = 0; i < ORC_HISTO_LENGTH; i++) {
            if (i == i1_ || i == i2_ || i == i3_) continue;
            for (int j = 0; j < hist[i].n; j++) {
                int idx1 = hist[i].v[j];
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
            }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (matches12[i1] >= 0) {
            prev_xy[2 * i1] = kps2[matches12[i1]].x;
            prev_xy[2 * i1 + 1] = kps2[matches12[i1]].y;
        }
    for (int i = 0; i < ORC_HISTO_LENGTH; i++) free(hist[i].v);
    free(cs); free(ci); free(cand); free(mdist); free(m21);
    return nmatches;
}

/* The three Tracking-side SearchByProjection overloads
 * (src/ORBmatcher.cc:45-129, 1379-1521, 1523-1650) behind one entry point; the
 * projection geometry that produces (u, v, r, level window, ur) is the
 * caller's job (it is outside the descriptor hot path). */
int orc_search_by_projection(int mode, const orc_wquery *q, const uint8_t *qdesc, int nq,
                             const orc_keypoint *kps2, const uint8_t *desc2,
                             const float *uright2, int n2, const float *bounds4,
                             const uint8_t *occupied, int32_t *assign, int th_dist,
                             float nnratio, int checkOri)
{
    float g[6];
    orc_geom_from_bounds(bounds4, g);
    return orc_search_by_projection_geom(mode, q, qdesc, nq, kps2, desc2, uright2, n2, g, occupied, assign, th_dist,
                                         nnratio, checkOri);
}

int orc_search_by_projection_geom(int mode, const orc_wquery *q, const uint8_t *qdesc, int nq,
                                  const orc_keypoint *kps2, const uint8_t *desc2,
                                  const float *uright2, int n2, const float *geom6,
                                  const uint8_t *occupied, int32_t *assign, int th_dist,
                                  float nnratio, int checkOri)
{
    int nmatches = 0;
    int32_t *cs = (int32_t *)malloc(sizeof(int32_t) * (ORC_GRID_COLS * ORC_GRID_ROWS + 1));
    int32_t *ci = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 > 0 ? n2 : 1));
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 > 0 ? n2 : 1));
    uint8_t *occ = (uint8_t *)malloc((size_t)(n2 > 0 ? n2 : 1));
    ivec hist[ORC_HISTO_LENGTH];
    memset(hist, 0, sizeof(hist));
    orc_grid_build_geom(kps2, n2, geom6, cs, ci);
    for (int i = 0; i < n2; i++) { assign[i] = -1; occ[i] = occupied ? occupied[i] : 0; }

    for (int i = 0; i < nq; i++) {
        if (!(q[i].flags & 1)) continue;
        const float r = q[i].r;
        int nc = orc_features_in_area_geom(kps2, geom6, cs, ci, q[i].u, q[i].v, r, q[i].minLevel,
                                           q[i].maxLevel, cand, n2);
        if (nc == 0) continue;
        int bestDist = 256, bestDist2 = 256, bestLevel = -1, bestLevel2 = -1, bestIdx = -1;
        for (int k = 0; k < nc; k++) {
            int i2 = cand[k];
            if (occ[i2]) continue;
            if (mode != 2 && uright2 && uright2[i2] > 0) {
                float er = fabsf(q[i].ur - uright2[i2]);
                if (er > r) continue;
            }
            int dist = orc_hamming256(qdesc + (size_t)i * 32, desc2 + (size_t)i2 * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist;
                bestLevel2 = bestLevel; bestLevel = kps2[i2].octave;
                bestIdx = i2;
            } else if (mode == 0 && dist < bestDist2) {
                bestLevel2 = kps2[i2].octave;
                bestDist2 = dist;
            }
        }
        /* bestIdx < 0: every candidate was skipped; the reference never gets here with one (its
         * thresholds are below the initial 256), a caller-supplied th_dist >= 256 must not either */
        if (bestIdx >= 0 && bestDist <= th_dist) {
            if (mode == 0 && bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2)
                continue;
            assign[bestIdx] = i;
            occ[bestIdx] = (mode == 2) ? 1 : ((q[i].flags & 2) ? 1 : 0);
            nmatches++;
            if (mode != 0 && checkOri)
                ivec_push(&hist[rot_bin(q[i].angle, kps2[bestIdx].angle)], bestIdx);
        }
    }
    if (mode != 0 && checkOri) {
        int sizes[ORC_HISTO_LENGTH], i1_, i2_, i3_;
        for (int i = 0; i < ORC_HISTO_LENGTH; i++) sizes[i] = hist[i].n;
        orc_three_maxima(sizes, ORC_HISTO_LENGTH, &i1_, &i2_, &i3_);
        for (int i = 0; i < ORC_HISTO_LENGTH; i++) {
            if (i == i1_ || i == i2_ || i == i3_) continue;
            /* the reference sets the keypoint's map point to NULL here (:1513, :1642): -2 = "cleared", as
             * distinct from -1 = "never touched" (an entry that held a map point without observations keeps it) */
            for (int j = 0; j < hist[i].n; j++) { assign[hist[i].v[j]] = -2; nmatches--; }
        }
    }
    for (int i = 0; i < ORC_HISTO_LENGTH; i++) free(hist[i].v);
    free(cs); free(ci); free(cand); free(occ);
    return nmatches;
}

/* ORBmatcher::SearchByBoW (src/ORBmatcher.cc:159-288 and :573-706). */
int orc_search_by_bow(const orc_wquery *q, const uint8_t *qdesc, int nq, const orc_keypoint *kps2,
                      const uint8_t *desc2, int n2, const int32_t *seg_start,
                      const int32_t *seg_index, const uint8_t *excluded, int th_dist,
                      float nnratio, int checkOri, int32_t *assign, int32_t *qmatch)
{
    int nmatches = 0;
    uint8_t *taken = (uint8_t *)calloc((size_t)(n2 > 0 ? n2 : 1), 1); /* vpMapPointMatches[i] != NULL / vbMatched2 */
    ivec hist[ORC_HISTO_LENGTH];
    memset(hist, 0, sizeof(hist));
    for (int i = 0; i < n2; i++) assign[i] = -1;
    for (int i = 0; i < nq; i++) qmatch[i] = -1;
    for (int i = 0; i < nq; i++) {
        if (!(q[i].flags & 1)) continue; /* :193-197 no (good) map point */
        const int s = q[i].minLevel;
        int bestDist1 = 256, bestIdx = -1, bestDist2 = 256;
        for (int k = seg_start[s]; k < seg_start[s + 1]; k++) { /* :205-226 */
            const int t = seg_index[k];
            if (taken[t] || (excluded && excluded[t])) continue;
            const int dist = orc_hamming256(qdesc + (size_t)i * 32, desc2 + (size_t)t * 32);
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx = t; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestIdx >= 0 && bestDist1 <= th_dist && (float)bestDist1 < nnratio * (float)bestDist2) { /* :228-230 */
            assign[bestIdx] = i;
            qmatch[i] = bestIdx;
            taken[bestIdx] = 1;
            if (checkOri) ivec_push(&hist[rot_bin(q[i].angle, kps2[bestIdx].angle)], i);
            nmatches++;
        }
    }
    if (checkOri) { /* :267-285 */
        int sizes[ORC_HISTO_LENGTH], i1_, i2_, i3_;
        for (int i = 0; i < ORC_HISTO_LENGTH; i++) sizes[i] = hist[i].n;
        orc_three_maxima(sizes, ORC_HISTO_LENGTH, &i1_, &i2_, &i3_);
        for (int i = 0; i < ORC_HISTO_LENGTH; i++) {
            if (i == i1_ || i == i2_ || i == i3_) continue;
            for (int j = 0; j < hist[i].n; j++) {
                const int qi = hist[i].v[j];
                assign[qmatch[qi]] = -1;
                qmatch[qi] = -1;
                nmatches--;
            }
        }
    }
    for (int i = 0; i < ORC_HISTO_LENGTH; i++) free(hist[i].v);
    free(taken);
    return nmatches;
}

/* ORBmatcher::CheckDistEpipolarLine (src/ORBmatcher.cc:140-157). */
static int check_dist_epipolar_line(float x1, float y1, const orc_keypoint *kp2, const float *F12, const float *sigma2)
{
    const float a = x1 * F12[0] + y1 * F12[3] + F12[6];
    const float b = x1 * F12[1] + y1 * F12[4] + F12[7];
    const float c = x1 * F12[2] + y1 * F12[5] + F12[8];
    const float num = a * kp2->x + b * kp2->y + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2[kp2->octave];
}

/* ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:708-874). */
int orc_search_for_triangulation(const orc_wquery *q, const uint8_t *qdesc, int nq,
                                 const orc_keypoint *kps2, const uint8_t *desc2, const float *uright2,
                                 int n2, const int32_t *seg_start, const int32_t *seg_index,
                                 const uint8_t *excluded, const float *F12, float ex, float ey,
                                 const float *scale2, const float *sigma2, int onlyStereo,
                                 int checkOri, int32_t *qmatch)
{
    (void)n2;
    int nmatches = 0;
    ivec hist[ORC_HISTO_LENGTH];
    memset(hist, 0, sizeof(hist));
    for (int i = 0; i < nq; i++) qmatch[i] = -1;
    for (int i = 0; i < nq; i++) {
        if (!(q[i].flags & 1)) continue;
        const int bStereo1 = (q[i].flags & 4) != 0;
        const int s = q[i].minLevel;
        int bestDist = ORC_TH_LOW, bestIdx2 = -1;
        for (int k = seg_start[s]; k < seg_start[s + 1]; k++) {
            const int idx2 = seg_index[k];
            if (excluded && excluded[idx2]) continue; /* vbMatched2 is never set by the reference (:776) */
            const int bStereo2 = uright2 && uright2[idx2] >= 0;
            if (onlyStereo && !bStereo2) continue;
            const int dist = orc_hamming256(qdesc + (size_t)i * 32, desc2 + (size_t)idx2 * 32);
            if (dist > ORC_TH_LOW || dist > bestDist) continue;
            const orc_keypoint *kp2 = &kps2[idx2];
            if (!bStereo1 && !bStereo2) {
                const float distex = ex - kp2->x;
                const float distey = ey - kp2->y;
                if (distex * distex + distey * distey < 100 * scale2[kp2->octave]) continue;
            }
            if (check_dist_epipolar_line(q[i].u, q[i].v, kp2, F12, sigma2)) { bestIdx2 = idx2; bestDist = dist; }
        }
        if (bestIdx2 >= 0) {
            qmatch[i] = bestIdx2;
            nmatches++;
            if (checkOri) ivec_push(&hist[rot_bin(q[i].angle, kps2[bestIdx2].angle)], i);
        }
    }
    if (checkOri) {
        int sizes[ORC_HISTO_LENGTH], i1_, i2_, i3_;
        for (int i = 0; i < ORC_HISTO_LENGTH; i++) sizes[i] = hist[i].n;
        orc_three_maxima(sizes, ORC_HISTO_LENGTH, &i1_, &i2_, &i3_);
        for (int i = 0; i < ORC_HISTO_LENGTH; i++) {
            if (i == i1_ || i == i2_ || i == i3_) continue;
            for (int j = 0; j < hist[i].n; j++) { qmatch[hist[i].v[j]] = -1; nmatches--; }
        }
    }
    for (int i = 0; i < ORC_HISTO_LENGTH; i++) free(hist[i].v);
    return nmatches;
}

/* Candidate loops of ORBmatcher::Fuse (src/ORBmatcher.cc:929-979, 1097-1115) and
 * ORBmatcher::SearchBySim3 (:1240-1256, 1320-1336). */
int orc_search_in_area(const orc_wquery *q, const uint8_t *qdesc, int nq, const orc_keypoint *kps2,
                       const uint8_t *desc2, const float *uright2, int n2, const float *bounds4,
                       const float *inv_sigma2, int chi2_gate, int th_dist, int32_t *qmatch)
{
    float g[6];
    orc_geom_from_bounds(bounds4, g);
    return orc_search_in_area_geom(q, qdesc, nq, kps2, desc2, uright2, n2, g, inv_sigma2, chi2_gate, th_dist, qmatch);
}

int orc_search_in_area_geom(const orc_wquery *q, const uint8_t *qdesc, int nq, const orc_keypoint *kps2,
                            const uint8_t *desc2, const float *uright2, int n2, const float *geom6,
                            const float *inv_sigma2, int chi2_gate, int th_dist, int32_t *qmatch)
{
    int nmatches = 0;
    int32_t *cs = (int32_t *)malloc(sizeof(int32_t) * (ORC_GRID_COLS * ORC_GRID_ROWS + 1));
    int32_t *ci = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 > 0 ? n2 : 1));
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 > 0 ? n2 : 1));
    orc_grid_build_geom(kps2, n2, geom6, cs, ci);
    for (int i = 0; i < nq; i++) {
        qmatch[i] = -1;
        if (!(q[i].flags & 1)) continue;
        const float u = q[i].u, v = q[i].v, ur = q[i].ur;
        /* KeyFrame::GetFeaturesInArea has no level filter; the callers apply [predicted-1, predicted] right after */
        const int nc = orc_features_in_area_geom(kps2, geom6, cs, ci, u, v, q[i].r, -1, -1, cand, n2);
        int bestDist = 256, bestIdx = -1;
        for (int k = 0; k < nc; k++) {
            const int idx = cand[k];
            const orc_keypoint *kp = &kps2[idx];
            const int kpLevel = kp->octave;
            if (kpLevel < q[i].minLevel || kpLevel > q[i].maxLevel) continue;
            if (chi2_gate) {
                if (uright2 && uright2[idx] >= 0) {
                    const float ex = u - kp->x, ey = v - kp->y, er = ur - uright2[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_sigma2[kpLevel] > 7.8) continue;
                } else {
                    const float ex = u - kp->x, ey = v - kp->y;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_sigma2[kpLevel] > 5.99) continue;
                }
            }
            const int dist = orc_hamming256(qdesc + (size_t)i * 32, desc2 + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestIdx >= 0 && bestDist <= th_dist) { qmatch[i] = bestIdx; nmatches++; }
    }
    free(cs); free(ci); free(cand);
    return nmatches;
}

static int int_cmp(const void *a, const void *b)
{
    const int x = *(const int *)a, y = *(const int *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:297-334). */
void orc_distinctive_descriptors(const uint8_t *desc, const int32_t *start, int npoints,
                                 int32_t *best_idx, int32_t *best_median)
{
    for (int p = 0; p < npoints; p++) {
        const int N = start[p + 1] - start[p];
        const uint8_t *D = desc + (size_t)start[p] * 32;
        if (N <= 0) { best_idx[p] = -1; if (best_median) best_median[p] = -1; continue; }
        int *dist = (int *)malloc(sizeof(int) * (size_t)N * N), *row = (int *)malloc(sizeof(int) * (size_t)N);
        for (int i = 0; i < N; i++) {
            dist[i * N + i] = 0;
            for (int j = i + 1; j < N; j++) {
                const int d = orc_hamming256(D + (size_t)i * 32, D + (size_t)j * 32);
                dist[i * N + j] = d;
                dist[j * N + i] = d;
            }
        }
        int BestMedian = INT_MAX, BestIdx = 0;
        for (int i = 0; i < N; i++) {
            memcpy(row, dist + (size_t)i * N, sizeof(int) * (size_t)N);
            qsort(row, (size_t)N, sizeof(int), int_cmp);
            const int median = row[(size_t)(0.5 * (N - 1))];
            if (median < BestMedian) { BestMedian = median; BestIdx = i; }
        }
        best_idx[p] = BestIdx;
        if (best_median) best_median[p] = BestMedian;
        free(dist); free(row);
    }
}

/* Frame::ComputeStereoFromRGBD (src/Frame.cc:777-798). */
void orc_stereo_from_rgbd(const orc_keypoint *kps, const orc_keypoint *kps_un, int n, const float *depth,
                          int w, int h, int stride_floats, float bf, float *uright, float *out_depth)
{
    for (int i = 0; i < n; i++) {
        uright[i] = -1; out_depth[i] = -1;
        const int v = (int)kps[i].y, u = (int)kps[i].x; /* cv::Mat::at<float>(int, int) from float arguments */
        if (u < 0 || u >= w || v < 0 || v >= h) continue; /* the reference does not check; keypoints lie inside */
        const float d = depth[(size_t)v * stride_floats + u];
        if (d > 0) { out_depth[i] = d; uright[i] = kps_un[i].x - bf / d; }
    }
}

/* Frame::UndistortKeyPoints (src/Frame.cc:517-558); cv::undistortPoints [OpenCV-knowledge]. */
void orc_undistort_keypoints(const orc_keypoint *kps, int n, const float *K4, const float *D5, orc_keypoint *out)
{
    if (D5[0] == 0.0f) { /* :519-523 */
        for (int i = 0; i < n; i++) out[i] = kps[i];
        return;
    }
    const double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
    const double k[12] = {D5[0], D5[1], D5[2], D5[3], D5[4], 0, 0, 0, 0, 0, 0, 0};
    const double ifx = 1. / fx, ify = 1. / fy;
    const double RR[3][3] = {{fx, 0, cx}, {0, fy, cy}, {0, 0, 1}}; /* P * R with R = I, P = K */
    for (int i = 0; i < n; i++) {
        double x = kps[i].x, y = kps[i].y;
        const double x0 = x = (x - cx) * ifx;
        const double y0 = y = (y - cy) * ify;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        const double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
        const double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
        const double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
        out[i] = kps[i];
        out[i].x = (float)(xx * ww);
        out[i].y = (float)(yy * ww);
    }
}

/* ------------------------------------------------------------------------ */
/* DBoW2 vocabulary transform (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h)    */
/* ------------------------------------------------------------------------ */
static int int32_cmp(const void *a, const void *b)
{
    const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
typedef struct { int32_t node, feat; } node_feat;
static int node_feat_cmp(const void *a, const void *b)
{
    const node_feat *x = (const node_feat *)a, *y = (const node_feat *)b;
    if (x->node != y->node) return x->node < y->node ? -1 : 1;
    return x->feat < y->feat ? -1 : (x->feat > y->feat ? 1 : 0);
}

int orc_voc_transform(int L, int scoring, int weighting, int n_nodes, const int32_t *parent,
                      const uint8_t *is_leaf, const uint8_t *node_desc, const double *weight,
                      const uint8_t *feat_desc, int n, int levelsup, int32_t *word_id,
                      int32_t *node_id, int32_t *bow_ids, double *bow_vals, int32_t *n_bow,
                      int32_t *fv_nodes, int32_t *fv_start, int32_t *fv_feat, int32_t *n_fv)
{
    *n_bow = 0; *n_fv = 0; fv_start[0] = 0; /* v.clear(); fv.clear(); :1131-1132 */
    /* m_nodes[pid].children.push_back(nid) in file order (:1385-1392); words in file order of the leaves (:1408-1415) */
    int32_t *cstart = (int32_t *)calloc((size_t)n_nodes + 2, sizeof(int32_t));
    int32_t *child = (int32_t *)malloc(((size_t)n_nodes + 1) * sizeof(int32_t));
    int32_t *cur = (int32_t *)malloc(((size_t)n_nodes + 1) * sizeof(int32_t));
    int32_t *wid = (int32_t *)malloc(((size_t)n_nodes + 1) * sizeof(int32_t));
    int rc = 0, n_words = 0;
    for (int i = 0; i < n_nodes; i++) {
        if (parent[i] < 0 || parent[i] > i) { rc = -1; break; }
        cstart[parent[i] + 1]++;
    }
    if (!rc) {
        for (int j = 0; j <= n_nodes; j++) { cstart[j + 1] += cstart[j]; }
        for (int j = 0; j <= n_nodes; j++) cur[j] = cstart[j];
        for (int i = 0; i < n_nodes; i++) child[cur[parent[i]]++] = i + 1;
        wid[0] = -1;
        for (int i = 0; i < n_nodes; i++) {
            wid[i + 1] = is_leaf[i] ? n_words++ : -1;
            if ((cstart[i + 2] - cstart[i + 1] == 0) != (is_leaf[i] != 0)) rc = -1;
        }
    }
    if (rc || n_words == 0) { /* empty(): :1134 */
        free(cstart); free(child); free(cur); free(wid);
        return rc;
    }
    double *acc = (double *)calloc((size_t)n_words, sizeof(double));
    uint8_t *seen = (uint8_t *)calloc((size_t)n_words, 1);
    node_feat *nf = (node_feat *)malloc(((size_t)n + 1) * sizeof(node_feat));
    int nkept = 0, nb = 0;
    const int nid_level = L - levelsup; /* :1226 */
    const int tf = weighting == 0 || weighting == 1; /* TF_IDF, TF (BowVector.h:36-42) */
    for (int f = 0; f < n; f++) {
        const uint8_t *d = feat_desc + (size_t)f * 32;
        int final_id = 0, level = 0, nid = nid_level <= 0 ? 0 : -1; /* :1227-1230 */
        do { /* :1232-1254 */
            ++level;
            const int32_t *ch = child + cstart[final_id];
            const int nch = cstart[final_id + 1] - cstart[final_id];
            final_id = ch[0];
            int best_d = orc_hamming256(d, node_desc + (size_t)(final_id - 1) * 32);
            for (int c = 1; c < nch; c++) {
                const int dd = orc_hamming256(d, node_desc + (size_t)(ch[c] - 1) * 32);
                if (dd < best_d) { best_d = dd; final_id = ch[c]; }
            }
            if (level == nid_level) nid = final_id;
        } while (cstart[final_id + 1] - cstart[final_id] > 0);
        /* the reference leaves *nid unset when the leaf lies above nid_level (an uninitialised local of
         * :1151); this restatement reports the leaf itself there */
        if (nid < 0) nid = final_id;
        const int w_id = wid[final_id];
        const double w = weight[final_id - 1];
        word_id[f] = w_id; node_id[f] = nid;
        if (w > 0) { /* :1157, :1185 */
            if (!seen[w_id]) { seen[w_id] = 1; bow_ids[nb++] = w_id; acc[w_id] = w; } /* insert */
            else if (tf) acc[w_id] += w;                                              /* addWeight; addIfNotExist keeps the first */
            nf[nkept].node = nid; nf[nkept].feat = f; nkept++;
        }
    }
    qsort(bow_ids, (size_t)nb, sizeof(int32_t), int32_cmp); /* std::map iteration order */
    for (int j = 0; j < nb; j++) bow_vals[j] = acc[bow_ids[j]];
    /* mustNormalize (ScoringObject.h:74-89): every scoring but DOT_PRODUCT; L2 norm only for L2_NORM */
    const int must = scoring != 5, l2 = scoring == 1;
    if (tf && nb > 0 && !must) { /* :1164-1170 */
        const double nd = (double)nb;
        for (int j = 0; j < nb; j++) bow_vals[j] /= nd;
    }
    if (must) { /* BowVector::normalize, BowVector.cpp:62-84 */
        double norm = 0.0;
        if (!l2) { for (int j = 0; j < nb; j++) norm += fabs(bow_vals[j]); }
        else { for (int j = 0; j < nb; j++) norm += bow_vals[j] * bow_vals[j]; norm = sqrt(norm); }
        if (norm > 0.0) for (int j = 0; j < nb; j++) bow_vals[j] /= norm;
    }
    *n_bow = nb;
    qsort(nf, (size_t)nkept, sizeof(node_feat), node_feat_cmp);
    int nn = 0;
    for (int j = 0; j < nkept; j++) {
        if (j == 0 || nf[j].node != nf[j - 1].node) { fv_nodes[nn] = nf[j].node; fv_start[nn] = j; nn++; }
        fv_feat[j] = nf[j].feat;
    }
    fv_start[nn] = nkept;
    *n_fv = nn;
    free(acc); free(seen); free(nf); free(cstart); free(child); free(cur); free(wid);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Frame::ComputeStereoMatches (src/Frame.cc:600-774)                         */
/* ------------------------------------------------------------------------ */
typedef struct { int d, i; } dist_idx;
static int dist_idx_cmp(const void *a, const void *b)
{
    const dist_idx *x = (const dist_idx *)a, *y = (const dist_idx *)b;
    if (x->d != y->d) return x->d < y->d ? -1 : 1;
    return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);
}

int orc_stereo_matches(const orc_keypoint *kpsL, const uint8_t *descL, int nL,
                       const orc_keypoint *kpsR, const uint8_t *descR, int nR,
                       const uint8_t *const *pyrL, const uint8_t *const *pyrR,
                       const int *pyr_w, const int *pyr_h, const int *pyr_strideL,
                       const int *pyr_strideR, int nlevels, const float *scale,
                       const float *inv_scale, float bf, float b, float *uRight, float *depth)
{
    (void)nlevels; (void)pyr_h;
    for (int i = 0; i < nL; i++) { uRight[i] = -1.0f; depth[i] = -1.0f; }
    const int thOrbDist = (ORC_TH_HIGH + ORC_TH_LOW) / 2;
    const int nRows = pyr_h[0];
    ivec *rows = (ivec *)calloc((size_t)nRows, sizeof(ivec));
    dist_idx *vd = (dist_idx *)malloc(sizeof(dist_idx) * (size_t)(nL > 0 ? nL : 1));
    int nvd = 0;
    for (int iR = 0; iR < nR; iR++) { /* :617-627 */
        const float kpY = kpsR[iR].y;
        const float r = 2.0f * scale[kpsR[iR].octave];
        const int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) ivec_push(&rows[yi], iR); /* reference: unchecked */
    }
    const float minZ = b, minD = 0, maxD = bf / minZ;
    for (int iL = 0; iL < nL; iL++) {
        const orc_keypoint *kpL = &kpsL[iL];
        const int levelL = kpL->octave;
        const float vL = kpL->y, uL = kpL->x;
        int row = (int)vL;
        if (row < 0 || row >= nRows) continue;
        const ivec *cands = &rows[row];
        if (cands->n == 0) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = ORC_TH_HIGH;
        int bestIdxR = 0;
        for (int k = 0; k < cands->n; k++) {
            const int iR = cands->v[k];
            const orc_keypoint *kpR = &kpsR[iR];
            if (kpR->octave < levelL - 1 || kpR->octave > levelL + 1) continue;
            const float uR = kpR->x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_hamming256(descL + (size_t)iL * 32, descR + (size_t)iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) { /* :686-757 */
            const float uR0 = kpsR[bestIdxR].x;
            const float sf = inv_scale[levelL];
            const float scaleduL = roundf(kpL->x * sf);
            const float scaledvL = roundf(kpL->y * sf);
            const float scaleduR0 = roundf(uR0 * sf);
            const int w = 5, L = 5;
            const uint8_t *PL = pyrL[levelL], *PR = pyrR[levelL];
            const int sL = pyr_strideL[levelL], sR = pyr_strideR[levelL];
            const int y0 = (int)(scaledvL - w), xL0 = (int)(scaleduL - w);
            const int cL = PL[(ptrdiff_t)(y0 + w) * sL + xL0 + w];
            int bestSad = INT_MAX, bestinc = 0;
            float vDists[11];
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= (float)pyr_w[levelL]) continue;
            for (int inc = -L; inc <= L; inc++) {
                const int xR0 = (int)(scaleduR0 + inc - w);
                const int cR = PR[(ptrdiff_t)(y0 + w) * sR + xR0 + w];
                float acc = 0; /* exact: integers below 2^24 */
                for (int yy = 0; yy < 2 * w + 1; yy++)
                    for (int xx = 0; xx < 2 * w + 1; xx++) {
                        float a = (float)PL[(ptrdiff_t)(y0 + yy) * sL + xL0 + xx] - (float)cL;
                        float c = (float)PR[(ptrdiff_t)(y0 + yy) * sR + xR0 + xx] - (float)cR;
                        acc += fabsf(a - c);
                    }
                float dist = acc;
                if (dist < (float)bestSad) { bestSad = (int)dist; bestinc = inc; }
                vDists[L + inc] = dist;
            }
            if (bestinc == -L || bestinc == L) continue;
            const float d1 = vDists[L + bestinc - 1], d2 = vDists[L + bestinc], d3 = vDists[L + bestinc + 1];
            const float deltaR = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = scale[levelL] * ((float)scaleduR0 + (float)bestinc + deltaR);
            float disparity = uL - bestuR;
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = (float)0.01; bestuR = (float)((double)uL - 0.01); }
                depth[iL] = bf / disparity;
                uRight[iL] = bestuR;
                vd[nvd].d = bestSad; vd[nvd].i = iL; nvd++;
            }
        }
    }
    int valid = nvd;
    if (nvd > 0) { /* :760-773 (the reference indexes an empty vector when nvd == 0) */
        qsort(vd, (size_t)nvd, sizeof(dist_idx), dist_idx_cmp);
        const float median = (float)vd[nvd / 2].d;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nvd - 1; i >= 0; i--) {
            if ((float)vd[i].d < thDist) break;
            uRight[vd[i].i] = -1; depth[vd[i].i] = -1; valid--;
        }
    }
    for (int i = 0; i < nRows; i++) free(rows[i].v);
    free(rows); free(vd);
    return valid;
}
