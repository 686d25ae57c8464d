/*
 * aev_oracle.c -- TEST INFRASTRUCTURE, NOT THE PRODUCT.
 *
 * A single-threaded, plain-C CPU restatement of the reference's asynchronous
 * (event-driven) conv / max-pool forward path.  It exists only so that
 *   - tests/ can check the HIP path against it,
 *   - __graft_entry__.smoke() can check one small step against it,
 *   - bench.py's `cpu_baseline` leg can time it on the GPU box's host cores.
 * Nothing under async_ev_cnn_amd/ may import, link or call this file.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement
 * against golden vectors under tests/golden/ that were produced by importing
 * the reference itself (tests/golden/gen_golden.py, run in the build
 * container where /root/reference exists).
 *
 * Every function cites the reference file:line (relative to the reference
 * repository root) whose algorithm it restates.  The restatement keeps the
 * reference's *algorithm* (dense getters, dense leak sweep, first-touch
 * im2col, min_argmax tie-break, pending-recompute bitmap) but not its code:
 * the reference is NumPy/Cython, this is scalar C over caller-owned buffers.
 *
 * Arithmetic contract (shared with the HIP kernels, see DESIGN.md "Numerics"):
 *   - all state is float32 (the semantics of the reference's pinned
 *     NumPy 1.18.5, requirements.yml:5; under NumPy >= 2 the reference
 *     silently widens some state to float64, SURVEY.md 8c caveat 1);
 *   - delta_leak is computed in float64 and rounded to float32 once before
 *     it touches a float32 map (integration.py:60,65; conv2d.py:115);
 *   - leak extrapolation is  fm = fm - (act * dl)  with two roundings
 *     (no FMA contraction);
 *   - a convolution output is a float32 fmaf chain starting from +0.0f, the
 *     bias is added afterwards with its own rounding (conv2d.py:175-178).
 *     Order of the chain: input channels in groups of four (the last group
 *     may be smaller); per group the kernel taps dy*kw + dx ascending; per
 *     tap the channels of the group ascending:
 *         for g: for tap: for c in 4g .. min(4g+3, Cin-1): acc = fmaf(w, x, acc)
 *     The reference leaves the order to BLAS sgemm (conv2d.py:175); any order
 *     is within the 1e-5 tolerance.  This one is what gfx950's f32 MFMAs do
 *     bit for bit when the four (or two) k of one instruction are the
 *     channels of a group at one tap: v_mfma_f32_16x16x4_f32 and
 *     v_mfma_f32_32x32x2_f32 accumulate their k in ascending order, one fmaf
 *     each (tools/probes/mfma_order.hip).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_E_STRIDE (-1)
#define ORC_E_SHAPE (-2)
#define ORC_E_BOUNDS (-3)
#define ORC_E_ALLOC (-4)
#define ORC_E_ARG (-5)

enum { ORC_INTEGRATION = 0, ORC_CONV = 1, ORC_POOL = 2 };
enum { ORC_SURFACE = 0, ORC_LAYER_ACTFN = 1, ORC_CONV_ACTFN = 2, ORC_FEATUREMAP = 3,
       ORC_RAW_ACT = 4, ORC_IDX_MAX = 5, ORC_RECOMPUTE = 6 };

/* ------------------------------------------------------------------------- */
/* Op 1: im2col_event  (src/libs/cutils.pyx:29-134)                           */
/* ------------------------------------------------------------------------- */
/*
 * image  [C,H,W] float32, C-contiguous (already padded by the caller)
 * ey/ex  [N] int32 event coordinates in image space
 * out_cols: column-major (Fortran) matrix, leading dimension = rows
 *    chan_as_cols == 0: rows = C*kh*kw, column j = site j,
 *                       row = c*kh*kw + dy*kw + dx         (cutils.pyx:125)
 *    chan_as_cols != 0: rows = kh*kw, column = C*site + c   (cutils.pyx:122-123)
 * out_y/out_x [capacity outH*outW] int32, first-touch order (cutils.pyx:108-112)
 * Returns ORC_OK and *n_out = number of distinct sites.
 */
int orc_im2col_event(const float *image, int C, int H, int W, const int32_t *ey, const int32_t *ex,
                     int N, int kh, int kw, int stride, int chan_as_cols, float *out_cols,
                     int32_t *out_y, int32_t *out_x, int *n_out)
{
    /* validate before allocating (the reference leaks here) */
    if (!(stride == 1 || (stride == kh && stride == kw))) return ORC_E_STRIDE; /* cutils.pyx:78-89 */
    if (C <= 0 || H < kh || W < kw || kh <= 0 || kw <= 0) return ORC_E_SHAPE;
    const int oh = (H - kh) / stride + 1, ow = (W - kw) / stride + 1; /* cutils.pyx:51-52 */
    if (stride != 1 && (H % kh || W % kw)) return ORC_E_SHAPE;       /* reference reads OOB */
    for (int i = 0; i < N; ++i)
        if (ey[i] < 0 || ey[i] >= H || ex[i] < 0 || ex[i] >= W) return ORC_E_BOUNDS;
    uint8_t *seen = (uint8_t *)calloc((size_t)oh * ow, 1);
    if (!seen) return ORC_E_ALLOC;
    const int kk = kh * kw;
    int n = 0;
    for (int i = 0; i < N; ++i) {
        const int y = ey[i], x = ex[i];
        int y0, y1, x0, x1; /* inclusive range of top-left corners whose window holds (y,x) */
        if (stride == 1) {
            y0 = y - (kh - 1) < 0 ? 0 : y - (kh - 1);
            x0 = x - (kw - 1) < 0 ? 0 : x - (kw - 1);
            y1 = y > H - kh ? H - kh : y;
            x1 = x > W - kw ? W - kw : x;
        } else {
            y0 = y1 = (y / stride) * kh;
            x0 = x1 = (x / stride) * kw;
        }
        for (int ty = y0; ty <= y1; ty += stride)
            for (int tx = x0; tx <= x1; tx += stride) {
                const int oy = ty / stride, ox = tx / stride;
                if (seen[oy * ow + ox]) continue;
                seen[oy * ow + ox] = 1;
                out_y[n] = oy;
                out_x[n] = ox;
                for (int c = 0; c < C; ++c)
                    for (int dy = 0; dy < kh; ++dy)
                        for (int dx = 0; dx < kw; ++dx) {
                            const float v = image[((size_t)c * H + ty + dy) * W + tx + dx];
                            if (chan_as_cols)
                                out_cols[(size_t)(C * n + c) * kk + dy * kw + dx] = v;
                            else
                                out_cols[(size_t)n * (C * kk) + c * kk + dy * kw + dx] = v;
                        }
                ++n;
            }
    }
    free(seen);
    *n_out = n;
    return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* Op 2: min_argmax  (src/libs/cutils.pyx:139-179)                            */
/* ------------------------------------------------------------------------- */
/* mx/mn: column-major [R, M] float32.  argmax[M], not_argmin[M] int32. */
int orc_min_argmax(const float *mx, const float *mn, int R, int M, int32_t *argmax,
                   int32_t *not_argmin)
{
    if (R <= 0 || M < 0) return ORC_E_SHAPE;
    for (int m = 0; m < M; ++m) {
        const float *a = mx + (size_t)m * R, *b = mn + (size_t)m * R;
        int best = 0, slow = 0;
        for (int r = 0; r < R; ++r) {
            if (a[r] > a[best])
                best = r; /* cutils.pyx:164-165 */
            else if (r > 0 && a[r] == a[best] && b[r] < b[best])
                best = r; /* tie -> strictly smaller decay wins, cutils.pyx:166-168 */
            if (b[r] < b[slow]) slow = r; /* cutils.pyx:171-172 */
        }
        argmax[m] = best;
        not_argmin[m] = b[best] != b[slow]; /* value compare, cutils.pyx:177 */
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* Layer chain                                                                */
/* ------------------------------------------------------------------------- */
typedef struct {
    int kind;
    int C, H, W; /* output shape (layer.py out_shape) */
    /* conv */
    int kh, kw, cin, stride;
    int pt, pb, pl, pr;   /* conv2d.py:38-54 */
    float alpha;          /* rounded to f32 once: conv2d.py:87 promotes alpha into a f32 array */
    float *wrows;         /* [Cout][K], K = cin*kh*kw  (conv2d.py:26 layout) */
    float *bias;          /* [Cout] */
    float *fm, *act;      /* _featuremap, _conv_actfn  conv2d.py:61-63 */
    float *init_fm;       /* conv2d.py:59-60 */
    /* pool */
    int32_t *idx, *init_idx; /* _idx_max[0]  maxpool.py:33-35 */
    uint8_t *pending;        /* _recompute_coords maxpool.py:36 */
    /* integration */
    double leak;
    long long prev_ts;
    float *surf;
    /* last output events */
    int32_t *evy, *evx;
    int nev;
    /* stats */
    long long n_sites_total;
} orc_layer;

typedef struct orc_net {
    int n;
    int cap;
    orc_layer *L;
    /* scratch */
    float *ga, *gb; /* dense getter outputs of the previous layer */
    float *cols;
    float *out;
    size_t ga_cap, cols_cap, out_cap;
} orc_net;

static void layer_free(orc_layer *l)
{
    free(l->wrows); free(l->bias); free(l->fm); free(l->act); free(l->init_fm);
    free(l->idx); free(l->init_idx); free(l->pending); free(l->surf); free(l->evy); free(l->evx);
}

void orc_net_destroy(orc_net *n)
{
    if (!n) return;
    for (int i = 0; i < n->n; ++i) layer_free(&n->L[i]);
    free(n->L); free(n->ga); free(n->gb); free(n->cols); free(n->out); free(n);
}

static int ensure(float **p, size_t *cap, size_t need)
{
    if (*cap >= need) return 0;
    float *q = (float *)realloc(*p, need * sizeof(float));
    if (!q) return -1;
    *p = q; *cap = need;
    return 0;
}

static orc_layer *push_layer(orc_net *n)
{
    if (n->n == n->cap) {
        int nc = n->cap ? n->cap * 2 : 16;
        orc_layer *q = (orc_layer *)realloc(n->L, nc * sizeof(orc_layer));
        if (!q) return NULL;
        n->L = q; n->cap = nc;
    }
    orc_layer *l = &n->L[n->n++];
    memset(l, 0, sizeof(*l));
    return l;
}

/* IntegrationLayer.__init__  (src/layers/integration.py:12-28) */
int orc_net_create(int H, int W, double leak, orc_net **out)
{
    if (H <= 0 || W <= 0) return ORC_E_SHAPE;
    orc_net *n = (orc_net *)calloc(1, sizeof(orc_net));
    if (!n) return ORC_E_ALLOC;
    orc_layer *l = push_layer(n);
    if (!l) { free(n); return ORC_E_ALLOC; }
    l->kind = ORC_INTEGRATION; l->C = 1; l->H = H; l->W = W; l->leak = leak; l->prev_ts = 0;
    l->surf = (float *)calloc((size_t)H * W, sizeof(float));
    l->evy = (int32_t *)malloc((size_t)H * W * sizeof(int32_t));
    l->evx = (int32_t *)malloc((size_t)H * W * sizeof(int32_t));
    *out = n;
    return ORC_OK;
}

/* ---- dense getters (layer.py:77-80; conv2d.py:80-94; integration.py:30-43; maxpool.py:42-79) ---- */
static inline float slope_of(float s, float alpha) { return s > 0.0f ? 1.0f : alpha; } /* conv2d.py:85-88 */

/* which: ORC_SURFACE / ORC_LAYER_ACTFN / ORC_CONV_ACTFN / ORC_FEATUREMAP ; dst [C*H*W] */
static void dense_get(const orc_net *n, int li, int which, float *dst)
{
    const orc_layer *l = &n->L[li];
    const size_t sz = (size_t)l->C * l->H * l->W;
    if (l->kind == ORC_INTEGRATION) {
        for (size_t i = 0; i < sz; ++i) {
            const float s = l->surf[i], p = s > 0.0f ? 1.0f : 0.0f; /* integration.py:36,42 */
            dst[i] = which == ORC_SURFACE ? s : which == ORC_FEATUREMAP ? s * p : p;
        }
    } else if (l->kind == ORC_CONV) {
        for (size_t i = 0; i < sz; ++i) {
            const float s = l->fm[i], sl = slope_of(s, l->alpha);
            dst[i] = which == ORC_SURFACE ? s
                   : which == ORC_LAYER_ACTFN ? sl
                   : which == ORC_CONV_ACTFN ? l->act[i] * sl   /* conv2d.py:93 */
                   : s * sl;                                     /* layer.py:80 */
        }
    } else { /* pool: gather the previous layer's map through idx_max, maxpool.py:42-79 */
        const orc_layer *p = &n->L[li - 1];
        float *src = (float *)malloc((size_t)p->C * p->H * p->W * sizeof(float));
        if (which == ORC_FEATUREMAP) {
            /* surface()*layer_actfn(): both gathered through the same index */
            float *s2 = (float *)malloc((size_t)p->C * p->H * p->W * sizeof(float));
            dense_get(n, li - 1, ORC_SURFACE, src);
            dense_get(n, li - 1, ORC_LAYER_ACTFN, s2);
            for (size_t i = 0; i < (size_t)p->C * p->H * p->W; ++i) src[i] = src[i] * s2[i];
            free(s2);
        } else
            dense_get(n, li - 1, which, src);
        for (int c = 0; c < l->C; ++c)
            for (int y = 0; y < l->H; ++y)
                for (int x = 0; x < l->W; ++x) {
                    const int j = l->idx[((size_t)c * l->H + y) * l->W + x];
                    const int sy = y * l->stride + j / l->kw, sx = x * l->stride + j % l->kw;
                    dst[((size_t)c * l->H + y) * l->W + x] = src[((size_t)c * p->H + sy) * p->W + sx];
                }
        free(src);
    }
}

/* zero-pad [C,H,W] -> [C,H+pt+pb,W+pl+pr]  (conv2d.py:68-72) */
static void pad_map(const float *src, int C, int H, int W, int pt, int pb, int pl, int pr, float *dst)
{
    const int Hp = H + pt + pb, Wp = W + pl + pr;
    memset(dst, 0, (size_t)C * Hp * Wp * sizeof(float));
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < H; ++y)
            memcpy(dst + ((size_t)c * Hp + y + pt) * Wp + pl, src + ((size_t)c * H + y) * W, W * sizeof(float));
}

/* out[co][j] = fmaf-chain_k wrows[co][k]*cols[j*K+k]  (+bias)   (conv2d.py:172-179)
 * cols is column-major [K, n] as im2col_event returns it. */
static void gemm_cols(const float *wrows, const float *bias, int Cout, int K, int kk, const float *cols, int n, float *out)
{
    enum { BS = 256 };
    float acc[BS];
    float *colsT = (float *)malloc((size_t)K * BS * sizeof(float));
    /* the chain's order (arithmetic contract): korder[q] = row k = c*kk + tap of im2col's layout */
    int *korder = (int *)malloc((size_t)K * sizeof(int));
    {
        const int cin = K / kk;
        int q = 0;
        for (int g = 0; g < cin; g += 4)
            for (int tap = 0; tap < kk; ++tap)
                for (int c = g; c < cin && c < g + 4; ++c) korder[q++] = c * kk + tap;
    }
    for (int j0 = 0; j0 < n; j0 += BS) {
        const int nb = n - j0 < BS ? n - j0 : BS;
        for (int j = 0; j < nb; ++j)
            for (int k = 0; k < K; ++k) colsT[(size_t)k * BS + j] = cols[(size_t)(j0 + j) * K + k];
        for (int co = 0; co < Cout; ++co) {
            const float *w = wrows + (size_t)co * K;
            for (int j = 0; j < nb; ++j) acc[j] = 0.0f;
            for (int q = 0; q < K; ++q) {
                const int k = korder[q];
                const float wk = w[k];
                const float *cr = colsT + (size_t)k * BS;
                for (int j = 0; j < nb; ++j) acc[j] = fmaf(wk, cr[j], acc[j]);
            }
            if (bias)
                for (int j = 0; j < nb; ++j) out[(size_t)co * n + j0 + j] = acc[j] + bias[co];
            else
                for (int j = 0; j < nb; ++j) out[(size_t)co * n + j0 + j] = acc[j];
        }
    }
    free(colsT);
    free(korder);
}

/* Conv2DLayer.__init__  (src/layers/conv2d.py:15-66)
 * kernel_hwio: [kh,kw,cin,cout] float32 (TF layout, conv2d.py:19); padding: 0 VALID, 1 SAME. */
int orc_net_add_conv(orc_net *n, const float *kernel_hwio, const float *bias, int kh, int kw, int cin,
                     int cout, int stride, double alpha, int padding)
{
    const orc_layer *p = &n->L[n->n - 1];
    if (cin != p->C) return ORC_E_SHAPE;
    if (stride != 1) return ORC_E_STRIDE; /* only stride 1 is reachable in the reference */
    const int ih = p->H, iw = p->W;
    int pt = 0, pb = 0, pl = 0, pr = 0, oh, ow;
    if (padding == 0) { /* conv2d.py:34-37 */
        if (ih < kh || iw < kw) return ORC_E_SHAPE;
        oh = (ih - kh) / stride + 1; ow = (iw - kw) / stride + 1;
    } else if (padding == 1) { /* conv2d.py:38-54 */
        oh = (ih + stride - 1) / stride; ow = (iw + stride - 1) / stride;
        int ph = ih % stride == 0 ? kh - stride : kh - ih % stride; if (ph < 0) ph = 0;
        int pw = iw % stride == 0 ? kw - stride : kw - iw % stride; if (pw < 0) pw = 0;
        pt = ph / 2; pb = ph - pt; pl = pw / 2; pr = pw - pl;
        /* conv2d.py:69,75: padding is applied only when top or bottom > 0 */
        if (!(pt > 0 || pb > 0)) { if (pl > 0 || pr > 0) return ORC_E_SHAPE; }
    } else
        return ORC_E_ARG;
    const int pidx = n->n - 1; /* push_layer may move n->L */
    orc_layer *l = push_layer(n);
    if (!l) return ORC_E_ALLOC;
    p = &n->L[pidx];
    l->kind = ORC_CONV; l->C = cout; l->H = oh; l->W = ow; l->kh = kh; l->kw = kw; l->cin = cin;
    l->stride = stride; l->pt = pt; l->pb = pb; l->pl = pl; l->pr = pr; l->alpha = (float)alpha;
    const int K = cin * kh * kw;
    const size_t sz = (size_t)cout * oh * ow;
    l->wrows = (float *)malloc((size_t)cout * K * sizeof(float));
    l->bias = (float *)malloc(cout * sizeof(float));
    l->fm = (float *)malloc(sz * sizeof(float));
    l->act = (float *)calloc(sz, sizeof(float));
    l->init_fm = (float *)malloc(sz * sizeof(float));
    l->evy = (int32_t *)malloc((size_t)oh * ow * sizeof(int32_t));
    l->evx = (int32_t *)malloc((size_t)oh * ow * sizeof(int32_t));
    for (int co = 0; co < cout; ++co) { /* transpose([3,2,0,1]) conv2d.py:26 */
        l->bias[co] = bias ? bias[co] : 0.0f;
        for (int c = 0; c < cin; ++c)
            for (int dy = 0; dy < kh; ++dy)
                for (int dx = 0; dx < kw; ++dx)
                    l->wrows[(size_t)co * K + (c * kh + dy) * kw + dx] =
                        kernel_hwio[(((size_t)dy * kw + dx) * cin + c) * cout + co];
    }
    /* initial feature map = dense stride-1 conv of the previous layer's current feature map (conv2d.py:59-60) */
    const int Hp = ih + pt + pb, Wp = iw + pl + pr;
    float *fmap = (float *)malloc((size_t)cin * ih * iw * sizeof(float));
    float *padded = (float *)malloc((size_t)cin * Hp * Wp * sizeof(float));
    dense_get(n, pidx, ORC_FEATUREMAP, fmap);
    pad_map(fmap, cin, ih, iw, pt, pb, pl, pr, padded);
    float *cols = (float *)malloc((size_t)K * oh * ow * sizeof(float));
    for (int oy = 0; oy < oh; ++oy) /* dense im2col, functional.py:4-34 */
        for (int ox = 0; ox < ow; ++ox)
            for (int c = 0; c < cin; ++c)
                for (int dy = 0; dy < kh; ++dy)
                    for (int dx = 0; dx < kw; ++dx)
                        cols[(size_t)(oy * ow + ox) * K + (c * kh + dy) * kw + dx] =
                            padded[((size_t)c * Hp + oy + dy) * Wp + ox + dx];
    gemm_cols(l->wrows, l->bias, cout, K, kh * kw, cols, oh * ow, l->init_fm);
    memcpy(l->fm, l->init_fm, sz * sizeof(float));
    free(cols); free(padded); free(fmap);
    return n->n - 1;
}

/* MaxPoolLayer.__init__  (src/layers/maxpool.py:14-40) */
int orc_net_add_pool(orc_net *n, int kh, int kw, int stride)
{
    const orc_layer *p = &n->L[n->n - 1];
    if (!(stride == kh && stride == kw)) return ORC_E_STRIDE; /* cutils.pyx:83,89: pool path needs stride == k */
    if (p->H % kh || p->W % kw) return ORC_E_SHAPE;          /* SURVEY Q7 */
    const int oh = (p->H - kh) / stride + 1, ow = (p->W - kw) / stride + 1; /* maxpool.py:27-28 */
    const int pidx = n->n - 1;
    orc_layer *l = push_layer(n);
    if (!l) return ORC_E_ALLOC;
    p = &n->L[pidx];
    l->kind = ORC_POOL; l->C = p->C; l->H = oh; l->W = ow; l->kh = kh; l->kw = kw; l->stride = stride;
    const size_t sz = (size_t)l->C * oh * ow;
    l->idx = (int32_t *)malloc(sz * sizeof(int32_t));
    l->init_idx = (int32_t *)malloc(sz * sizeof(int32_t));
    l->pending = (uint8_t *)calloc((size_t)oh * ow, 1);
    l->evy = (int32_t *)malloc((size_t)oh * ow * sizeof(int32_t));
    l->evx = (int32_t *)malloc((size_t)oh * ow * sizeof(int32_t));
    float *surf = (float *)malloc((size_t)p->C * p->H * p->W * sizeof(float));
    dense_get(n, pidx, ORC_SURFACE, surf);
    for (int c = 0; c < l->C; ++c) /* plain first-occurrence argmax, maxpool.py:33 */
        for (int y = 0; y < oh; ++y)
            for (int x = 0; x < ow; ++x) {
                int best = 0;
                float bv = surf[((size_t)c * p->H + y * stride) * p->W + x * stride];
                for (int j = 1; j < kh * kw; ++j) {
                    const float v = surf[((size_t)c * p->H + y * stride + j / kw) * p->W + x * stride + j % kw];
                    if (v > bv) { bv = v; best = j; }
                }
                l->init_idx[((size_t)c * oh + y) * ow + x] = best;
            }
    memcpy(l->idx, l->init_idx, sz * sizeof(int32_t));
    free(surf);
    return n->n - 1;
}

/* reset()  (integration.py:48-51; conv2d.py:99-103; maxpool.py:84-90) */
int orc_net_reset(orc_net *n)
{
    for (int i = 0; i < n->n; ++i) {
        orc_layer *l = &n->L[i];
        const size_t sz = (size_t)l->C * l->H * l->W;
        if (l->kind == ORC_INTEGRATION) { memset(l->surf, 0, sz * sizeof(float)); l->prev_ts = 0; }
        else if (l->kind == ORC_CONV) { memcpy(l->fm, l->init_fm, sz * sizeof(float)); memset(l->act, 0, sz * sizeof(float)); }
        else { memcpy(l->idx, l->init_idx, sz * sizeof(int32_t)); memset(l->pending, 0, (size_t)l->H * l->W); }
        l->nev = 0;
    }
    return ORC_OK;
}

/* IntegrationLayer.compute  (src/layers/integration.py:53-91); yxt = [N][3] int32 (y,x,ts) */
static int step_integration(orc_layer *l, const int32_t *yxt, int N, double *delta_leak)
{
    if (N <= 0) return ORC_E_ARG;
    const int H = l->H, W = l->W;
    long long last = yxt[2];
    for (int i = 0; i < N; ++i) {
        if (yxt[3 * i] < 0 || yxt[3 * i] >= H || yxt[3 * i + 1] < 0 || yxt[3 * i + 1] >= W) return ORC_E_BOUNDS;
        if (yxt[3 * i + 2] > last) last = yxt[3 * i + 2];
    }
    const double dl = (double)(last - l->prev_ts) * l->leak; /* :60 */
    const float dlf = (float)dl;
    uint8_t *mask = (uint8_t *)calloc((size_t)H * W, 1);
    float *s = l->surf;
    /* bit0 before_positive (:63), bit1 after_leak_negative (:66) */
    for (int i = 0; i < H * W; ++i) {
        uint8_t m = s[i] > 0.0f ? 1 : 0;
        float v = s[i] - dlf; /* :65 */
        if (v <= 0.0f) { v = 0.0f; m |= 2; }
        s[i] = v;
        mask[i] = m;
    }
    /* :71  fancy-index += : every duplicate reads the pre-update value, the last writer wins.
     * Walk the events backwards and let only the first visit (= last event) of a pixel write. */
    uint8_t *done = (uint8_t *)calloc((size_t)H * W, 1);
    for (int i = N - 1; i >= 0; --i) {
        const int p = yxt[3 * i] * W + yxt[3 * i + 1];
        if (done[p]) continue;
        done[p] = 1;
        const double v = 1.0 - (double)(last - yxt[3 * i + 2]) * l->leak;
        s[p] = (float)((double)s[p] + v);
    }
    l->nev = 0;
    for (int i = 0; i < H * W; ++i) {
        int neg2 = 0;
        if (s[i] <= 0.0f) { s[i] = 0.0f; neg2 = 1; } /* :72-74 */
        const int ev = ((mask[i] & 1) && ((mask[i] & 2) || neg2)) || done[i]; /* :78-80 */
        if (ev) { l->evy[l->nev] = i / W; l->evx[l->nev] = i % W; ++l->nev; } /* :83 row-major */
    }
    l->prev_ts = last; /* :86 */
    free(mask); free(done);
    *delta_leak = dl;
    return ORC_OK;
}

/* Conv2DLayer.compute  (src/layers/conv2d.py:105-137) */
static int step_conv(orc_net *n, int li, const int32_t *ey, const int32_t *ex, int N, double delta_leak)
{
    orc_layer *l = &n->L[li];
    const orc_layer *p = &n->L[li - 1];
    const int ih = p->H, iw = p->W, cin = p->C;
    const int Hp = ih + l->pt + l->pb, Wp = iw + l->pl + l->pr;
    const int K = cin * l->kh * l->kw, HW = l->H * l->W;
    for (int i = 0; i < N; ++i)
        if (ey[i] < 0 || ey[i] >= ih || ex[i] < 0 || ex[i] >= iw) return ORC_E_BOUNDS;
    const size_t psz = (size_t)cin * ih * iw, ppsz = (size_t)cin * Hp * Wp;
    if (ensure(&n->ga, &n->ga_cap, psz + 2 * ppsz)) return ORC_E_ALLOC;
    if (ensure(&n->cols, &n->cols_cap, (size_t)K * HW)) return ORC_E_ALLOC;
    if (ensure(&n->out, &n->out_cap, (size_t)l->C * HW)) return ORC_E_ALLOC;
    float *tmp = n->ga, *pfm = n->ga + psz, *pact = pfm + ppsz;
    dense_get(n, li - 1, ORC_FEATUREMAP, tmp);  /* :108 */
    pad_map(tmp, cin, ih, iw, l->pt, l->pb, l->pl, l->pr, pfm);
    dense_get(n, li - 1, ORC_CONV_ACTFN, tmp);  /* :109 */
    pad_map(tmp, cin, ih, iw, l->pt, l->pb, l->pl, l->pr, pact);
    int32_t *py = (int32_t *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int32_t));
    int32_t *px = (int32_t *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int32_t));
    for (int i = 0; i < N; ++i) { py[i] = ey[i] + l->pt; px[i] = ex[i] + l->pl; } /* :110, pad_events :74-78 */

    uint8_t *chg = (uint8_t *)calloc((size_t)HW, 1);
    const float dlf = (float)delta_leak;
    const size_t sz = (size_t)l->C * HW;
    uint8_t *before = (uint8_t *)malloc(sz);
    for (size_t i = 0; i < sz; ++i) {
        before[i] = l->fm[i] >= 0.0f; /* :113 */
        const float d = l->act[i] * dlf;
        l->fm[i] = l->fm[i] - d;      /* :115, two roundings */
    }
    int32_t *oy = (int32_t *)malloc((size_t)HW * sizeof(int32_t));
    int32_t *ox = (int32_t *)malloc((size_t)HW * sizeof(int32_t));
    int ns = 0, rc;
    rc = orc_im2col_event(pfm, cin, Hp, Wp, py, px, N, l->kh, l->kw, 1, 0, n->cols, oy, ox, &ns); /* :118 -> :172 */
    if (rc == ORC_OK) {
        gemm_cols(l->wrows, l->bias, l->C, K, l->kh * l->kw, n->cols, ns, n->out);
        for (int c = 0; c < l->C; ++c)
            for (int j = 0; j < ns; ++j) l->fm[(size_t)c * HW + oy[j] * l->W + ox[j]] = n->out[(size_t)c * ns + j]; /* :120 */
        rc = orc_im2col_event(pact, cin, Hp, Wp, py, px, N, l->kh, l->kw, 1, 0, n->cols, oy, ox, &ns); /* :121 */
    }
    if (rc == ORC_OK) {
        gemm_cols(l->wrows, NULL, l->C, K, l->kh * l->kw, n->cols, ns, n->out);
        for (int c = 0; c < l->C; ++c)
            for (int j = 0; j < ns; ++j) l->act[(size_t)c * HW + oy[j] * l->W + ox[j]] = n->out[(size_t)c * ns + j]; /* :123 */
        for (int c = 0; c < l->C; ++c) /* :125-127 */
            for (int i = 0; i < HW; ++i)
                if ((l->fm[(size_t)c * HW + i] >= 0.0f) != before[(size_t)c * HW + i]) chg[i] = 1;
        for (int j = 0; j < ns; ++j) chg[oy[j] * l->W + ox[j]] = 1; /* :129 */
        l->nev = 0;
        for (int i = 0; i < HW; ++i)
            if (chg[i]) { l->evy[l->nev] = i / l->W; l->evx[l->nev] = i % l->W; ++l->nev; } /* :131 */
        l->n_sites_total += ns;
    }
    free(py); free(px); free(chg); free(before); free(oy); free(ox);
    return rc;
}

/* MaxPoolLayer.compute  (src/layers/maxpool.py:105-161) */
static int step_pool(orc_net *n, int li, const int32_t *ey, const int32_t *ex, int N)
{
    orc_layer *l = &n->L[li];
    const orc_layer *p = &n->L[li - 1];
    const int C = l->C, oh = l->H, ow = l->W, s = l->stride, kk = l->kh * l->kw;
    for (int i = 0; i < N; ++i)
        if (ey[i] < 0 || ey[i] >= p->H || ex[i] < 0 || ex[i] >= p->W) return ORC_E_BOUNDS;
    const size_t psz = (size_t)C * p->H * p->W;
    if (ensure(&n->ga, &n->ga_cap, 2 * psz)) return ORC_E_ALLOC;
    if (ensure(&n->cols, &n->cols_cap, 2 * (size_t)kk * C * oh * ow)) return ORC_E_ALLOC;
    float *surf = n->ga, *dec = n->ga + psz;
    dense_get(n, li - 1, ORC_SURFACE, surf);     /* :108 */
    dense_get(n, li - 1, ORC_CONV_ACTFN, dec);   /* :109 */
    for (int i = 0; i < N; ++i) l->pending[(ey[i] / s) * ow + ex[i] / s] = 0; /* :118-120 */
    int np = 0;
    for (int i = 0; i < oh * ow; ++i) np += l->pending[i];
    int32_t *ay = (int32_t *)malloc((size_t)(N + np + 1) * sizeof(int32_t));
    int32_t *ax = (int32_t *)malloc((size_t)(N + np + 1) * sizeof(int32_t));
    memcpy(ay, ey, (size_t)N * sizeof(int32_t));
    memcpy(ax, ex, (size_t)N * sizeof(int32_t));
    int m = N;
    for (int i = 0; i < oh * ow; ++i) /* :122-126, np.where order = row-major */
        if (l->pending[i]) { ay[m] = (i / ow) * s; ax[m] = (i % ow) * s; ++m; }
    float *c0 = n->cols, *c1 = n->cols + (size_t)kk * C * oh * ow;
    int ns = 0, rc;
    rc = orc_im2col_event(surf, C, p->H, p->W, ay, ax, m, l->kh, l->kw, s, 1, c0, l->evy, l->evx, &ns); /* :130 */
    if (rc == ORC_OK) rc = orc_im2col_event(dec, C, p->H, p->W, ay, ax, m, l->kh, l->kw, s, 1, c1, l->evy, l->evx, &ns); /* :134 */
    if (rc == ORC_OK) {
        int32_t *am = (int32_t *)malloc((size_t)(C * ns + 1) * sizeof(int32_t));
        int32_t *na = (int32_t *)malloc((size_t)(C * ns + 1) * sizeof(int32_t));
        orc_min_argmax(c0, c1, kk, C * ns, am, na); /* :138 */
        for (int j = 0; j < ns; ++j) {
            int any = 0;
            for (int c = 0; c < C; ++c) {
                any |= na[C * j + c];
                l->idx[((size_t)c * oh + l->evy[j]) * ow + l->evx[j]] = am[C * j + c]; /* :145-151 */
            }
            if (any) l->pending[l->evy[j] * ow + l->evx[j]] = 1; /* :140-142 (only ever sets) */
        }
        l->nev = ns; /* :154 */
        l->n_sites_total += ns;
        free(am); free(na);
    }
    free(ay); free(ax);
    return rc;
}

/* Layer.compute for one layer with caller-provided events.
 * layer 0: ev_a = yxt [N][3], ev_b ignored, *delta_leak is OUTPUT.
 * others : ev_a = y[N], ev_b = x[N], *delta_leak is INPUT. */
int orc_layer_step(orc_net *n, int li, const int32_t *ev_a, const int32_t *ev_b, int N, double *delta_leak)
{
    if (li < 0 || li >= n->n) return ORC_E_ARG;
    orc_layer *l = &n->L[li];
    if (l->kind == ORC_INTEGRATION) return step_integration(l, ev_a, N, delta_leak);
    if (l->kind == ORC_CONV) return step_conv(n, li, ev_a, ev_b, N, *delta_leak);
    return step_pool(n, li, ev_a, ev_b, N);
}

/* compute_all over the whole chain (event_numpy.py:75-79; conv2d.py:139-141) */
int orc_net_step(orc_net *n, const int32_t *yxt, int N)
{
    double dl = 0.0;
    int rc = step_integration(&n->L[0], yxt, N, &dl);
    for (int i = 1; i < n->n && rc == ORC_OK; ++i) {
        const orc_layer *p = &n->L[i - 1];
        rc = n->L[i].kind == ORC_CONV ? step_conv(n, i, p->evy, p->evx, p->nev, dl)
                                      : step_pool(n, i, p->evy, p->evx, p->nev);
    }
    return rc;
}

int orc_net_num_layers(const orc_net *n) { return n->n; }

int orc_layer_shape(const orc_net *n, int li, int *chw)
{
    if (li < 0 || li >= n->n) return ORC_E_ARG;
    chw[0] = n->L[li].C; chw[1] = n->L[li].H; chw[2] = n->L[li].W;
    return ORC_OK;
}

int orc_layer_num_events(const orc_net *n, int li) { return n->L[li].nev; }
long long orc_layer_sites_total(const orc_net *n, int li) { return n->L[li].n_sites_total; }

int orc_layer_events(const orc_net *n, int li, int32_t *y, int32_t *x)
{
    if (li < 0 || li >= n->n) return ORC_E_ARG;
    memcpy(y, n->L[li].evy, (size_t)n->L[li].nev * sizeof(int32_t));
    memcpy(x, n->L[li].evx, (size_t)n->L[li].nev * sizeof(int32_t));
    return ORC_OK;
}

/* Read one of the layer's maps into dst.  Float maps: [C,H,W] float32.
 * ORC_IDX_MAX: int32 [C*H*W]; ORC_RECOMPUTE: uint8 [H*W]; ORC_RAW_ACT: raw _conv_actfn (conv only). */
int orc_layer_read(const orc_net *n, int li, int which, void *dst)
{
    if (li < 0 || li >= n->n) return ORC_E_ARG;
    const orc_layer *l = &n->L[li];
    const size_t sz = (size_t)l->C * l->H * l->W;
    if (which <= ORC_FEATUREMAP) { dense_get(n, li, which, (float *)dst); return ORC_OK; }
    if (which == ORC_RAW_ACT && l->kind == ORC_CONV) { memcpy(dst, l->act, sz * sizeof(float)); return ORC_OK; }
    if (which == ORC_IDX_MAX && l->kind == ORC_POOL) { memcpy(dst, l->idx, sz * sizeof(int32_t)); return ORC_OK; }
    if (which == ORC_RECOMPUTE && l->kind == ORC_POOL) { memcpy(dst, l->pending, (size_t)l->H * l->W); return ORC_OK; }
    return ORC_E_ARG;
}

/* viz.integrate_frame  (src/libs/viz.py:7-24): stateless dense twin of the integration layer.
 * frame [H*W] float32 in/out, prev_ts in/out. */
int orc_integrate_frame(const int32_t *yxt, int N, double leak, int H, int W, float *frame, long long *prev_ts)
{
    if (N <= 0) return ORC_E_ARG;
    long long last = yxt[2];
    for (int i = 0; i < N; ++i) if (yxt[3 * i + 2] > last) last = yxt[3 * i + 2];
    const float dlf = (float)((double)(last - *prev_ts) * leak);
    for (int i = 0; i < H * W; ++i) { float v = frame[i] - dlf; frame[i] = v < 0.0f ? 0.0f : v; } /* :18-19 */
    uint8_t *done = (uint8_t *)calloc((size_t)H * W, 1);
    for (int i = N - 1; i >= 0; --i) { /* :21 last-writer-wins */
        const int p = yxt[3 * i] * W + yxt[3 * i + 1];
        if (done[p]) continue;
        done[p] = 1;
        frame[p] = (float)((double)frame[p] + (1.0 - (double)(last - yxt[3 * i + 2]) * leak));
    }
    for (int i = 0; i < H * W; ++i) if (frame[i] < 0.0f) frame[i] = 0.0f; /* :22 */
    free(done);
    *prev_ts = last;
    return ORC_OK;
}

/* Dense (frame-based) forward of the same chain on a given input frame [1,H,W].
 * double_act == 0: the comparator of test_correctness.py:73-80 (activation applied once, TF semantics);
 * double_act != 0: the reference's dense NumPy model, src/models/frame_numpy.py:62-88, which applies the activation
 *                  np.maximum(x, x*alpha) (functional.py:44) again after every pool layer (:77-78).
 * Writes each layer's feature map into outs[li] (may be NULL to skip). */
int orc_dense_forward2(const orc_net *n, const float *frame, float **outs, int double_act)
{
    float q5_alpha = 0.0f; /* the model applies ONE actfn everywhere (frame_numpy.py:93): the conv layers' alpha */
    for (int li = 1; li < n->n; ++li) if (n->L[li].kind == ORC_CONV) { q5_alpha = n->L[li].alpha; break; }
    const orc_layer *l0 = &n->L[0];
    size_t cur_sz = (size_t)l0->H * l0->W;
    float *cur = (float *)malloc(cur_sz * sizeof(float));
    memcpy(cur, frame, cur_sz * sizeof(float));
    if (outs && outs[0]) memcpy(outs[0], cur, cur_sz * sizeof(float));
    int C = 1, H = l0->H, W = l0->W;
    for (int li = 1; li < n->n; ++li) {
        const orc_layer *l = &n->L[li];
        const size_t osz = (size_t)l->C * l->H * l->W;
        float *nxt = (float *)malloc(osz * sizeof(float));
        if (l->kind == ORC_CONV) {
            const int Hp = H + l->pt + l->pb, Wp = W + l->pl + l->pr, K = C * l->kh * l->kw;
            float *padded = (float *)malloc((size_t)C * Hp * Wp * sizeof(float));
            pad_map(cur, C, H, W, l->pt, l->pb, l->pl, l->pr, padded);
            float *cols = (float *)malloc((size_t)K * l->H * l->W * sizeof(float));
            for (int oy = 0; oy < l->H; ++oy)
                for (int ox = 0; ox < l->W; ++ox)
                    for (int c = 0; c < C; ++c)
                        for (int dy = 0; dy < l->kh; ++dy)
                            for (int dx = 0; dx < l->kw; ++dx)
                                cols[(size_t)(oy * l->W + ox) * K + (c * l->kh + dy) * l->kw + dx] =
                                    padded[((size_t)c * Hp + oy + dy) * Wp + ox + dx];
            gemm_cols(l->wrows, l->bias, l->C, K, l->kh * l->kw, cols, l->H * l->W, nxt);
            for (size_t i = 0; i < osz; ++i) nxt[i] = nxt[i] * slope_of(nxt[i], l->alpha);
            free(cols); free(padded);
        } else {
            for (int c = 0; c < l->C; ++c)
                for (int y = 0; y < l->H; ++y)
                    for (int x = 0; x < l->W; ++x) {
                        float bv = cur[((size_t)c * H + y * l->stride) * W + x * l->stride];
                        for (int j = 1; j < l->kh * l->kw; ++j) {
                            const float v = cur[((size_t)c * H + y * l->stride + j / l->kw) * W + x * l->stride + j % l->kw];
                            if (v > bv) bv = v;
                        }
                        nxt[((size_t)c * l->H + y) * l->W + x] = bv;
                    }
            if (double_act) /* frame_numpy.py:77-78: actfn(maxpool(x)), actfn = maximum(x, x*alpha) */
                for (size_t i = 0; i < osz; ++i) { const float v = nxt[i], va = v * q5_alpha; nxt[i] = v > va ? v : va; }
        }
        if (outs && outs[li]) memcpy(outs[li], nxt, osz * sizeof(float));
        free(cur);
        cur = nxt; C = l->C; H = l->H; W = l->W;
    }
    free(cur);
    return ORC_OK;
}

int orc_dense_forward(const orc_net *n, const float *frame, float **outs) { return orc_dense_forward2(n, frame, outs, 0); }

/* ------------------------------------------------------------------------- */
/* The free functions callers import by name                                   */
/* ------------------------------------------------------------------------- */
/* conv2d_event  (src/layers/conv2d.py:144-181): im2col_event, then kernel_rows . img_cols (+ bias).
 * kernel_oihw [Cout,C,kh,kw] -- reshape(out_channel, -1) gives exactly the rows gemm_cols wants (conv2d.py:173).
 * out_conv [Cout][n] C order (capacity Cout*outH*outW), out_y/out_x first-touch order. */
int orc_conv2d_event(const float *image, int C, int H, int W, const int32_t *ey, const int32_t *ex, int N,
                     const float *kernel_oihw, const float *bias, int Cout, int kh, int kw, int stride,
                     float *out_conv, int32_t *out_y, int32_t *out_x, int *n_out)
{
    if (!(stride == 1 || (stride == kh && stride == kw))) return ORC_E_STRIDE;
    const int oh = (H - kh) / stride + 1, ow = (W - kw) / stride + 1, K = C * kh * kw;
    float *cols = (float *)malloc((size_t)K * (oh * ow > 0 ? oh * ow : 1) * sizeof(float));
    if (!cols) return ORC_E_ALLOC;
    int n = 0;
    const int rc = orc_im2col_event(image, C, H, W, ey, ex, N, kh, kw, stride, 0, cols, out_y, out_x, &n); /* :172 */
    if (rc == ORC_OK && n > 0) gemm_cols(kernel_oihw, bias, Cout, K, kh * kw, cols, n, out_conv);          /* :175-178 */
    free(cols);
    if (rc == ORC_OK) *n_out = n;
    return rc;
}

/* conv2d  (src/layers/conv2d.py:184-230): optional zero padding (the caller evaluates the SAME rule, :199-213),
 * dense strided im2col (functional.py:4-34), the same contraction.  image [B,C,H,W] -> out [B,Cout,oh,ow]. */
int orc_conv2d(const float *image, int B, int C, int H, int W, const float *kernel_oihw, const float *bias, int Cout,
               int kh, int kw, int pt, int pb, int pl, int pr, int stride, float *out)
{
    const int Hp = H + pt + pb, Wp = W + pl + pr, K = C * kh * kw;
    if (Hp < kh || Wp < kw || stride <= 0) return ORC_E_SHAPE;
    const int oh = (Hp - kh) / stride + 1, ow = (Wp - kw) / stride + 1;
    float *padded = (float *)malloc((size_t)C * Hp * Wp * sizeof(float));
    float *cols = (float *)malloc((size_t)K * oh * ow * sizeof(float));
    if (!padded || !cols) { free(padded); free(cols); return ORC_E_ALLOC; }
    for (int b = 0; b < B; ++b) {
        pad_map(image + (size_t)b * C * H * W, C, H, W, pt, pb, pl, pr, padded);
        for (int oy = 0; oy < oh; ++oy)
            for (int ox = 0; ox < ow; ++ox)
                for (int c = 0; c < C; ++c)
                    for (int dy = 0; dy < kh; ++dy)
                        for (int dx = 0; dx < kw; ++dx)
                            cols[(size_t)(oy * ow + ox) * K + (c * kh + dy) * kw + dx] =
                                padded[((size_t)c * Hp + oy * stride + dy) * Wp + ox * stride + dx];
        gemm_cols(kernel_oihw, bias, Cout, K, kh * kw, cols, oh * ow, out + (size_t)b * Cout * oh * ow);
    }
    free(padded); free(cols);
    return ORC_OK;
}

/* im2col  (src/layers/functional.py:4-34): [B,C,H,W] -> [C*kh*kw][B*oh*ow] C order; elements copied as raw bytes. */
int orc_im2col(const void *image, int elem, int B, int C, int H, int W, int kh, int kw, int stride, void *out)
{
    if (H < kh || W < kw || stride <= 0) return ORC_E_SHAPE;
    const int oh = (H - kh) / stride + 1, ow = (W - kw) / stride + 1;
    const size_t cols = (size_t)B * oh * ow;
    const char *src = (const char *)image;
    char *dst = (char *)out;
    for (int c = 0; c < C; ++c)
        for (int dy = 0; dy < kh; ++dy)
            for (int dx = 0; dx < kw; ++dx)
                for (int b = 0; b < B; ++b)
                    for (int oy = 0; oy < oh; ++oy)
                        for (int ox = 0; ox < ow; ++ox)
                            memcpy(dst + (((size_t)(c * kh + dy) * kw + dx) * cols + ((size_t)b * oh + oy) * ow + ox) * elem,
                                   src + ((((size_t)b * C + c) * H + oy * stride + dy) * W + ox * stride + dx) * elem, elem);
    return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* NEXT ROW: YOLO decode + NMS, the step right after the path   */
/* ------------------------------------------------------------------------- */
/* Parity: NMS pinned by the reference's own apply_nms (tests/golden/nms.npz); the box geometry of the decode pinned by the
 * reference's own convert_bboxes (tests/golden/bbox_decode.npz; viz.py's top-level `import cv2` is satisfied by an empty
 * placeholder, the function is pure NumPy); class argmax / validity / masking restate viz.py:141-164, whose enclosing
 * function ends in OpenCV drawing and cannot run here. */
/* apply_nms for one batch entry, src/libs/utils.py:38-118.  boxes [n][4] = centre x, y, w, h; picks by descending
 * score (ties: larger index first, see below); order[] receives the original indices in pick order. */
int orc_nms(const float *boxes, const float *scores, const uint8_t *valid, int n, float iou_thr, int32_t *order, int *n_picked)
{
    uint8_t *alive = (uint8_t *)malloc(n > 0 ? n : 1);
    for (int i = 0; i < n; ++i) alive[i] = valid ? valid[i] != 0 : 1;
    int np_ = 0;
    for (;;) { /* utils.py:84-110 */
        int i = -1;
        for (int j = 0; j < n; ++j)
            if (alive[j] && (i < 0 || scores[j] > scores[i] || (scores[j] == scores[i] && j > i))) i = j;
        if (i < 0) break;
        alive[i] = 0;
        order[np_++] = i;
        const float xi = boxes[4 * i], yi = boxes[4 * i + 1], wi = boxes[4 * i + 2], hi = boxes[4 * i + 3];
        const float ai = wi * hi;
        for (int j = 0; j < n; ++j) {
            if (!alive[j]) continue;
            const float xj = boxes[4 * j], yj = boxes[4 * j + 1], wj = boxes[4 * j + 2], hj = boxes[4 * j + 3];
            const float xx1 = fmaxf(xi - wi / 2, xj - wj / 2), yy1 = fmaxf(yi - hi / 2, yj - hj / 2);
            const float xx2 = fminf(xi + wi / 2, xj + wj / 2), yy2 = fminf(yi + hi / 2, yj + hj / 2);
            const float ww = fmaxf(0.0f, xx2 - xx1), hh = fmaxf(0.0f, yy2 - yy1);
            const float inter = ww * hh;
            const float iou = inter / (wj * hj + ai - inter);
            if (iou >= iou_thr) alive[j] = 0; /* utils.py:109-110 */
        }
    }
    free(alive);
    *n_picked = np_;
    return ORC_OK;
}

/* Restates draw_bboxes up to (not including) the drawing: src/libs/viz.py:125-164, with convert_bboxes
 * viz.py:27-46 (sqrt = True, viz.py:140) and apply_nms src/libs/utils.py:38-118.  float32 arithmetic as NumPy
 * performs it on the float32 prediction tensor.
 *   pred   [grid_h][grid_w][C + 5*B]     (the model output, event_numpy.py:101)
 *   out    [nb = grid_h*grid_w*B][7]  = x, y, w, h (pixels, centre form), conf, class index, valid (0/1)
 *   order  [nb] (optional) original indices in the order NMS picked them, n_picked of them
 * NMS ties: boxes are visited by descending score, equal scores by descending index (what a stable ascending
 * argsort read from the end gives; NumPy's default sort leaves tie order unspecified, utils.py:81). */
int orc_yolo_detect(const float *pred, int grid_h, int grid_w, int C, int B, int h_image, int w_image, float conf_thr,
                    int use_nms, float iou_thr, float *out, int32_t *order, int *n_picked)
{
    const int nb = grid_h * grid_w * B, D = C + 5 * B;
    int any_valid = 0;
    for (int i = 0; i < nb; ++i) {
        const int cell = i / B, b = i % B, row = cell / grid_w, col = cell % grid_w;
        const float *p = pred + (size_t)cell * D + C + b * 5;
        float *o = out + (size_t)i * 7;
        o[0] = ((p[0] + (float)col) / (float)grid_w) * (float)w_image; /* viz.py:40 */
        o[1] = ((p[1] + (float)row) / (float)grid_h) * (float)h_image; /* viz.py:41 */
        o[2] = (p[2] * p[2]) * (float)w_image;                         /* viz.py:42 (sqrt) */
        o[3] = (p[3] * p[3]) * (float)h_image;                         /* viz.py:43 */
        o[4] = p[4];                                                   /* viz.py:149 */
        int best = 0;
        float bv = pred[(size_t)cell * D] * p[4];                      /* viz.py:144 label * conf, argmax :170 */
        for (int c = 1; c < C; ++c) {
            const float v = pred[(size_t)cell * D + c] * p[4];
            if (v > bv) { bv = v; best = c; }
        }
        o[5] = (float)best;
        o[6] = p[4] > conf_thr ? 1.0f : 0.0f;                          /* viz.py:150 */
        any_valid |= o[6] != 0.0f;
    }
    if (n_picked) *n_picked = 0;
    if (!(use_nms && any_valid)) return ORC_OK;                        /* viz.py:152 */
    uint8_t *picked = (uint8_t *)calloc(nb, 1);
    float *bx = (float *)malloc((size_t)nb * 4 * sizeof(float)), *sc = (float *)malloc(nb * sizeof(float));
    uint8_t *vl = (uint8_t *)malloc(nb);
    int32_t *ord = (int32_t *)malloc(nb * sizeof(int32_t));
    for (int i = 0; i < nb; ++i) {
        memcpy(bx + 4 * i, out + (size_t)i * 7, 4 * sizeof(float));
        sc[i] = out[(size_t)i * 7 + 4];
        vl[i] = out[(size_t)i * 7 + 6] != 0.0f;
    }
    int np_ = 0;
    orc_nms(bx, sc, vl, nb, iou_thr, ord, &np_);
    for (int k = 0; k < np_; ++k) { picked[ord[k]] = 1; if (order) order[k] = ord[k]; }
    free(bx); free(sc); free(vl); free(ord);
    for (int i = 0; i < nb; ++i)
        if (!picked[i]) { float *o = out + (size_t)i * 7; o[0] = o[1] = o[2] = o[3] = o[4] = 0.0f; o[6] = 0.0f; } /* viz.py:154-164 */
    if (n_picked) *n_picked = np_;
    free(picked);
    return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* NEXT ROW: N-MNIST / N-Caltech101 binary decoder              */
/* ------------------------------------------------------------------------- */
/* Restates NReader.read_example, src/readers/file_reader.py:30-58.  Parity PINNED: tests/golden/nformat.npz holds byte
 * streams and what the reference's own NReader.read_example returns for them (tests/golden/gen_golden.py: gen_nformat --
 * the function is pure NumPy; the two modules its file imports without using them here, bitstring and numba, are absent
 * and are satisfied by empty placeholders); tests/test_oracle_golden.py compares, bit for bit.
 * 5 bytes per event: x, y, (p << 7 | ts[22:16]), ts[15:8], ts[7:0]; an event with y == 240 is a timestamp-overflow
 * marker: it adds 2^13 to its own and every later timestamp (:44-47) and is dropped (:50). */
long long orc_decode_nformat(const uint8_t *raw, long long nbytes, int32_t *x, int32_t *y, int32_t *ts, int32_t *p)
{
    const long long n = nbytes / 5; /* raw[k::5] slices: a trailing partial record contributes to the early slices only */
    long long m = 0;
    uint32_t carry = 0;
    for (long long i = 0; i < n; ++i) {
        const uint8_t *r = raw + 5 * i;
        if (r[1] == 240) carry += 1u << 13; /* :44-47: all_ts[overflow_index:] += 2**13 */
        const uint32_t t = ((((uint32_t)r[2] & 127u) << 16) | ((uint32_t)r[3] << 8) | (uint32_t)r[4]) + carry; /* :41 */
        if (r[1] != 240) { /* :50 */
            x[m] = r[0]; y[m] = r[1]; ts[m] = (int32_t)t; p[m] = (r[2] & 128) >> 7; /* :38-40 */
            ++m;
        }
    }
    return m;
}
