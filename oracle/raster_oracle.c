/*
 * oracle/raster_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, float32, OpenMP-optional) of the tile-based
 * alpha-compositing Gaussian rasterizer that DreamMesh4D calls as
 *   diff_gaussian_rasterization.GaussianRasterizer(...)(means3D, means2D, ...)
 * at  custom/threestudio-dreammesh4d/renderer/diff_sugar_rasterizer_temporal.py:129-178
 * (RGB pass) and :202-211 (normal pass), and at
 *   custom/threestudio-dreammesh4d/renderer/diff_sugar_rasterizer_normal.py:117-132,161-170,186-195.
 *
 * The algorithm itself lives in a third-party dependency that is ABSENT from
 * /root/reference: `diff-gaussian-rasterization`, ashawkey fork (depth + alpha
 * outputs), un-pinned (requirements.txt:49, README.md:35).  This file restates
 * its published algorithm (3D Gaussian Splatting, Kerbl et al. 2023, plus the
 * fork's depth/alpha channels): preprocess -> duplicate-with-keys -> stable
 * radix sort on (tile<<32 | depth_bits) -> tile ranges -> front-to-back blend,
 * and the analytic backward.
 *
 * PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures for
 * this path and the CUDA package cannot be built or run
 * here.  The oracle is instead pinned (tests/test_oracle_raster.py) against
 *   - an independent dense fp64 PyTorch restatement with autograd (forward
 *     values and every gradient),
 *   - closed-form single-splat cases,
 *   - invariants (permutation of the input order, alpha == 1 - final_T).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file.  The product path (dreammesh4d_amd/) never does.
 *
 * Arithmetic contract shared with the HIP kernels (DESIGN.md "arithmetic
 * contract"): float32, no FMA contraction except where fmaf() is written,
 * correctly rounded / and sqrt, and the polynomial exp below -- so that tile
 * keys, radii, n_contrib and the forward image can be compared bit-for-bit.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16            /* BLOCK_X == BLOCK_Y == 16 in the upstream rasterizer */
#define NCH 3

typedef struct {
    int32_t N, W, H;
    int32_t sh_degree;     /* only 0 is supported; M = 1 coefficient */
    int32_t prefiltered;
    float tanfovx, tanfovy, scale_modifier;
    const float *bg;             /* [3] */
    const float *means3D;        /* [N,3] */
    const float *colors_precomp; /* [N,3] or NULL */
    const float *shs;            /* [N,M,3] or NULL (M = (deg+1)^2) */
    const float *opacities;      /* [N] */
    const float *scales;         /* [N,3] or NULL */
    const float *rotations;      /* [N,4] (w,x,y,z) or NULL */
    const float *cov3D_precomp;  /* [N,6] or NULL */
    const float *viewmatrix;     /* [16] row-vector convention: flat[j*4+i] */
    const float *projmatrix;     /* [16] */
    const float *campos;         /* [3] */
} dm4d_oracle_in;

typedef struct {
    /* image outputs */
    float *out_color;   /* [3,H,W] */
    float *out_depth;   /* [H,W]   */
    float *out_alpha;   /* [H,W]   */
    int32_t *radii;     /* [N]     */
    /* per-Gaussian state */
    float *xy;             /* [N,2] pixel centre */
    float *depths;         /* [N]   view-space z */
    float *conic_opacity;  /* [N,4] */
    float *rgb;            /* [N,3] colour actually blended */
    float *cov3D;          /* [N,6] */
    uint8_t *clamped;      /* [N,3] SH clamp flags */
    uint32_t *tiles_touched; /* [N] */
    /* binning state */
    uint64_t *keys;        /* [cap] sorted keys  tile<<32 | depth_bits */
    uint32_t *values;      /* [cap] sorted Gaussian ids */
    int64_t cap;
    uint32_t *ranges;      /* [tiles,2] */
    /* per-pixel state */
    uint32_t *n_contrib;   /* [H,W] */
    float *final_T;        /* [H,W] */
} dm4d_oracle_state;

typedef struct {
    float *dL_dmeans2D;   /* [N,3] (z = 0) */
    float *dL_dconic;     /* [N,3] true dL/d(A,B,C) of the conic, for inspection */
    float *dL_dopacity;   /* [N] */
    float *dL_dcolors;    /* [N,3] (w.r.t. colors_precomp, or the SH-evaluated rgb) */
    float *dL_ddepths;    /* [N] */
    float *dL_dmeans3D;   /* [N,3] */
    float *dL_dcov3D;     /* [N,6] */
    float *dL_dsh;        /* [N,M,3] or NULL */
    float *dL_dscales;    /* [N,3] */
    float *dL_drots;      /* [N,4] */
} dm4d_oracle_grads;

/* ------------------------------------------------------------------ */
/* deterministic exp for x <= 0 : shared arithmetic contract            */
/* 2^f minimax (degree 6) on [-0.5,0.5]; <= 1.4 ulp                     */
/* Round 4: the same polynomial, with the two steps the GPU paid most   */
/* for restated in operations it has as ONE instruction each (the       */
/* forward blend kernel's time is its instruction count):               */
/*   n = rint(x log2 e)   ->  t = fma(x, log2e_hi, 1.5 * 2^23);         */
/*                            n = t - 1.5 * 2^23   (exact; t's low      */
/*                            mantissa bits hold n in two's complement) */
/*   ldexp(p, n)          ->  bits(p) + (bits(t) << 23): n added to p's */
/*                            exponent field (p in [0.70, 1.42], n >=   */
/*                            -124 after the clamp at -86: no underflow)*/
/* n is now the rounding of the EXACT product (one rounding), where     */
/* rint(float(x log2 e)) rounded twice: results differ from round 3's   */
/* in the last bits for some x -- both sides of every comparison use    */
/* this function.  exp(x) for x < -86 returns exp(-86) = 4.5e-38 (was   */
/* -87): far below any alpha >= 1/255.                                  */
/* ------------------------------------------------------------------ */
static inline float dm4d_expf(float x)
{
    const float L2E_HI = 0x1.715476p+0f;   /* float(log2 e) */
    const float L2E_LO = 0x1.4ae0c0p-26f;  /* log2 e - L2E_HI */
    const float MAGIC = 12582912.0f;       /* 1.5 * 2^23 */
    x = fmaxf(x, -86.0f);
    float t = fmaf(x, L2E_HI, MAGIC);
    float n = t - MAGIC;
    float f = fmaf(x, L2E_HI, -n);
    f = fmaf(x, L2E_LO, f);
    float p = 0x1.446c7ep-13f;
    p = fmaf(p, f, 0x1.5f48c8p-10f);
    p = fmaf(p, f, 0x1.3b29d8p-7f);
    p = fmaf(p, f, 0x1.c6aeccp-5f);
    p = fmaf(p, f, 0x1.ebfbe0p-3f);
    p = fmaf(p, f, 0x1.62e430p-1f);
    p = fmaf(p, f, 1.0f);
    uint32_t pb, tb;
    memcpy(&pb, &p, 4);
    memcpy(&tb, &t, 4);
    pb += tb << 23;
    memcpy(&p, &pb, 4);
    return p;
}

float dm4d_oracle_expf(float x) { return dm4d_expf(x); }

static inline int f2i_sat(float v)
{
    if (!(v > -1073741824.0f)) return -1073741824;   /* also catches NaN */
    if (v > 1073741824.0f) return 1073741824;
    return (int)v;
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* row-vector convention: x' = M[0] x + M[4] y + M[8] z + M[12] */
static inline void xform4x3(const float *p, const float *M, float *o)
{
    o[0] = ((M[0] * p[0] + M[4] * p[1]) + M[8] * p[2]) + M[12];
    o[1] = ((M[1] * p[0] + M[5] * p[1]) + M[9] * p[2]) + M[13];
    o[2] = ((M[2] * p[0] + M[6] * p[1]) + M[10] * p[2]) + M[14];
}
static inline void xform4x4(const float *p, const float *M, float *o)
{
    o[0] = ((M[0] * p[0] + M[4] * p[1]) + M[8] * p[2]) + M[12];
    o[1] = ((M[1] * p[0] + M[5] * p[1]) + M[9] * p[2]) + M[13];
    o[2] = ((M[2] * p[0] + M[6] * p[1]) + M[10] * p[2]) + M[14];
    o[3] = ((M[3] * p[0] + M[7] * p[1]) + M[11] * p[2]) + M[15];
}

static inline void quat_to_R(const float *q, float R[9])
{
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z);
    R[1] = 2.f * (x * y - r * z);
    R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z);
    R[4] = 1.f - 2.f * (x * x + z * z);
    R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y);
    R[7] = 2.f * (y * z + r * x);
    R[8] = 1.f - 2.f * (x * x + y * y);
}

/* Sigma = (R S)(R S)^T, upper triangle */
static inline void compute_cov3D(const float *scale, float mod, const float *q, float *cov6)
{
    float R[9], M[9];
    quat_to_R(q, R);
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    for (int a = 0; a < 3; ++a)
        for (int j = 0; j < 3; ++j) M[a * 3 + j] = R[a * 3 + j] * s[j];
#define SIG(a, b) ((M[a * 3 + 0] * M[b * 3 + 0] + M[a * 3 + 1] * M[b * 3 + 1]) + M[a * 3 + 2] * M[b * 3 + 2])
    cov6[0] = SIG(0, 0); cov6[1] = SIG(0, 1); cov6[2] = SIG(0, 2);
    cov6[3] = SIG(1, 1); cov6[4] = SIG(1, 2); cov6[5] = SIG(2, 2);
#undef SIG
}

typedef struct { float T[6]; float tz, tcx, tcy; int xclamped, yclamped; } cov2d_aux;

/* EWA projection: cov2D = (J W) Sigma (J W)^T ; returns (c00, c01, c11) WITHOUT the low-pass */
static inline void compute_cov2D(const float *mean, float fx, float fy, float tanfovx, float tanfovy,
                                 const float *cov6, const float *V, float *c3, cov2d_aux *aux)
{
    float t[3];
    xform4x3(mean, V, t);
    float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    float txtz = t[0] / t[2], tytz = t[1] / t[2];
    float cx = fminf(limx, fmaxf(-limx, txtz));
    float cy = fminf(limy, fmaxf(-limy, tytz));
    float tcx = cx * t[2], tcy = cy * t[2];
    float J00 = fx / t[2];
    float J02 = -(fx * tcx) / (t[2] * t[2]);
    float J11 = fy / t[2];
    float J12 = -(fy * tcy) / (t[2] * t[2]);
    /* W_ij = V[j*4+i] */
    float T0[3], T1[3];
    for (int j = 0; j < 3; ++j) {
        T0[j] = J00 * V[j * 4 + 0] + J02 * V[j * 4 + 2];
        T1[j] = J11 * V[j * 4 + 1] + J12 * V[j * 4 + 2];
    }
    float S[9] = {cov6[0], cov6[1], cov6[2], cov6[1], cov6[3], cov6[4], cov6[2], cov6[4], cov6[5]};
    float v0[3], v1[3];
    for (int j = 0; j < 3; ++j) {
        v0[j] = (T0[0] * S[0 * 3 + j] + T0[1] * S[1 * 3 + j]) + T0[2] * S[2 * 3 + j];
        v1[j] = (T1[0] * S[0 * 3 + j] + T1[1] * S[1 * 3 + j]) + T1[2] * S[2 * 3 + j];
    }
    c3[0] = (v0[0] * T0[0] + v0[1] * T0[1]) + v0[2] * T0[2];
    c3[1] = (v0[0] * T1[0] + v0[1] * T1[1]) + v0[2] * T1[2];
    c3[2] = (v1[0] * T1[0] + v1[1] * T1[1]) + v1[2] * T1[2];
    if (aux) {
        for (int j = 0; j < 3; ++j) { aux->T[j] = T0[j]; aux->T[3 + j] = T1[j]; }
        aux->tz = t[2]; aux->tcx = tcx; aux->tcy = tcy;
        aux->xclamped = (txtz < -limx || txtz > limx);
        aux->yclamped = (tytz < -limy || tytz > limy);
    }
}

static inline void get_rect(float px, float py, int r, int gx, int gy, int *mn, int *mx)
{
    float fr = (float)r;
    mn[0] = imin(gx, imax(0, f2i_sat((px - fr) / (float)TILE)));
    mn[1] = imin(gy, imax(0, f2i_sat((py - fr) / (float)TILE)));
    mx[0] = imin(gx, imax(0, f2i_sat((px + fr + (float)(TILE - 1)) / (float)TILE)));
    mx[1] = imin(gy, imax(0, f2i_sat((py + fr + (float)(TILE - 1)) / (float)TILE)));
}

#define SH_C0 0.28209479177387814f

static void preprocess_one(const dm4d_oracle_in *in, dm4d_oracle_state *st, int i, int gx, int gy, float fx, float fy)
{
    st->radii[i] = 0;
    st->tiles_touched[i] = 0;
    const float *p = in->means3D + 3 * i;
    float pv[3];
    xform4x3(p, in->viewmatrix, pv);
    if (pv[2] <= 0.2f) return;
    float ph[4];
    xform4x4(p, in->projmatrix, ph);
    float pw = 1.0f / (ph[3] + 0.0000001f);
    float ppx = ph[0] * pw, ppy = ph[1] * pw;

    float *cov6 = st->cov3D + 6 * i;
    if (in->cov3D_precomp) {
        memcpy(cov6, in->cov3D_precomp + 6 * i, 6 * sizeof(float));
    } else {
        compute_cov3D(in->scales + 3 * i, in->scale_modifier, in->rotations + 4 * i, cov6);
    }
    float c[3];
    compute_cov2D(p, fx, fy, in->tanfovx, in->tanfovy, cov6, in->viewmatrix, c, NULL);
    c[0] += 0.3f;
    c[2] += 0.3f;
    float det = c[0] * c[2] - c[1] * c[1];
    if (det == 0.0f) return;
    float det_inv = 1.f / det;
    float conic[3] = {c[2] * det_inv, -c[1] * det_inv, c[0] * det_inv};
    float mid = 0.5f * (c[0] + c[2]);
    float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
    float lambda1 = mid + disc, lambda2 = mid - disc;
    int my_radius = f2i_sat(ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2))));
    float pix[2] = {((ppx + 1.0f) * (float)in->W - 1.0f) * 0.5f, ((ppy + 1.0f) * (float)in->H - 1.0f) * 0.5f};
    int mn[2], mx[2];
    get_rect(pix[0], pix[1], my_radius, gx, gy, mn, mx);
    if ((mx[0] - mn[0]) * (mx[1] - mn[1]) == 0) return;

    float *rgb = st->rgb + 3 * i;
    if (in->colors_precomp) {
        rgb[0] = in->colors_precomp[3 * i]; rgb[1] = in->colors_precomp[3 * i + 1]; rgb[2] = in->colors_precomp[3 * i + 2];
        st->clamped[3 * i] = st->clamped[3 * i + 1] = st->clamped[3 * i + 2] = 0;
    } else {
        int M = (in->sh_degree + 1) * (in->sh_degree + 1);
        const float *sh = in->shs + (size_t)i * M * 3;
        for (int ch = 0; ch < 3; ++ch) {
            float v = SH_C0 * sh[ch] + 0.5f;
            st->clamped[3 * i + ch] = (v < 0.f);
            rgb[ch] = fmaxf(v, 0.f);
        }
    }
    st->depths[i] = pv[2];
    st->radii[i] = my_radius;
    st->xy[2 * i] = pix[0];
    st->xy[2 * i + 1] = pix[1];
    st->conic_opacity[4 * i + 0] = conic[0];
    st->conic_opacity[4 * i + 1] = conic[1];
    st->conic_opacity[4 * i + 2] = conic[2];
    st->conic_opacity[4 * i + 3] = in->opacities[i];
    st->tiles_touched[i] = (uint32_t)((mx[0] - mn[0]) * (mx[1] - mn[1]));
}

/* stable LSD radix sort of (key,value) on bits [0,nbits) -- what cub::DeviceRadixSort::SortPairs does */
static void radix_sort_pairs(uint64_t *keys, uint32_t *vals, int64_t n, int nbits)
{
    uint64_t *k2 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
    uint32_t *v2 = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    uint64_t *ka = keys, *kb = k2;
    uint32_t *va = vals, *vb = v2;
    for (int shift = 0; shift < nbits; shift += 8) {
        int64_t hist[257];
        memset(hist, 0, sizeof(hist));
        for (int64_t i = 0; i < n; ++i) hist[((ka[i] >> shift) & 0xFF) + 1]++;
        for (int b = 0; b < 256; ++b) hist[b + 1] += hist[b];
        for (int64_t i = 0; i < n; ++i) {
            int64_t d = hist[(ka[i] >> shift) & 0xFF]++;
            kb[d] = ka[i];
            vb[d] = va[i];
        }
        uint64_t *tk = ka; ka = kb; kb = tk;
        uint32_t *tv = va; va = vb; vb = tv;
    }
    if (ka != keys) {
        memcpy(keys, ka, sizeof(uint64_t) * (size_t)n);
        memcpy(vals, va, sizeof(uint32_t) * (size_t)n);
    }
    free(k2);
    free(v2);
}

static void render_tile(const dm4d_oracle_in *in, dm4d_oracle_state *st, int tx, int ty, int gx)
{
    int W = in->W, H = in->H;
    uint32_t start = st->ranges[2 * (ty * gx + tx)], end = st->ranges[2 * (ty * gx + tx) + 1];
    for (int ly = 0; ly < TILE; ++ly)
        for (int lx = 0; lx < TILE; ++lx) {
            int px = tx * TILE + lx, py = ty * TILE + ly;
            if (px >= W || py >= H) continue;
            float pixf[2] = {(float)px, (float)py};
            float T = 1.0f, C[NCH] = {0, 0, 0}, D = 0.f, Wt = 0.f;
            uint32_t contributor = 0, last = 0;
            for (uint32_t e = start; e < end; ++e) {
                contributor++;
                uint32_t g = st->values[e];
                float dx = st->xy[2 * g] - pixf[0], dy = st->xy[2 * g + 1] - pixf[1];
                const float *co = st->conic_opacity + 4 * g;
                float power = -0.5f * ((co[0] * dx) * dx + (co[2] * dy) * dy) - (co[1] * dx) * dy;
                if (power > 0.0f) continue;
                float alpha = fminf(0.99f, co[3] * dm4d_expf(power));
                if (alpha < 1.0f / 255.0f) continue;
                float test_T = T * (1.0f - alpha);
                if (test_T < 0.0001f) break;
                float w = alpha * T;
                for (int ch = 0; ch < NCH; ++ch) C[ch] = fmaf(st->rgb[3 * g + ch], w, C[ch]);
                D = fmaf(st->depths[g], w, D);
                Wt = Wt + w;
                T = test_T;
                last = contributor;
            }
            int pid = py * W + px;
            st->final_T[pid] = T;
            st->n_contrib[pid] = last;
            for (int ch = 0; ch < NCH; ++ch) st->out_color[ch * H * W + pid] = fmaf(T, in->bg[ch], C[ch]);
            st->out_depth[pid] = D;
            st->out_alpha[pid] = Wt;
        }
}

static int nbits_for(int tiles)
{
    /* getHigherMsb(tile_grid.x * tile_grid.y) */
    int msb = 0;
    uint32_t n = (uint32_t)tiles;
    while (n >> msb) msb++;  /* number of bits needed */
    return msb;
}

/* Returns num_rendered (D).  If st->cap < D only the preprocess state is filled. */
int64_t dm4d_oracle_rasterize_forward(const dm4d_oracle_in *in, dm4d_oracle_state *st)
{
    int N = in->N, W = in->W, H = in->H;
    int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    float fy = (float)H / (2.0f * in->tanfovy);
    float fx = (float)W / (2.0f * in->tanfovx);
    if (in->sh_degree != 0 && !in->colors_precomp) return -2;

#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) preprocess_one(in, st, i, gx, gy, fx, fy);

    int64_t D = 0;
    for (int i = 0; i < N; ++i) D += st->tiles_touched[i];
    if (D > st->cap) return D;

    /* duplicateWithKeys: Gaussian-major, rect scanned y-major */
    int64_t off = 0;
    for (int i = 0; i < N; ++i) {
        if (st->radii[i] <= 0) continue;
        int mn[2], mx[2];
        get_rect(st->xy[2 * i], st->xy[2 * i + 1], st->radii[i], gx, gy, mn, mx);
        uint32_t dbits;
        memcpy(&dbits, &st->depths[i], 4);
        for (int y = mn[1]; y < mx[1]; ++y)
            for (int x = mn[0]; x < mx[0]; ++x) {
                uint64_t key = (uint64_t)(y * gx + x);
                key <<= 32;
                key |= dbits;
                st->keys[off] = key;
                st->values[off] = (uint32_t)i;
                off++;
            }
    }
    radix_sort_pairs(st->keys, st->values, D, 32 + nbits_for(gx * gy));

    /* identifyTileRanges */
    memset(st->ranges, 0, sizeof(uint32_t) * 2 * (size_t)(gx * gy));
    for (int64_t e = 0; e < D; ++e) {
        uint32_t tile = (uint32_t)(st->keys[e] >> 32);
        if (e == 0) st->ranges[2 * tile] = 0;
        else {
            uint32_t prev = (uint32_t)(st->keys[e - 1] >> 32);
            if (prev != tile) { st->ranges[2 * prev + 1] = (uint32_t)e; st->ranges[2 * tile] = (uint32_t)e; }
        }
        if (e == D - 1) st->ranges[2 * tile + 1] = (uint32_t)D;
    }

#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int ty = 0; ty < gy; ++ty)
        for (int tx = 0; tx < gx; ++tx) render_tile(in, st, tx, ty, gx);
    return D;
}

/* ------------------------------------------------------------------ */
/* backward                                                             */
/* ------------------------------------------------------------------ */
static void render_tile_backward(const dm4d_oracle_in *in, const dm4d_oracle_state *st, int tx, int ty, int gx,
                                 const float *dL_dpix, const float *dL_ddepth, const float *dL_dalpha_pix,
                                 double *acc /* [D][10]: xy(2) conic(3) opac color(3) depth */)
{
    int W = in->W, H = in->H;
    uint32_t start = st->ranges[2 * (ty * gx + tx)];   /* the walk stops at n_contrib, inside [start, range end) */
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    for (int ly = 0; ly < TILE; ++ly)
        for (int lx = 0; lx < TILE; ++lx) {
            int px = tx * TILE + lx, py = ty * TILE + ly;
            if (px >= W || py >= H) continue;
            int pid = py * W + px;
            float pixf[2] = {(float)px, (float)py};
            float T_final = st->final_T[pid];
            float T = T_final;
            uint32_t last = st->n_contrib[pid];
            float gC[NCH];
            for (int ch = 0; ch < NCH; ++ch) gC[ch] = dL_dpix[ch * H * W + pid];
            float gD = dL_ddepth ? dL_ddepth[pid] : 0.f;
            float gA = dL_dalpha_pix ? dL_dalpha_pix[pid] : 0.f;
            float bgdot = 0.f;
            for (int ch = 0; ch < NCH; ++ch) bgdot += in->bg[ch] * gC[ch];
            /* S = sum_{j>i} V_j alpha_j T_j  (V_j = <attributes_j, pixel grads>) */
            float S = 0.f;
            for (int64_t k = (int64_t)last - 1; k >= 0; --k) {
                uint32_t e = start + (uint32_t)k;
                uint32_t g = st->values[e];
                float dx = st->xy[2 * g] - pixf[0], dy = st->xy[2 * g + 1] - pixf[1];
                const float *co = st->conic_opacity + 4 * g;
                float power = -0.5f * ((co[0] * dx) * dx + (co[2] * dy) * dy) - (co[1] * dx) * dy;
                if (power > 0.0f) continue;
                float G = dm4d_expf(power);
                float alpha = fminf(0.99f, co[3] * G);
                if (alpha < 1.0f / 255.0f) continue;
                T = T / (1.f - alpha);
                float w = alpha * T;
                float V = gA;
                for (int ch = 0; ch < NCH; ++ch) V += st->rgb[3 * g + ch] * gC[ch];
                V += st->depths[g] * gD;
                float dL_dalpha = T * V - (S + T_final * bgdot) / (1.f - alpha);
                S += V * w;
                double *a = acc + (size_t)e * 10;
                for (int ch = 0; ch < NCH; ++ch) a[6 + ch] += (double)(w * gC[ch]);
                a[9] += (double)(w * gD);
                a[5] += (double)(G * dL_dalpha);
                float dL_dG = co[3] * dL_dalpha;
                float gdx = G * dx, gdy = G * dy;
                float dG_ddelx = -gdx * co[0] - gdy * co[1];
                float dG_ddely = -gdy * co[2] - gdx * co[1];
                a[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                a[1] += (double)(dL_dG * dG_ddely * ddely_dy);
                a[2] += (double)(-0.5f * gdx * dx * dL_dG);
                a[3] += (double)(-gdx * dy * dL_dG);     /* true dL/dB (upstream stores half of it) */
                a[4] += (double)(-0.5f * gdy * dy * dL_dG);
            }
        }
}

static void preprocess_backward_one(const dm4d_oracle_in *in, const dm4d_oracle_state *st, int i, float fx, float fy,
                                    dm4d_oracle_grads *g)
{
    float *dmean = g->dL_dmeans3D + 3 * i;
    dmean[0] = dmean[1] = dmean[2] = 0.f;
    float *dcov = g->dL_dcov3D + 6 * i;
    for (int k = 0; k < 6; ++k) dcov[k] = 0.f;
    if (g->dL_dscales) { g->dL_dscales[3 * i] = g->dL_dscales[3 * i + 1] = g->dL_dscales[3 * i + 2] = 0.f; }
    if (g->dL_drots) { for (int k = 0; k < 4; ++k) g->dL_drots[4 * i + k] = 0.f; }
    if (g->dL_dsh) { for (int k = 0; k < 3; ++k) g->dL_dsh[3 * i + k] = 0.f; }
    if (st->radii[i] <= 0) return;
    const float *V = in->viewmatrix, *P = in->projmatrix;
    const float *m = in->means3D + 3 * i;
    const float *cov6 = st->cov3D + 6 * i;

    /* ---- cov2D / conic backward ---- */
    float c[3];
    cov2d_aux aux;
    compute_cov2D(m, fx, fy, in->tanfovx, in->tanfovy, cov6, V, c, &aux);
    float a = c[0] + 0.3f, b = c[1], cc = c[2] + 0.3f;
    float denom = a * cc - b * b;
    float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float gA = g->dL_dconic[3 * i], gB = g->dL_dconic[3 * i + 1], gC = g->dL_dconic[3 * i + 2];
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    if (denom2inv != 0) {
        dL_da = denom2inv * (-cc * cc * gA + b * cc * gB + (denom - a * cc) * gC);
        dL_dc = denom2inv * (-a * a * gC + a * b * gB + (denom - a * cc) * gA);
        dL_db = denom2inv * (2 * b * cc * gA - (denom + 2 * b * b) * gB + 2 * a * b * gC);
        const float *T0 = aux.T, *T1 = aux.T + 3;
        dcov[0] = T0[0] * T0[0] * dL_da + T0[0] * T1[0] * dL_db + T1[0] * T1[0] * dL_dc;
        dcov[3] = T0[1] * T0[1] * dL_da + T0[1] * T1[1] * dL_db + T1[1] * T1[1] * dL_dc;
        dcov[5] = T0[2] * T0[2] * dL_da + T0[2] * T1[2] * dL_db + T1[2] * T1[2] * dL_dc;
        dcov[1] = 2 * T0[0] * T0[1] * dL_da + (T0[0] * T1[1] + T0[1] * T1[0]) * dL_db + 2 * T1[0] * T1[1] * dL_dc;
        dcov[2] = 2 * T0[0] * T0[2] * dL_da + (T0[0] * T1[2] + T0[2] * T1[0]) * dL_db + 2 * T1[0] * T1[2] * dL_dc;
        dcov[4] = 2 * T0[2] * T0[1] * dL_da + (T0[1] * T1[2] + T0[2] * T1[1]) * dL_db + 2 * T1[1] * T1[2] * dL_dc;
    }
    {
        const float *T0 = aux.T, *T1 = aux.T + 3;
        float S[9] = {cov6[0], cov6[1], cov6[2], cov6[1], cov6[3], cov6[4], cov6[2], cov6[4], cov6[5]};
        float dT0[3], dT1[3];
        for (int k = 0; k < 3; ++k) {
            float s0 = S[k * 3 + 0] * T0[0] + S[k * 3 + 1] * T0[1] + S[k * 3 + 2] * T0[2];
            float s1 = S[k * 3 + 0] * T1[0] + S[k * 3 + 1] * T1[1] + S[k * 3 + 2] * T1[2];
            dT0[k] = 2 * s0 * dL_da + s1 * dL_db;
            dT1[k] = 2 * s1 * dL_dc + s0 * dL_db;
        }
        float dJ00 = 0, dJ02 = 0, dJ11 = 0, dJ12 = 0;
        for (int k = 0; k < 3; ++k) {
            dJ00 += V[k * 4 + 0] * dT0[k];
            dJ02 += V[k * 4 + 2] * dT0[k];
            dJ11 += V[k * 4 + 1] * dT1[k];
            dJ12 += V[k * 4 + 2] * dT1[k];
        }
        float tz = 1.f / aux.tz, tz2 = tz * tz, tz3 = tz2 * tz;
        float xm = aux.xclamped ? 0.f : 1.f, ym = aux.yclamped ? 0.f : 1.f;
        float dtx = xm * -fx * tz2 * dJ02;
        float dty = ym * -fy * tz2 * dJ12;
        float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * aux.tcx) * tz3 * dJ02 + (2 * fy * aux.tcy) * tz3 * dJ12;
        for (int j = 0; j < 3; ++j) dmean[j] = V[j * 4 + 0] * dtx + V[j * 4 + 1] * dty + V[j * 4 + 2] * dtz;
    }
    /* ---- projection of the mean: NDC-space grads ---- */
    {
        float ph[4];
        xform4x4(m, P, ph);
        float mw = 1.0f / (ph[3] + 0.0000001f);
        float mul1 = ph[0] * mw * mw, mul2 = ph[1] * mw * mw;
        float g2x = g->dL_dmeans2D[3 * i], g2y = g->dL_dmeans2D[3 * i + 1];
        for (int j = 0; j < 3; ++j)
            dmean[j] += (P[j * 4 + 0] * mw - P[j * 4 + 3] * mul1) * g2x + (P[j * 4 + 1] * mw - P[j * 4 + 3] * mul2) * g2y;
    }
    /* ---- depth -> mean (fork's depth channel) ---- */
    {
        float gd = g->dL_ddepths[i];
        float mul3 = V[2] * m[0] + V[6] * m[1] + V[10] * m[2] + V[14];
        for (int j = 0; j < 3; ++j) dmean[j] += (V[j * 4 + 2] - V[j * 4 + 3] * mul3) * gd;
    }
    /* ---- colour -> SH (degree 0) ---- */
    if (in->shs && g->dL_dsh) {
        for (int ch = 0; ch < 3; ++ch)
            g->dL_dsh[3 * i + ch] = st->clamped[3 * i + ch] ? 0.f : SH_C0 * g->dL_dcolors[3 * i + ch];
    }
    /* ---- cov3D -> scale, rotation ---- */
    if (in->scales && g->dL_dscales && g->dL_drots) {
        const float *q = in->rotations + 4 * i;
        float R[9];
        quat_to_R(q, R);
        float mod = in->scale_modifier;
        float s[3] = {mod * in->scales[3 * i], mod * in->scales[3 * i + 1], mod * in->scales[3 * i + 2]};
        float Gs[9] = {dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3], 0.5f * dcov[4],
                       0.5f * dcov[2], 0.5f * dcov[4], dcov[5]};
        float dM[9]; /* dL/dM = 2 G M, M = R S */
        for (int aa = 0; aa < 3; ++aa)
            for (int j = 0; j < 3; ++j) {
                float acc2 = 0;
                for (int k = 0; k < 3; ++k) acc2 += Gs[aa * 3 + k] * (R[k * 3 + j] * s[j]);
                dM[aa * 3 + j] = 2.f * acc2;
            }
        float dR[9];
        for (int j = 0; j < 3; ++j) {
            float ds = 0;
            for (int aa = 0; aa < 3; ++aa) { ds += R[aa * 3 + j] * dM[aa * 3 + j]; dR[aa * 3 + j] = dM[aa * 3 + j] * s[j]; }
            g->dL_dscales[3 * i + j] = mod * ds;
        }
        float r = q[0], x = q[1], y = q[2], z = q[3];
        g->dL_drots[4 * i + 0] = 2 * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
        g->dL_drots[4 * i + 1] = 2 * (y * dR[1] + z * dR[2] + y * dR[3] - 2 * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2 * x * dR[8]);
        g->dL_drots[4 * i + 2] = 2 * (-2 * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2 * y * dR[8]);
        g->dL_drots[4 * i + 3] = 2 * (-2 * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2 * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
    }
}

int dm4d_oracle_rasterize_backward(const dm4d_oracle_in *in, const dm4d_oracle_state *st, int64_t D,
                                   const float *dL_dpix, const float *dL_ddepth, const float *dL_dalpha,
                                   dm4d_oracle_grads *g)
{
    int N = in->N, W = in->W, H = in->H;
    int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    float fy = (float)H / (2.0f * in->tanfovy);
    float fx = (float)W / (2.0f * in->tanfovx);
    double *acc = (double *)calloc((size_t)(D > 0 ? D : 1) * 10, sizeof(double));
    if (!acc) return -1;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int ty = 0; ty < gy; ++ty)
        for (int tx = 0; tx < gx; ++tx) render_tile_backward(in, st, tx, ty, gx, dL_dpix, dL_ddepth, dL_dalpha, acc);

    double *gacc = (double *)calloc((size_t)N * 10, sizeof(double));
    if (!gacc) { free(acc); return -1; }
    for (int64_t e = 0; e < D; ++e) {
        uint32_t gi = st->values[e];
        for (int k = 0; k < 10; ++k) gacc[(size_t)gi * 10 + k] += acc[(size_t)e * 10 + k];
    }
    free(acc);
    for (int i = 0; i < N; ++i) {
        const double *a = gacc + (size_t)i * 10;
        g->dL_dmeans2D[3 * i] = (float)a[0];
        g->dL_dmeans2D[3 * i + 1] = (float)a[1];
        g->dL_dmeans2D[3 * i + 2] = 0.f;
        g->dL_dconic[3 * i] = (float)a[2];
        g->dL_dconic[3 * i + 1] = (float)a[3];
        g->dL_dconic[3 * i + 2] = (float)a[4];
        g->dL_dopacity[i] = (float)a[5];
        g->dL_dcolors[3 * i] = (float)a[6];
        g->dL_dcolors[3 * i + 1] = (float)a[7];
        g->dL_dcolors[3 * i + 2] = (float)a[8];
        g->dL_ddepths[i] = (float)a[9];
    }
    free(gacc);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) preprocess_backward_one(in, st, i, fx, fy, g);
    return 0;
}

/* Stage 2 of the backward alone: g->dL_dmeans2D, dL_dconic, dL_dcolors and dL_ddepths are INPUTS (as the blend stage
 * left them, or perturbed by the caller), the per-parameter gradients are recomputed from them.  The parity tests use
 * it to measure how much this ill-conditioned chain (conic -> cov2D -> cov3D -> scale / rotation) amplifies rounding
 * noise of the blend-level sums on the scene at hand. */
int dm4d_oracle_preprocess_backward(const dm4d_oracle_in *in, const dm4d_oracle_state *st, dm4d_oracle_grads *g)
{
    int N = in->N, W = in->W, H = in->H;
    float fy = (float)H / (2.0f * in->tanfovy);
    float fx = (float)W / (2.0f * in->tanfovx);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) preprocess_backward_one(in, st, i, fx, fy, g);
    return 0;
}

/* markVisible: view-space z > 0.2 */
void dm4d_oracle_mark_visible(int N, const float *means3D, const float *viewmatrix, uint8_t *present)
{
    for (int i = 0; i < N; ++i) {
        float pv[3];
        xform4x3(means3D + 3 * i, viewmatrix, pv);
        present[i] = pv[2] > 0.2f;
    }
}
