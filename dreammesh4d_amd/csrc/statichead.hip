// statichead.hip -- the image-space terms of a static-stage iteration in one launch each way (gfx950).
//
// What `SuGaRStatic.training_step` (custom/threestudio-dreammesh4d/system/sugar_static.py:110-340) and the static renderer's
// epilogue (renderer/diff_sugar_rasterizer_normal.py:196-226) do with the rendered batch [B, 6 | 1 | 1, H, W] (RGB | normal, depth,
// opacity; view 0 .. = reference views, the rest random views):
//   comp_rgb    = clamp(rgb, 0, 1)
//   comp_normal = where(alpha > 0.99, n, n.detach()),  n = normalize(normal) 0.5 alpha + 0.5       (:198-207)
//   comp_depth  = where(alpha > 0.99, depth, depth.detach())
//   reference views: mse(gt_rgb m, comp_rgb m), mse(m, alpha)                                        (sugar_static.py:151-160)
//   random views:    the Zero123 guidance's input (a bilinear resize to half the size = the 2 x 2 mean), and the total-variation
//                    terms of comp_rgb / comp_depth / comp_normal (threestudio/utils/loss.py:8-16; sugar_static.py:274-290)
// As torch operators: ~130 launches over 5 x 512^2 images per iteration (normalize, three `where`s, six shifted differences with
// their squares and sums, five slice backwards that each fill and copy a full image, ...): ~1.3 ms of a 13.9 ms iteration.  Here:
//   forward : one pass; per workgroup eight partial sums (mse rgb, mse mask, h / w total variation of rgb, depth, normal map), summed
//             by the caller in a fixed order; the half-size images of the random views;
//   backward: one pass that WRITES dL/dcolor [B,6,H,W], dL/ddepth, dL/dalpha [B,1,H,W]: a pixel recomputes the three quantities at
//             its four neighbours (the total variation's gradient is a five-point stencil of them) instead of reading them back.
#include "common.h"
#include "../../include/dm4d.h"

namespace dm4d {

constexpr int kSHThreads = 256;
constexpr int kSHSums = 8;

struct SHeadArgs {
    int B, H, W;
    const float *color, *depth, *alpha;            // [B][6][H][W], [B][1][H][W], [B][1][H][W]
    const int32_t *ref_pos, *rnd_pos;              // [B]: index among the reference / random views, or -1
    const float *ref_images, *ref_masks;           // [L][H][W][3], [L][H][W][1]
    const int64_t *fidx_ref;                       // [n_ref]
    int n_ref, n_rnd;
};

__device__ __forceinline__ float sh_clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// the seven quantities the total variation runs over at pixel p of a view: clamp(rgb) (3), depth, the normal map (3)
struct SHQ { float q[7]; };
__device__ __forceinline__ SHQ sh_quantities(const float *c0, const float *dp, const float *al, int HW, int p)
{
    SHQ r;
#pragma unroll
    for (int k = 0; k < 3; ++k) r.q[k] = sh_clamp01(c0[(size_t)k * HW + p]);
    r.q[3] = dp[p];
    const float nx = c0[(size_t)3 * HW + p], ny = c0[(size_t)4 * HW + p], nz = c0[(size_t)5 * HW + p], a = al[p];
    const float len = fmaxf(sqrtf((nx * nx + ny * ny) + nz * nz), 1e-12f);         // F.normalize(dim): x / max(|x|, eps)
    r.q[4] = (nx / len) * 0.5f * a + 0.5f;
    r.q[5] = (ny / len) * 0.5f * a + 0.5f;
    r.q[6] = (nz / len) * 0.5f * a + 0.5f;
    return r;
}

// grid (pixel blocks, B); partial [B][gridDim.x][8]
__global__ __launch_bounds__(kSHThreads) void k_static_head_fwd(SHeadArgs a, float *__restrict__ partial, float *__restrict__ half_rgb)
{
    __shared__ float red[kSHThreads / 64][kSHSums];
    const int v = blockIdx.y, HW = a.H * a.W, W = a.W, H = a.H;
    const int r = a.ref_pos[v], n = a.rnd_pos[v];
    const float *c0 = a.color + (size_t)v * 6 * HW, *dp = a.depth + (size_t)v * HW, *al = a.alpha + (size_t)v * HW;
    float s[kSHSums];
#pragma unroll
    for (int k = 0; k < kSHSums; ++k) s[k] = 0.f;
    if (r >= 0 && r < a.n_ref) {
        const size_t f = (size_t)a.fidx_ref[r];
        const float *gt = a.ref_images + f * HW * 3, *gm = a.ref_masks + f * HW;
        for (int p = blockIdx.x * kSHThreads + threadIdx.x; p < HW; p += gridDim.x * kSHThreads) {
            const float m = gm[p];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float d = gt[3 * (size_t)p + k] * m - sh_clamp01(c0[(size_t)k * HW + p]) * m;
                s[0] = __builtin_fmaf(d, d, s[0]);
            }
            const float dm = m - al[p];
            s[1] = __builtin_fmaf(dm, dm, s[1]);
        }
    }
    if (n >= 0 && n < a.n_rnd) {
        for (int p = blockIdx.x * kSHThreads + threadIdx.x; p < HW; p += gridDim.x * kSHThreads) {
            const int y = p / W, x = p - y * W;
            const SHQ q0 = sh_quantities(c0, dp, al, HW, p);
            if (y + 1 < H) {
                const SHQ q1 = sh_quantities(c0, dp, al, HW, p + W);
#pragma unroll
                for (int k = 0; k < 7; ++k) { const float d = q1.q[k] - q0.q[k]; const int t = k < 3 ? 2 : k == 3 ? 4 : 6; s[t] = __builtin_fmaf(d, d, s[t]); }
            }
            if (x + 1 < W) {
                const SHQ q1 = sh_quantities(c0, dp, al, HW, p + 1);
#pragma unroll
                for (int k = 0; k < 7; ++k) { const float d = q1.q[k] - q0.q[k]; const int t = k < 3 ? 3 : k == 3 ? 5 : 7; s[t] = __builtin_fmaf(d, d, s[t]); }
            }
        }
        if (half_rgb) {
            const int Wh = W >> 1, HWh = (H >> 1) * Wh;
            float *o = half_rgb + (size_t)n * HWh * 3;
            for (int p = blockIdx.x * kSHThreads + threadIdx.x; p < HWh; p += gridDim.x * kSHThreads) {
                const int oy = p / Wh, ox = p - oy * Wh;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float *q = c0 + (size_t)k * HW + (size_t)(2 * oy) * W + 2 * ox;
                    const float h0 = 0.5f * sh_clamp01(q[0]) + 0.5f * sh_clamp01(q[1]), h1 = 0.5f * sh_clamp01(q[W]) + 0.5f * sh_clamp01(q[W + 1]);
                    o[3 * (size_t)p + k] = 0.5f * h0 + 0.5f * h1;
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kSHSums; ++k) {
        const float t = wave_sum_row3(s[k]);
        if (lane == 63) red[wv][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < kSHSums) {
        float t = 0.f;
        for (int w = 0; w < kSHThreads / 64; ++w) t += red[w][threadIdx.x];
        partial[((size_t)v * gridDim.x + blockIdx.x) * kSHSums + threadIdx.x] = t;
    }
}

// g: upstream gradients of (mse_rgb, mse_mask, tv_rgb, tv_depth, tv_normal) (device, 5 floats); g_half [n_rnd][H/2][W/2][3] or nullptr
__global__ __launch_bounds__(kSHThreads) void k_static_head_bwd(SHeadArgs a, const float *__restrict__ g, const float *__restrict__ g_half,
                                                                float *__restrict__ g_color, float *__restrict__ g_depth, float *__restrict__ g_alpha)
{
    const int v = blockIdx.y, HW = a.H * a.W, W = a.W, H = a.H;
    const int r = a.ref_pos[v], n = a.rnd_pos[v];
    const float *c0 = a.color + (size_t)v * 6 * HW, *dp = a.depth + (size_t)v * HW, *al = a.alpha + (size_t)v * HW;
    float *gc = g_color + (size_t)v * 6 * HW, *gd = g_depth + (size_t)v * HW, *ga = g_alpha + (size_t)v * HW;
    const bool is_ref = r >= 0 && r < a.n_ref, is_rnd = n >= 0 && n < a.n_rnd;
    const float k_rgb = is_ref ? g[0] * 2.0f / ((float)a.n_ref * (float)HW * 3.0f) : 0.f;
    const float k_mask = is_ref ? g[1] * 2.0f / ((float)a.n_ref * (float)HW) : 0.f;
    // d tv / d (sum of squared differences): tv = 2 (h_tv / (c (H - 1) W) + w_tv / (c H (W - 1))) / b
    const float bn = (float)a.n_rnd;
    float ch[3], cw[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const float c = t == 1 ? 1.0f : 3.0f;
        ch[t] = is_rnd ? g[2 + t] * 2.0f / (c * (float)(H - 1) * (float)W * bn) : 0.f;
        cw[t] = is_rnd ? g[2 + t] * 2.0f / (c * (float)H * (float)(W - 1) * bn) : 0.f;
    }
    const float *gt = nullptr, *gm = nullptr;
    if (is_ref) {
        const size_t f = (size_t)a.fidx_ref[r];
        gt = a.ref_images + f * HW * 3;
        gm = a.ref_masks + f * HW;
    }
    const int Wh = W >> 1;
    const float *gh = (is_rnd && g_half) ? g_half + (size_t)n * (H >> 1) * Wh * 3 : nullptr;
    for (int p = blockIdx.x * kSHThreads + threadIdx.x; p < HW; p += gridDim.x * kSHThreads) {
        const int y = p / W, x = p - y * W;
        float dq[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) dq[k] = 0.f;
        if (is_rnd) {
            const SHQ q0 = sh_quantities(c0, dp, al, HW, p);
            // gradient of sum (q[y+1] - q[y])^2 + (q[x+1] - q[x])^2 with respect to q at this pixel
            if (y > 0) { const SHQ q1 = sh_quantities(c0, dp, al, HW, p - W);
#pragma unroll
                for (int k = 0; k < 7; ++k) dq[k] += ch[k < 3 ? 0 : k == 3 ? 1 : 2] * (2.0f * (q0.q[k] - q1.q[k])); }
            if (y + 1 < H) { const SHQ q1 = sh_quantities(c0, dp, al, HW, p + W);
#pragma unroll
                for (int k = 0; k < 7; ++k) dq[k] -= ch[k < 3 ? 0 : k == 3 ? 1 : 2] * (2.0f * (q1.q[k] - q0.q[k])); }
            if (x > 0) { const SHQ q1 = sh_quantities(c0, dp, al, HW, p - 1);
#pragma unroll
                for (int k = 0; k < 7; ++k) dq[k] += cw[k < 3 ? 0 : k == 3 ? 1 : 2] * (2.0f * (q0.q[k] - q1.q[k])); }
            if (x + 1 < W) { const SHQ q1 = sh_quantities(c0, dp, al, HW, p + 1);
#pragma unroll
                for (int k = 0; k < 7; ++k) dq[k] -= cw[k < 3 ? 0 : k == 3 ? 1 : 2] * (2.0f * (q1.q[k] - q0.q[k])); }
        }
        const float alpha = al[p];
        const bool solid = alpha > 0.99f;              // the renderer detaches depth and the normal map elsewhere
        float d_alpha = 0.f;
        // rgb
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float c = c0[(size_t)k * HW + p];
            float gk = dq[k];
            if (gt) { const float m = gm[p]; gk += k_rgb * ((sh_clamp01(c) * m - gt[3 * (size_t)p + k] * m) * m); }
            if (gh) gk += 0.25f * gh[3 * ((size_t)(y >> 1) * Wh + (x >> 1)) + k];
            gc[(size_t)k * HW + p] = (c >= 0.0f && c <= 1.0f) ? gk : 0.f;      // torch.clamp passes the gradient on [min, max]
        }
        // depth
        gd[p] = solid ? dq[3] : 0.f;
        // normal map = normalize(v) 0.5 alpha + 0.5
        float gv[3] = {0.f, 0.f, 0.f};
        if (solid && is_rnd) {
            const float vx = c0[(size_t)3 * HW + p], vy = c0[(size_t)4 * HW + p], vz = c0[(size_t)5 * HW + p];
            const float nrm = sqrtf((vx * vx + vy * vy) + vz * vz), len = fmaxf(nrm, 1e-12f);
            const float nh[3] = {vx / len, vy / len, vz / len};
            float gn[3], dotp = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                gn[k] = dq[4 + k] * (0.5f * alpha);
                d_alpha += dq[4 + k] * (nh[k] * 0.5f);
                dotp += nh[k] * gn[k];
            }
            if (nrm > 1e-12f) {
#pragma unroll
                for (int k = 0; k < 3; ++k) gv[k] = (gn[k] - nh[k] * dotp) / len;
            } else {                                    // (the clamp of the norm is active: x / eps)
#pragma unroll
                for (int k = 0; k < 3; ++k) gv[k] = gn[k] / len;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) gc[(size_t)(3 + k) * HW + p] = gv[k];
        if (gm) d_alpha += k_mask * (alpha - gm[p]);
        ga[p] = d_alpha;
    }
}

}  // namespace dm4d

using namespace dm4d;

extern "C" {

static int shead_check(int B, int H, int W, const void *color, const void *depth, const void *alpha, const void *ref_pos, const void *rnd_pos,
                       int n_ref, int n_rnd, const void *ref_images, const void *ref_masks, const void *fidx_ref)
{
    if (B < 0 || H < 2 || W < 2 || n_ref < 0 || n_rnd < 0) { set_error("static head: bad shape (B %d, %d x %d)", B, H, W); return DM4D_ERR_INVALID; }
    if ((H | W) & 1) { set_error("static head: H and W must be even (the guidance's resize to half the size is a 2 x 2 mean), got %d x %d", H, W); return DM4D_ERR_UNSUPPORTED; }
    if (B == 0) return DM4D_OK;
    if (!color || !depth || !alpha || !ref_pos || !rnd_pos) { set_error("static head: null tensor"); return DM4D_ERR_INVALID; }
    if (n_ref > 0 && (!ref_images || !ref_masks || !fidx_ref)) { set_error("static head: reference views without reference images"); return DM4D_ERR_INVALID; }
    return DM4D_OK;
}

int32_t dm4d_static_head_blocks(int32_t H, int32_t W)
{
    const int64_t px = (int64_t)H * W;
    const int64_t b = (px + 4 * kSHThreads - 1) / (4 * kSHThreads);
    return (int32_t)(b < 1 ? 1 : b > 256 ? 256 : b);
}

int dm4d_static_head_forward(int32_t B, int32_t H, int32_t W, const float *color, const float *depth, const float *alpha, const int32_t *ref_pos,
                             const int32_t *rnd_pos, const float *ref_images, const float *ref_masks, const int64_t *fidx_ref, int32_t n_ref,
                             int32_t n_rnd, float *partial, float *half_rgb, dm4d_stream_t stream)
{
    int rc = shead_check(B, H, W, color, depth, alpha, ref_pos, rnd_pos, n_ref, n_rnd, ref_images, ref_masks, fidx_ref);
    if (rc != DM4D_OK || B == 0) return rc;
    if (!partial) { set_error("static head: null output"); return DM4D_ERR_INVALID; }
    SHeadArgs a{B, H, W, color, depth, alpha, ref_pos, rnd_pos, ref_images, ref_masks, fidx_ref, n_ref, n_rnd};
    hipLaunchKernelGGL(k_static_head_fwd, dim3(dm4d_static_head_blocks(H, W), B), dim3(kSHThreads), 0, (hipStream_t)stream, a, partial, half_rgb);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_static_head_backward(int32_t B, int32_t H, int32_t W, const float *color, const float *depth, const float *alpha, const int32_t *ref_pos,
                              const int32_t *rnd_pos, const float *ref_images, const float *ref_masks, const int64_t *fidx_ref, int32_t n_ref,
                              int32_t n_rnd, const float *g_terms, const float *g_half, float *g_color, float *g_depth, float *g_alpha,
                              dm4d_stream_t stream)
{
    int rc = shead_check(B, H, W, color, depth, alpha, ref_pos, rnd_pos, n_ref, n_rnd, ref_images, ref_masks, fidx_ref);
    if (rc != DM4D_OK || B == 0) return rc;
    if (!g_terms || !g_color || !g_depth || !g_alpha) { set_error("static head: null tensor in backward"); return DM4D_ERR_INVALID; }
    SHeadArgs a{B, H, W, color, depth, alpha, ref_pos, rnd_pos, ref_images, ref_masks, fidx_ref, n_ref, n_rnd};
    hipLaunchKernelGGL(k_static_head_bwd, dim3(dm4d_static_head_blocks(H, W), B), dim3(kSHThreads), 0, (hipStream_t)stream, a, g_terms, g_half, g_color,
                       g_depth, g_alpha);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // extern "C"
