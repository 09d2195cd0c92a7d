// imagehead.hip -- the image-space head of a dynamic-stage iteration in two launches each way (gfx950).
//
// What the reference's training step does with the rendered batch (custom/threestudio-dreammesh4d/system/sugar_4dgen.py:148-190,
// 397-429): comp_rgb = clamp(render, 0, 1); on the REFERENCE views loss_rgb = mse(gt_rgb, comp_rgb) and loss_mask = mse(gt_mask,
// opacity); the RANDOM views go to the Zero123 guidance, which first resizes them to 256 x 256 with bilinear interpolation
// (guidance/...zero123...py:299-310) -- at exactly half the size that is the mean of each 2 x 2 block.  As torch operators this is ~45
// elementwise / index / reduction launches over 25 MB tensors per iteration (clamp and its mask, three gathers and their scatter
// backward, two mse with their reductions, the resize, the gradient adds): 0.4 ms of a 13.3 ms iteration.  Here:
//   forward:  one pass over the batch's pixels: squared-error partial sums of the reference views (one pair per workgroup, summed by
//             the caller in a fixed order: deterministic), the half-size clamped images of the random views;
//   backward: one pass that WRITES dL/dcolor [B, C, H, W] (the clamp's pass-through mask applied; channels >= 3 and views with no
//             loss get zeros) and dL/dalpha [B, 1, H, W] -- no zero fill, no accumulation.
#include "common.h"
#include "../../include/dm4d.h"

namespace dm4d {

constexpr int kHeadThreads = 256;

struct HeadArgs {
    int B, H, W, C;
    const float *color, *alpha;                    // [B][C][H][W], [B][1][H][W]
    const int32_t *ref_pos, *rnd_pos;              // [B]: index among the reference / random views, or -1
    const float *ref_images, *ref_masks;           // [L][H][W][3], [L][H][W][1]
    const int64_t *fidx_ref;                       // [n_ref]: frame of each reference view
    int n_ref, n_rnd;
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// grid: (pixel blocks, B).  partial: [B][gridDim.x][2] (rgb, mask) -- zeros for views that are not reference views
__global__ __launch_bounds__(kHeadThreads) void k_head_fwd(HeadArgs a, float *__restrict__ partial, float *__restrict__ half_rgb)
{
    __shared__ float red[kHeadThreads / 64][2];
    const int v = blockIdx.y, HW = a.H * a.W;
    const int r = a.ref_pos[v], n = a.rnd_pos[v];
    const float *c0 = a.color + (size_t)v * a.C * HW;
    float s_rgb = 0.f, s_mask = 0.f;
    if (r >= 0 && r < a.n_ref) {
        const size_t f = (size_t)a.fidx_ref[r];
        const float *gt = a.ref_images + f * HW * 3, *gm = a.ref_masks + f * HW, *al = a.alpha + (size_t)v * HW;
        for (int p = blockIdx.x * kHeadThreads + threadIdx.x; p < HW; p += gridDim.x * kHeadThreads) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float d = gt[3 * (size_t)p + k] - clamp01(c0[(size_t)k * HW + p]);
                s_rgb = __builtin_fmaf(d, d, s_rgb);
            }
            const float dm = al[p] - gm[p];
            s_mask = __builtin_fmaf(dm, dm, s_mask);
        }
    }
    if (n >= 0 && n < a.n_rnd && half_rgb) {          // (n_rnd = 0: the caller has no use for the random views)
        const int Wh = a.W >> 1, HWh = (a.H >> 1) * Wh;
        float *o = half_rgb + (size_t)n * HWh * 3;
        for (int p = blockIdx.x * kHeadThreads + threadIdx.x; p < HWh; p += gridDim.x * kHeadThreads) {
            const int oy = p / Wh, ox = p - oy * Wh;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float *q = c0 + (size_t)k * HW + (size_t)(2 * oy) * a.W + 2 * ox;
                // upsample_bilinear2d(align_corners=False) at scale 1/2: lambda = 0.5 on both axes, in the library's order
                const float h0 = 0.5f * clamp01(q[0]) + 0.5f * clamp01(q[1]), h1 = 0.5f * clamp01(q[a.W]) + 0.5f * clamp01(q[a.W + 1]);
                o[3 * (size_t)p + k] = 0.5f * h0 + 0.5f * h1;
            }
        }
    }
    s_rgb = wave_sum_row3(s_rgb);                  // (the wave's total, valid in lanes 48 .. 63)
    s_mask = wave_sum_row3(s_mask);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 63) { red[wv][0] = s_rgb; red[wv][1] = s_mask; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float x = 0.f, y = 0.f;
        for (int w = 0; w < kHeadThreads / 64; ++w) { x += red[w][0]; y += red[w][1]; }
        float *o = partial + ((size_t)v * gridDim.x + blockIdx.x) * 2;
        o[0] = x; o[1] = y;
    }
}

// dL/dcolor, dL/dalpha of  w_rgb mse_rgb + w_mask mse_mask + <g_half, half_rgb>;  g_rgb / g_mask: the upstream gradients of the two
// means (device scalars), g_half [n_rnd][H/2][W/2][3] or nullptr
__global__ __launch_bounds__(kHeadThreads) void k_head_bwd(HeadArgs a, const float *__restrict__ g_rgb, const float *__restrict__ g_mask,
                                                           const float *__restrict__ g_half, float *__restrict__ g_color, float *__restrict__ g_alpha)
{
    const int v = blockIdx.y, HW = a.H * a.W;
    const int r = a.ref_pos[v], n = a.rnd_pos[v];
    const float *c0 = a.color + (size_t)v * a.C * HW;
    float *gc = g_color + (size_t)v * a.C * HW, *ga = g_alpha + (size_t)v * HW;
    const float k_rgb = (r >= 0 && r < a.n_ref && g_rgb) ? g_rgb[0] * 2.0f / ((float)a.n_ref * (float)HW * 3.0f) : 0.f;
    const float k_mask = (r >= 0 && r < a.n_ref && g_mask) ? g_mask[0] * 2.0f / ((float)a.n_ref * (float)HW) : 0.f;
    const float *gt = nullptr, *gm = nullptr, *al = a.alpha + (size_t)v * HW;
    if (r >= 0 && r < a.n_ref) {
        const size_t f = (size_t)a.fidx_ref[r];
        gt = a.ref_images + f * HW * 3;
        gm = a.ref_masks + f * HW;
    }
    const int Wh = a.W >> 1;
    const float *gh = (n >= 0 && n < a.n_rnd && g_half) ? g_half + (size_t)n * (a.H >> 1) * Wh * 3 : nullptr;
    for (int p = blockIdx.x * kHeadThreads + threadIdx.x; p < HW; p += gridDim.x * kHeadThreads) {
        const int y = p / a.W, x = p - y * a.W;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float c = c0[(size_t)k * HW + p];
            float g = 0.f;
            if (gt) g = k_rgb * (clamp01(c) - gt[3 * (size_t)p + k]);
            if (gh) g += 0.25f * gh[3 * ((size_t)(y >> 1) * Wh + (x >> 1)) + k];
            gc[(size_t)k * HW + p] = (c >= 0.0f && c <= 1.0f) ? g : 0.f;              // torch.clamp passes the gradient on [min, max], bounds included
        }
        for (int k = 3; k < a.C; ++k) gc[(size_t)k * HW + p] = 0.f;
        ga[p] = gm ? k_mask * (al[p] - gm[p]) : 0.f;
    }
}

}  // namespace dm4d

using namespace dm4d;

// ---------------------------------------------------------------------------------------- the scalar arithmetic around the heads
// What the systems do with a head's partial sums and with the loss terms (`loss = 0.0 + lambda_a * a + lambda_b * b + ...`,
// system/sugar_4dgen.py:296-330, sugar_static.py:246-340) is a dozen 1-element torch operators forward and as many backward:
// 5 us of launch each for one multiply.  Three one-workgroup kernels instead.
constexpr int kGlueMax = 16;
struct SumArgs { int64_t n; int k, m; float mat[8 * 8]; };
// out[j] = sum_c mat[c][j] * (sum_i partial[i * k + c]); one workgroup, fixed order (deterministic)
__global__ __launch_bounds__(256) void k_partial_sums(SumArgs a, const float *__restrict__ partial, float *__restrict__ out)
{
    __shared__ float s[8][256];
    const int tid = threadIdx.x;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    for (int64_t i = tid; i < a.n; i += 256)
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < a.k) acc[c] += partial[i * a.k + c];
#pragma unroll
    for (int c = 0; c < 8; ++c) s[c][tid] = acc[c];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w)
#pragma unroll
            for (int c = 0; c < 8; ++c) s[c][tid] += s[c][tid + w];
        __syncthreads();
    }
    if (tid < a.m) {
        float o = 0.f;
        for (int c = 0; c < a.k; ++c) o += a.mat[c * a.m + tid] * s[c][0];
        out[tid] = o;
    }
}
struct WsumArgs { int n; const float *t[kGlueMax]; float w[kGlueMax]; };
__global__ void k_weighted_sum(WsumArgs a, float *__restrict__ out)
{
    float acc = 0.0f;
    for (int i = 0; i < a.n; ++i) acc = acc + a.w[i] * a.t[i][0];      // (the float32 roundings of the torch expression, left to right)
    out[0] = acc;
}
__global__ void k_weighted_sum_bwd(WsumArgs a, const float *__restrict__ g, float *__restrict__ out)
{
    const int i = threadIdx.x;
    if (i < a.n) out[i] = g[0] * a.w[i];
}

extern "C" {

static int head_check(int B, int H, int W, int C, const void *color, const void *alpha, const void *ref_pos, const void *rnd_pos, int n_ref,
                      const void *ref_images, const void *ref_masks, const void *fidx_ref)
{
    if (B < 0 || H <= 0 || W <= 0 || C < 3 || n_ref < 0) { set_error("image head: bad shape (B %d, %d x %d, C %d)", B, H, W, C); return DM4D_ERR_INVALID; }
    if ((H | W) & 1) { set_error("image head: H and W must be even (the guidance's resize to half the size is a 2 x 2 mean), got %d x %d", H, W); return DM4D_ERR_UNSUPPORTED; }
    if (B == 0) return DM4D_OK;
    if (!color || !alpha || !ref_pos || !rnd_pos) { set_error("image head: null tensor"); return DM4D_ERR_INVALID; }
    if (n_ref > 0 && (!ref_images || !ref_masks || !fidx_ref)) { set_error("image head: reference views without reference images"); return DM4D_ERR_INVALID; }
    return DM4D_OK;
}

int32_t dm4d_image_head_blocks(int32_t H, int32_t W)
{
    const int64_t px = (int64_t)H * W;
    const int64_t b = (px + 4 * kHeadThreads - 1) / (4 * kHeadThreads);
    return (int32_t)(b < 1 ? 1 : b > 256 ? 256 : b);
}

int dm4d_image_head_forward(int32_t B, int32_t H, int32_t W, int32_t C, const float *color, const float *alpha, const int32_t *ref_pos,
                            const int32_t *rnd_pos, const float *ref_images, const float *ref_masks, const int64_t *fidx_ref, int32_t n_ref,
                            int32_t n_rnd, float *partial, float *half_rgb, dm4d_stream_t stream)
{
    int rc = head_check(B, H, W, C, color, alpha, ref_pos, rnd_pos, n_ref, ref_images, ref_masks, fidx_ref);
    if (rc != DM4D_OK || B == 0) return rc;
    if (!partial || (n_rnd > 0 && !half_rgb)) { set_error("image head: null output"); return DM4D_ERR_INVALID; }
    HeadArgs a{B, H, W, C, color, alpha, ref_pos, rnd_pos, ref_images, ref_masks, fidx_ref, n_ref, n_rnd};
    hipLaunchKernelGGL(k_head_fwd, dim3(dm4d_image_head_blocks(H, W), B), dim3(kHeadThreads), 0, (hipStream_t)stream, a, partial, half_rgb);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_image_head_backward(int32_t B, int32_t H, int32_t W, int32_t C, const float *color, const float *alpha, const int32_t *ref_pos,
                             const int32_t *rnd_pos, const float *ref_images, const float *ref_masks, const int64_t *fidx_ref, int32_t n_ref,
                             int32_t n_rnd, const float *g_rgb, const float *g_mask, const float *g_half, float *g_color, float *g_alpha,
                             dm4d_stream_t stream)
{
    int rc = head_check(B, H, W, C, color, alpha, ref_pos, rnd_pos, n_ref, ref_images, ref_masks, fidx_ref);
    if (rc != DM4D_OK || B == 0) return rc;
    if (!g_color || !g_alpha) { set_error("image head: null output"); return DM4D_ERR_INVALID; }
    HeadArgs a{B, H, W, C, color, alpha, ref_pos, rnd_pos, ref_images, ref_masks, fidx_ref, n_ref, n_rnd};
    hipLaunchKernelGGL(k_head_bwd, dim3(dm4d_image_head_blocks(H, W), B), dim3(kHeadThreads), 0, (hipStream_t)stream, a, g_rgb, g_mask, g_half, g_color,
                       g_alpha);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_partial_sums(int64_t n, int32_t k, int32_t m, const float *partial, const float *matrix, float *out, dm4d_stream_t stream)
{
    if (n < 0 || k < 1 || k > 8 || m < 1 || m > 8) { set_error("partial sums: n >= 0, 1 <= k, m <= 8 (got %lld, %d, %d)", (long long)n, k, m); return DM4D_ERR_INVALID; }
    if ((n > 0 && !partial) || !matrix || !out) { set_error("partial sums: null pointer"); return DM4D_ERR_INVALID; }
    SumArgs a;
    a.n = n; a.k = k; a.m = m;
    for (int i = 0; i < k * m; ++i) a.mat[i] = matrix[i];
    hipLaunchKernelGGL(k_partial_sums, dim3(1), dim3(256), 0, (hipStream_t)stream, a, partial, out);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

static int wsum_args(WsumArgs &a, int32_t n, const float *const *terms, const float *weights)
{
    if (n < 1 || n > kGlueMax) { set_error("weighted sum: 1 <= n <= %d terms (got %d)", kGlueMax, n); return DM4D_ERR_INVALID; }
    if (!weights) { set_error("weighted sum: null weights"); return DM4D_ERR_INVALID; }
    a.n = n;
    for (int i = 0; i < n; ++i) {
        a.t[i] = terms ? terms[i] : nullptr;
        a.w[i] = weights[i];
    }
    return DM4D_OK;
}

int dm4d_weighted_sum(int32_t n, const float *const *terms, const float *weights, float *out, dm4d_stream_t stream)
{
    WsumArgs a;
    int rc = wsum_args(a, n, terms, weights);
    if (rc != DM4D_OK) return rc;
    if (!terms || !out) { set_error("weighted sum: null pointer"); return DM4D_ERR_INVALID; }
    for (int i = 0; i < n; ++i)
        if (!terms[i]) { set_error("weighted sum: term %d is null", i); return DM4D_ERR_INVALID; }
    hipLaunchKernelGGL(k_weighted_sum, dim3(1), dim3(1), 0, (hipStream_t)stream, a, out);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_weighted_sum_backward(int32_t n, const float *g, const float *weights, float *out, dm4d_stream_t stream)
{
    WsumArgs a;
    int rc = wsum_args(a, n, nullptr, weights);
    if (rc != DM4D_OK) return rc;
    if (!g || !out) { set_error("weighted sum: null pointer"); return DM4D_ERR_INVALID; }
    hipLaunchKernelGGL(k_weighted_sum_bwd, dim3(1), dim3(64), 0, (hipStream_t)stream, a, g, out);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // extern "C"
