// sds_glue.hip -- the arithmetic BETWEEN the networks of a Zero123 SDS step as two launches (gfx950).
//
// Where it sits (custom/threestudio-dreammesh4d/guidance/temporal_stable_zero123_guidance.py:299-374):
//
//   moments = quant_conv(encoder(images))                                   VAE encoder (csrc/conv_mfma.hip, groupnorm.hip ...)
//   ---- dm4d_sds_prepare ------------------------------------------------------------------------------------------------
//   mean, logvar = moments.chunk(2, 1); logvar.clamp(-30, 20)               extern/ldm_zero123/modules/distributions/distributions.py:24-41
//   latents = scale_factor (mean + exp(0.5 logvar) posterior_noise)         ddpm.py get_first_stage_encoding
//   noisy   = sqrt(ac[t]) latents + sqrt(1 - ac[t]) noise                   DDIMScheduler.add_noise
//   x_in    = cat([noisy] * 2) ++ cat([0, c_concat[frame]])                 guidance :330-344 (classifier-free pair, hybrid conditioning)
//   ---- the UNet -------------------------------------------------------------------------------------------------------
//   ---- dm4d_sds_finish -------------------------------------------------------------------------------------------------
//   pred    = uncond + guidance_scale (cond - uncond)                       :346-349
//   grad    = nan_to_num((1 - ac[t]) (pred - noise)), clipped               :351-357
//   loss    = 0.5 mse(latents, latents - grad, "sum") / B;  |grad|          :359-366
//   dL/dmoments (for an upstream gradient of 1): the chain of the first block backwards
//   ---- encoder backward -------------------------------------------------------------------------------------------------
//
// As torch operators the two blocks are ~70 launches of 16 K-element kernels (each at the ~5 us launch floor: 0.35 ms of a 9.8 ms
// step).  The kernels below evaluate the SAME expressions with the SAME roundings (every float16 operator of the torch graph rounds
// its float32 result to float16: so does each step here; no contraction: -ffp-contract=off): at full size the loss of a seeded
// step is the torch graph's to the printed digit, the image gradient agrees to ~2e-5 of its range (tests/test_zero123_graphs_gpu.py);
// the loss and the norm are sums in another order (float32, 16 K terms).
// Tensors come with their strides (the moments and the prediction are whatever layout the networks' last layers produce).
#include "common.h"
#include "dm4d.h"

namespace dm4d {

struct SdsStrides { int64_t b, c, h, w; };
__device__ __forceinline__ int64_t sds_off(const SdsStrides &s, int b, int c, int y, int x) { return b * s.b + c * s.c + y * s.h + x * s.w; }

struct SdsArgs {
    int B, H, W;
    float scale_factor, guidance_scale;
    const _Float16 *moments;  SdsStrides sm;      // [B, 8, H, W]
    const _Float16 *post;     SdsStrides sp;      // [B, 4, H, W]  posterior noise
    const float *noise;       SdsStrides sn;      // [B, 4, H, W]
    float *latents;           SdsStrides sl;      // [B, 4, H, W]
    const int64_t *t;                             // [B]
    const float *alphas;                          // [T]
    const _Float16 *c_concat; SdsStrides sc;      // [L, 4, H, W]
    const int64_t *fidx;                          // [B]
    _Float16 *x_in;           SdsStrides sx;      // [2 B, 8, H, W]
    int64_t *t2;                                  // [2 B]
    const _Float16 *pred;     SdsStrides sq;      // [2 B, 4, H, W]
    const float *clip;                            // scalar or NULL
    _Float16 *d_moments;      SdsStrides sd;      // [B, 8, H, W]
    float *loss, *grad_norm;                      // scalars
};

typedef _Float16 h16;
__device__ __forceinline__ h16 hmulf(h16 a, float b) { return (h16)((float)a * b); }

// std = exp(0.5 clamp(logvar)) and the clamp's gradient mask, rounded where the torch graph rounds
__device__ __forceinline__ h16 sds_std(h16 logvar, bool &inside)
{
    const float lv = (float)logvar;
    inside = lv >= -30.0f && lv <= 20.0f;                        // clamp backward: the gradient passes where min <= x <= max
    const h16 c = (h16)fminf(fmaxf(lv, -30.0f), 20.0f);
    const h16 half_lv = (h16)(0.5f * (float)c);
    return (h16)expf((float)half_lv);
}

__global__ __launch_bounds__(256) void k_sds_prepare(SdsArgs a)
{
    const int HW = a.H * a.W, n = a.B * HW;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 2 * a.B) a.t2[i] = a.t[i % a.B];
    if (i >= n) return;
    const int b = i / HW, y = (i % HW) / a.W, x = i % a.W;
    const float ac = a.alphas[a.t[b]];
    const float sa = sqrtf(ac), sb = sqrtf(1.0f - ac);
    const int64_t f = a.fidx[b];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const h16 mean = a.moments[sds_off(a.sm, b, c, y, x)];
        bool inside;
        const h16 sd = sds_std(a.moments[sds_off(a.sm, b, 4 + c, y, x)], inside);
        const h16 e = (h16)((float)sd * (float)a.post[sds_off(a.sp, b, c, y, x)]);
        const h16 s = (h16)((float)mean + (float)e);
        const float lat = (float)(h16)(a.scale_factor * (float)s);
        a.latents[sds_off(a.sl, b, c, y, x)] = lat;
        const float noisy = sa * lat + sb * a.noise[sds_off(a.sn, b, c, y, x)];
        const h16 nh = (h16)noisy;
        a.x_in[sds_off(a.sx, b, c, y, x)] = nh;
        a.x_in[sds_off(a.sx, a.B + b, c, y, x)] = nh;
        a.x_in[sds_off(a.sx, b, 4 + c, y, x)] = (h16)0.0f;
        a.x_in[sds_off(a.sx, a.B + b, 4 + c, y, x)] = a.c_concat[sds_off(a.sc, (int)f, c, y, x)];
    }
}

// ONE workgroup (the two sums in a fixed order: reproducible)
__global__ __launch_bounds__(1024) void k_sds_finish(SdsArgs a)
{
    __shared__ float s_l[16], s_g[16];
    const int HW = a.H * a.W, n = a.B * HW;
    const float cmul = (1.0f / (float)a.B) * 0.5f;              // d(0.5 mse / B) / d mse
    const float clipv = a.clip ? *a.clip : 0.f;
    float sum_l = 0.f, sum_g = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const int b = i / HW, y = (i % HW) / a.W, x = i % a.W;
        const float ac = a.alphas[a.t[b]];
        const float w = 1.0f - ac;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float unc = (float)a.pred[sds_off(a.sq, b, c, y, x)], cnd = (float)a.pred[sds_off(a.sq, a.B + b, c, y, x)];
            const float p = unc + a.guidance_scale * (cnd - unc);
            float g = w * (p - a.noise[sds_off(a.sn, b, c, y, x)]);
            if (g != g) g = 0.f;                                                     // nan_to_num
            else if (g == __builtin_inff()) g = 3.4028234663852886e38f;
            else if (g == -__builtin_inff()) g = -3.4028234663852886e38f;
            if (a.clip) g = fminf(fmaxf(g, -clipv), clipv);
            const float lat = a.latents[sds_off(a.sl, b, c, y, x)];
            const float target = lat - g;
            const float diff = lat - target;
            sum_l = __builtin_fmaf(diff, diff, sum_l);
            sum_g = __builtin_fmaf(g, g, sum_g);
            // backward of the first block for an upstream gradient of 1 (float16 where the torch graph is float16)
            const float d_lat = (2.0f * diff) * cmul;
            const h16 dh = (h16)d_lat;
            const h16 d_sum = hmulf(dh, a.scale_factor);
            bool inside;
            const h16 sd = sds_std(a.moments[sds_off(a.sm, b, 4 + c, y, x)], inside);
            const h16 d_std = (h16)((float)d_sum * (float)a.post[sds_off(a.sp, b, c, y, x)]);
            const h16 d_half = (h16)((float)d_std * (float)sd);                      // exp backward: grad * result
            const h16 d_lv = hmulf(d_half, 0.5f);
            a.d_moments[sds_off(a.sd, b, c, y, x)] = d_sum;
            a.d_moments[sds_off(a.sd, b, 4 + c, y, x)] = inside ? d_lv : (h16)0.0f;
        }
    }
    for (int m = 32; m >= 1; m >>= 1) { sum_l += __shfl_xor(sum_l, m); sum_g += __shfl_xor(sum_g, m); }
    if ((threadIdx.x & 63) == 0) { s_l[threadIdx.x >> 6] = sum_l; s_g[threadIdx.x >> 6] = sum_g; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float l = 0.f, g = 0.f;
        for (int k = 0; k < 16; ++k) { l += s_l[k]; g += s_g[k]; }
        *a.loss = (0.5f * l) / (float)a.B;
        *a.grad_norm = sqrtf(g);
    }
}

static void sds_strides(const int64_t *s, SdsStrides &o) { o.b = s[0]; o.c = s[1]; o.h = s[2]; o.w = s[3]; }

}  // namespace dm4d

using namespace dm4d;

extern "C" int dm4d_sds_prepare(int32_t B, int32_t H, int32_t W, float scale_factor, const void *moments, const int64_t *moments_strides,
                                const void *post, const int64_t *post_strides, const float *noise, const int64_t *noise_strides,
                                float *latents, const int64_t *latents_strides, const int64_t *t, const float *alphas_cumprod,
                                const void *c_concat, const int64_t *c_concat_strides, const int64_t *frame_index, void *x_in,
                                const int64_t *x_in_strides, int64_t *t2, dm4d_stream_t stream)
{
    if (B <= 0 || H <= 0 || W <= 0) { set_error("sds_prepare: B %d H %d W %d", B, H, W); return DM4D_ERR_INVALID; }
    if (!moments || !post || !noise || !latents || !t || !alphas_cumprod || !c_concat || !frame_index || !x_in || !t2 || !moments_strides ||
        !post_strides || !noise_strides || !latents_strides || !c_concat_strides || !x_in_strides) {
        set_error("sds_prepare: null pointer");
        return DM4D_ERR_INVALID;
    }
    SdsArgs a{};
    a.B = B; a.H = H; a.W = W; a.scale_factor = scale_factor;
    a.moments = (const h16 *)moments; sds_strides(moments_strides, a.sm);
    a.post = (const h16 *)post; sds_strides(post_strides, a.sp);
    a.noise = noise; sds_strides(noise_strides, a.sn);
    a.latents = latents; sds_strides(latents_strides, a.sl);
    a.t = t; a.alphas = alphas_cumprod;
    a.c_concat = (const h16 *)c_concat; sds_strides(c_concat_strides, a.sc);
    a.fidx = frame_index;
    a.x_in = (h16 *)x_in; sds_strides(x_in_strides, a.sx);
    a.t2 = t2;
    const int n = B * H * W;
    hipLaunchKernelGGL(k_sds_prepare, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

extern "C" int dm4d_sds_finish(int32_t B, int32_t H, int32_t W, float scale_factor, float guidance_scale, const void *pred,
                               const int64_t *pred_strides, const float *latents, const int64_t *latents_strides, const float *noise,
                               const int64_t *noise_strides, const int64_t *t, const float *alphas_cumprod, const float *clip,
                               const void *moments, const int64_t *moments_strides, const void *post, const int64_t *post_strides,
                               void *d_moments, const int64_t *d_moments_strides, float *loss, float *grad_norm, dm4d_stream_t stream)
{
    if (B <= 0 || H <= 0 || W <= 0) { set_error("sds_finish: B %d H %d W %d", B, H, W); return DM4D_ERR_INVALID; }
    if (!pred || !latents || !noise || !t || !alphas_cumprod || !moments || !post || !d_moments || !loss || !grad_norm || !pred_strides ||
        !latents_strides || !noise_strides || !moments_strides || !post_strides || !d_moments_strides) {
        set_error("sds_finish: null pointer");
        return DM4D_ERR_INVALID;
    }
    SdsArgs a{};
    a.B = B; a.H = H; a.W = W; a.scale_factor = scale_factor; a.guidance_scale = guidance_scale;
    a.pred = (const h16 *)pred; sds_strides(pred_strides, a.sq);
    a.latents = const_cast<float *>(latents); sds_strides(latents_strides, a.sl);
    a.noise = noise; sds_strides(noise_strides, a.sn);
    a.t = t; a.alphas = alphas_cumprod; a.clip = clip;
    a.moments = (const h16 *)moments; sds_strides(moments_strides, a.sm);
    a.post = (const h16 *)post; sds_strides(post_strides, a.sp);
    a.d_moments = (h16 *)d_moments; sds_strides(d_moments_strides, a.sd);
    a.loss = loss; a.grad_norm = grad_norm;
    hipLaunchKernelGGL(k_sds_finish, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}
