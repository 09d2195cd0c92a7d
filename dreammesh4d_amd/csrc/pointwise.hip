// pointwise.hip -- two fused elementwise operators of the Zero123 SDS step for NHWC / token-major activations (gfx950).
//
//   dm4d_add_bias_nhwc : y[r, c] = a[r, c] + b[r, c] + bias[c]      -- the end of a ResBlock, "skip(x) + conv(h)" with the
//       convolution's bias (extern/ldm_zero123/modules/diffusionmodules/openaimodel.py:259-275, model.py ResnetBlock):
//       MIOpen's NHWC implicit-GEMM convolutions add their bias in a kernel of their own (SubTensorOpWithScalar1d, 4 % of
//       the step in profiles/r02_zero123.md) and the residual add is another: one launch instead of two, one read less.
//   dm4d_geglu         : y[r, d] = p[r, d] * gelu(p[r, D + d])       -- GEGLU of the transformer blocks' feed-forward
//       (extern/ldm_zero123/modules/attention.py:48-56: `x, gate = proj(x).chunk(2, dim=-1); x * F.gelu(gate)`, exact erf
//       GELU): one launch instead of two strided ones.
// HBM-bound streams: 16-byte loads / stores, grid-stride.
#include "common.h"
#include "dm4d.h"

namespace dm4d {

template <typename T> struct PwVec;
template <> struct PwVec<_Float16> { static constexpr int n = 8; typedef _Float16 type __attribute__((ext_vector_type(8))); };
template <> struct PwVec<float> { static constexpr int n = 4; typedef float type __attribute__((ext_vector_type(4))); };

template <typename T>
__global__ __launch_bounds__(256) void k_add_bias(size_t n_vec, int cv, const T *__restrict__ a, const T *__restrict__ b,
                                                  const T *__restrict__ bias, T *__restrict__ y)
{
    using V = typename PwVec<T>::type;
    constexpr int VEC = PwVec<T>::n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
        const V av = reinterpret_cast<const V *>(a)[i], bv = reinterpret_cast<const V *>(b)[i];
        const V cv_ = reinterpret_cast<const V *>(bias)[i % (size_t)cv];
        V o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) o[k] = (T)(((float)av[k] + (float)bv[k]) + (float)cv_[k]);
        reinterpret_cast<V *>(y)[i] = o;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_geglu(size_t rows, int dv, const T *__restrict__ p, T *__restrict__ y)
{
    using V = typename PwVec<T>::type;
    constexpr int VEC = PwVec<T>::n;
    const size_t n_vec = rows * (size_t)dv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (size_t)dv, c = i % (size_t)dv;
        const V xv = reinterpret_cast<const V *>(p)[r * 2 * dv + c], gv = reinterpret_cast<const V *>(p)[r * 2 * dv + dv + c];
        V o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float g = (float)gv[k];
            o[k] = (T)((float)xv[k] * (0.5f * g * (1.0f + erff(g * 0.70710678118654752f))));
        }
        reinterpret_cast<V *>(y)[i] = o;
    }
}

// s = x (+ tok[row / rows_per_sample]);  normed = LayerNorm(s) gamma + beta;  xb = s (+ bias2)      -- float16 rows of C <= 2048
// One WAVE per row: a lane holds up to four 16-byte pieces of the row in registers (read once), mean and variance in float32
// (two passes over the registers), wave sums by xor shuffles.  The transformer block of the UNet
// (extern/ldm_zero123/modules/attention.py:196-213: x = attn1(norm1(x)) + x; x = attn2(norm2(x)) + x; x = ff(norm3(x)) + x)
// evaluated without gradients uses it twice per block: LayerNorm of the running activation together with the residual operand
// of the NEXT GEMM (the output bias already added, so that GEMM's epilogue is just "+ C"), and -- with the single-token
// cross-attention's broadcast row -- the two residual adds and the LayerNorm between the self-attention and the feed-forward.
__global__ __launch_bounds__(256) void k_add_layernorm_f16(size_t rows, int cv, int rows_per_sample, const _Float16 *__restrict__ x,
                                                           const _Float16 *__restrict__ tok, const _Float16 *__restrict__ gamma,
                                                           const _Float16 *__restrict__ beta, float eps, const _Float16 *__restrict__ bias2,
                                                           _Float16 *__restrict__ normed, _Float16 *__restrict__ xb)
{
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const h8 *xr = reinterpret_cast<const h8 *>(x) + row * cv;
    const h8 *tr = tok ? reinterpret_cast<const h8 *>(tok) + (row / (size_t)rows_per_sample) * cv : nullptr;
    float v[4][8];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = lane + 64 * k;
        if (c < cv) {
            const h8 a = xr[c];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[k][j] = (float)a[j];
            if (tr) {
                const h8 t = tr[c];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[k][j] = (float)(_Float16)(v[k][j] + (float)t[j]);      // (the separate add rounded to float16 too)
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[k][j];
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
    const float n = (float)(cv * 8), mean = sum / n;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (lane + 64 * k < cv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float dlt = v[k][j] - mean; sq += dlt * dlt; }
        }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m, 64);
    const float rstd = 1.0f / sqrtf(sq / n + eps);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = lane + 64 * k;
        if (c < cv) {
            const h8 g = reinterpret_cast<const h8 *>(gamma)[c], b = reinterpret_cast<const h8 *>(beta)[c];
            h8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (_Float16)((v[k][j] - mean) * rstd * (float)g[j] + (float)b[j]);
            reinterpret_cast<h8 *>(normed)[row * cv + c] = o;
            if (xb) {
                h8 w;
                if (bias2) {
                    const h8 b2 = reinterpret_cast<const h8 *>(bias2)[c];
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = (_Float16)(v[k][j] + (float)b2[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = (_Float16)v[k][j];
                }
                reinterpret_cast<h8 *>(xb)[row * cv + c] = w;
            }
        }
    }
}

static unsigned pw_blocks(size_t n_vec)
{
    const size_t b = (n_vec + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace dm4d

using namespace dm4d;

extern "C" int dm4d_add_bias_nhwc(int64_t rows, int32_t C, int32_t dtype, const void *a, const void *b, const void *bias, void *y,
                                  dm4d_stream_t stream)
{
    const int vec = dtype == DM4D_GN_F16 ? 8 : 4;
    if ((dtype != DM4D_GN_F16 && dtype != DM4D_GN_F32) || rows < 0 || C <= 0 || C % vec) { set_error("add_bias: rows %lld C %d dtype %d", (long long)rows, C, dtype); return DM4D_ERR_INVALID; }
    if (rows == 0) return DM4D_OK;
    if (!a || !b || !bias || !y) { set_error("add_bias: null pointer"); return DM4D_ERR_INVALID; }
    const size_t n_vec = (size_t)rows * (C / vec);
    if (dtype == DM4D_GN_F16)
        hipLaunchKernelGGL(k_add_bias<_Float16>, dim3(pw_blocks(n_vec)), dim3(256), 0, (hipStream_t)stream, n_vec, C / vec, (const _Float16 *)a, (const _Float16 *)b, (const _Float16 *)bias, (_Float16 *)y);
    else
        hipLaunchKernelGGL(k_add_bias<float>, dim3(pw_blocks(n_vec)), dim3(256), 0, (hipStream_t)stream, n_vec, C / vec, (const float *)a, (const float *)b, (const float *)bias, (float *)y);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

extern "C" int dm4d_geglu(int64_t rows, int32_t D, int32_t dtype, const void *proj, void *y, dm4d_stream_t stream)
{
    const int vec = dtype == DM4D_GN_F16 ? 8 : 4;
    if ((dtype != DM4D_GN_F16 && dtype != DM4D_GN_F32) || rows < 0 || D <= 0 || D % vec) { set_error("geglu: rows %lld D %d dtype %d", (long long)rows, D, dtype); return DM4D_ERR_INVALID; }
    if (rows == 0) return DM4D_OK;
    if (!proj || !y) { set_error("geglu: null pointer"); return DM4D_ERR_INVALID; }
    const size_t n_vec = (size_t)rows * (D / vec);
    if (dtype == DM4D_GN_F16)
        hipLaunchKernelGGL(k_geglu<_Float16>, dim3(pw_blocks(n_vec)), dim3(256), 0, (hipStream_t)stream, (size_t)rows, D / vec, (const _Float16 *)proj, (_Float16 *)y);
    else
        hipLaunchKernelGGL(k_geglu<float>, dim3(pw_blocks(n_vec)), dim3(256), 0, (hipStream_t)stream, (size_t)rows, D / vec, (const float *)proj, (float *)y);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

extern "C" int dm4d_add_layernorm_f16(int64_t rows, int32_t C, int32_t rows_per_sample, const void *x, const void *tok, const void *gamma,
                                      const void *beta, float eps, const void *bias2, void *normed, void *xb, dm4d_stream_t stream)
{
    if (rows < 0 || C <= 0 || C % 8 || C > 2048 || rows_per_sample <= 0) { set_error("add_layernorm: rows %lld C %d (C %% 8 == 0, <= 2048)", (long long)rows, C); return DM4D_ERR_INVALID; }
    if (rows == 0) return DM4D_OK;
    if (!x || !gamma || !beta || !normed) { set_error("add_layernorm: null pointer"); return DM4D_ERR_INVALID; }
    if ((((uintptr_t)x | (uintptr_t)tok | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)bias2 | (uintptr_t)normed | (uintptr_t)xb) & 15) != 0) { set_error("add_layernorm: pointers must be 16-byte aligned"); return DM4D_ERR_INVALID; }
    hipLaunchKernelGGL(k_add_layernorm_f16, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (size_t)rows, C / 8, rows_per_sample,
                       (const _Float16 *)x, (const _Float16 *)tok, (const _Float16 *)gamma, (const _Float16 *)beta, eps, (const _Float16 *)bias2,
                       (_Float16 *)normed, (_Float16 *)xb);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}
