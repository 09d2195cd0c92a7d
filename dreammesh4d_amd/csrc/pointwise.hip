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
