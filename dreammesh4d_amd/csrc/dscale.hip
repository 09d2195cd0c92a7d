// dscale.hip -- the `d_scale: true` branch of the dynamic geometry (C/geometry/dynamic_sugar.py:593-611, 697-704), forward and
// backward: per-vertex scale matrices blended from the graph nodes' strain matrices, and the bound Gaussians' scales as the
// barycentric blend of their three corner vertices' matrices applied to the static scaling.  Off the measured path (the shipped
// configurations set d_scale: false); round 3 moves the two blends from torch (a gather + einsum each, ~12 launches with their
// backward) to four small HIP kernels with atomic-free gather backwards over the same static adjacency lists the skinning
// backward uses.
//
//   S(m)      = I + sym(ds[m])                          ds = (xx, yy, zz, xy, xz, yz)         strain_tensor_to_matrix, :29-39
//   lbs:      Sv[v] = sum_k w[v,k] S(m_k)
//   hybrid:   o = sigmoid(d_opacity),  lw = min(sum_k w[v,k] o[m_k] + 0.4, 1)                  :572-578
//             Sv[v] = sum_k w[v,k] o[m_k] S(m_k) + (1 - lw) I                                  :593-611
//   scales[f, g] = (sum_c bary[g, c] Sv[faces[f, c]]) scaling[f, g]                            :697-704
#include "common.h"
#include "../../include/dm4d.h"

namespace dm4d {

constexpr int kDsThreads = 256;

__device__ __forceinline__ float ds_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// thread / (frame, vertex)
__global__ __launch_bounds__(kDsThreads) void k_vscale_fwd(int V, int K, int M, int hybrid, const int32_t *__restrict__ idx,
                                                           const float *__restrict__ w, const float *__restrict__ ds,
                                                           const float *__restrict__ dop, float *__restrict__ out)
{
    const int v = blockIdx.x * kDsThreads + threadIdx.x, f = blockIdx.y;
    if (v >= V) return;
    float c0 = 0.f, s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, wo = 0.f;
    for (int k = 0; k < K; ++k) {
        const int m = idx[(size_t)v * K + k];
        float c = w[(size_t)v * K + k];
        if (hybrid) {
            c *= ds_sigmoid(dop[(size_t)f * M + m]);
            wo += c;
        }
        const float *d = ds + ((size_t)f * M + m) * 6;
        c0 += c;
#pragma unroll
        for (int j = 0; j < 6; ++j) s[j] += c * d[j];
    }
    if (hybrid) c0 += 1.0f - fminf(wo + 0.4f, 1.0f);
    float *o = out + ((size_t)f * V + v) * 9;
    o[0] = c0 + s[0]; o[1] = s[3]; o[2] = s[4];
    o[3] = s[3]; o[4] = c0 + s[1]; o[5] = s[5];
    o[6] = s[4]; o[7] = s[5]; o[8] = c0 + s[2];
}

// thread / (frame, node): gather over the (vertex, k) pairs that reference the node (static CSR: item = v K + k)
__global__ __launch_bounds__(kDsThreads) void k_vscale_bwd(int V, int K, int M, int hybrid, const int32_t *__restrict__ idx,
                                                           const float *__restrict__ w, const float *__restrict__ ds,
                                                           const float *__restrict__ dop, const int32_t *__restrict__ csr_off,
                                                           const int32_t *__restrict__ csr_items, const float *__restrict__ g_out,
                                                           float *__restrict__ g_ds, float *__restrict__ g_dop)
{
    const int m = blockIdx.x * kDsThreads + threadIdx.x, f = blockIdx.y;
    if (m >= M) return;
    const float *d = ds + ((size_t)f * M + m) * 6;
    const float o = hybrid ? ds_sigmoid(dop[(size_t)f * M + m]) : 1.0f;
    float gd[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, go = 0.f;
    for (int e = csr_off[m]; e < csr_off[m + 1]; ++e) {
        const int item = csr_items[e], v = item / K;
        const float wk = w[item];
        const float *g = g_out + ((size_t)f * V + v) * 9;
        const float tr = g[0] + g[4] + g[8];
        const float sym[6] = {g[0], g[4], g[8], g[1] + g[3], g[2] + g[6], g[5] + g[7]};
        const float c = wk * o;
#pragma unroll
        for (int j = 0; j < 6; ++j) gd[j] += c * sym[j];
        if (hybrid) {
            // <g, S(m)> = tr g + <sym g, ds>;  the (1 - lw) I term passes -w tr g while lw is not clamped
            float dot = tr;
#pragma unroll
            for (int j = 0; j < 6; ++j) dot += sym[j] * d[j];
            float lw = 0.f;
            for (int k = 0; k < K; ++k) lw += w[(size_t)v * K + k] * ds_sigmoid(dop[(size_t)f * M + idx[(size_t)v * K + k]]);
            go += wk * (dot - ((lw + 0.4f < 1.0f) ? tr : 0.f));
        }
    }
    float *o_ds = g_ds + ((size_t)f * M + m) * 6;
#pragma unroll
    for (int j = 0; j < 6; ++j) o_ds[j] = gd[j];
    if (hybrid && g_dop) g_dop[(size_t)f * M + m] = go * o * (1.0f - o);
}

// thread / (frame, Gaussian)
__global__ __launch_bounds__(kDsThreads) void k_gscale_fwd(int F, int G, int V, const int32_t *__restrict__ faces, const float *__restrict__ bary,
                                                           const float *__restrict__ sv, const float *__restrict__ scaling,
                                                           float *__restrict__ out)
{
    const int i = blockIdx.x * kDsThreads + threadIdx.x, f = blockIdx.y, N = F * G;
    if (i >= N) return;
    const int face = i / G, g = i - face * G;
    float D[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float b = bary[g * 3 + c];
        const float *s = sv + ((size_t)f * V + faces[face * 3 + c]) * 9;
#pragma unroll
        for (int j = 0; j < 9; ++j) D[j] += b * s[j];
    }
    const float *sc = scaling + (size_t)i * 3;
    float *o = out + ((size_t)f * N + i) * 3;
#pragma unroll
    for (int r = 0; r < 3; ++r) o[r] = D[3 * r] * sc[0] + D[3 * r + 1] * sc[1] + D[3 * r + 2] * sc[2];
}

// thread / (frame, vertex): gather over the (face, corner) pairs of the vertex (static CSR: item = face 3 + corner)
__global__ __launch_bounds__(kDsThreads) void k_gscale_bwd_vertex(int F, int G, int V, const float *__restrict__ bary,
                                                                  const float *__restrict__ scaling, const int32_t *__restrict__ csr_off,
                                                                  const int32_t *__restrict__ csr_items, const float *__restrict__ g_out,
                                                                  float *__restrict__ g_sv)
{
    const int v = blockIdx.x * kDsThreads + threadIdx.x, f = blockIdx.y, N = F * G;
    if (v >= V) return;
    float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int e = csr_off[v]; e < csr_off[v + 1]; ++e) {
        const int item = csr_items[e], face = item / 3, c = item - face * 3;
        for (int g = 0; g < G; ++g) {
            const float b = bary[g * 3 + c];
            const float *go = g_out + ((size_t)f * N + (size_t)face * G + g) * 3;
            const float *sc = scaling + ((size_t)face * G + g) * 3;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[3 * r + j] += b * go[r] * sc[j];
        }
    }
    float *o = g_sv + ((size_t)f * V + v) * 9;
#pragma unroll
    for (int j = 0; j < 9; ++j) o[j] = acc[j];
}

// thread / Gaussian: dL/dscaling[i][j] = sum_frames sum_r D[f][r][j] dL/dout[f][i][r]
__global__ __launch_bounds__(kDsThreads) void k_gscale_bwd_scaling(int NF, int F, int G, int V, const int32_t *__restrict__ faces,
                                                                   const float *__restrict__ bary, const float *__restrict__ sv,
                                                                   const float *__restrict__ g_out, float *__restrict__ g_scaling)
{
    const int i = blockIdx.x * kDsThreads + threadIdx.x, N = F * G;
    if (i >= N) return;
    const int face = i / G, g = i - face * G;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int f = 0; f < NF; ++f) {
        float D[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float b = bary[g * 3 + c];
            const float *s = sv + ((size_t)f * V + faces[face * 3 + c]) * 9;
#pragma unroll
            for (int j = 0; j < 9; ++j) D[j] += b * s[j];
        }
        const float *go = g_out + ((size_t)f * N + i) * 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] += D[j] * go[0] + D[3 + j] * go[1] + D[6 + j] * go[2];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) g_scaling[(size_t)i * 3 + j] = acc[j];
}

}  // namespace dm4d

using namespace dm4d;

extern "C" {

int dm4d_vertex_scales_forward(int32_t method, int32_t n_frames, int32_t V, int32_t M, int32_t K, const int32_t *nbr_idx, const float *nbr_w,
                               const float *ds, const float *d_opacity, float *out, dm4d_stream_t stream)
{
    if (method != 0 && method != 2) { set_error("vertex scales: method must be lbs or hybrid (the reference defines none for dqs)"); return DM4D_ERR_UNSUPPORTED; }
    if (n_frames < 0 || V < 0 || M <= 0 || K <= 0) { set_error("vertex scales: bad sizes"); return DM4D_ERR_INVALID; }
    if (n_frames == 0 || V == 0) return DM4D_OK;
    const int hybrid = method == 2;
    if (!nbr_idx || !nbr_w || !ds || !out || (hybrid && !d_opacity)) { set_error("vertex scales: null tensor"); return DM4D_ERR_INVALID; }
    hipLaunchKernelGGL(k_vscale_fwd, dim3((V + kDsThreads - 1) / kDsThreads, n_frames), dim3(kDsThreads), 0, (hipStream_t)stream, V, K, M, hybrid,
                       nbr_idx, nbr_w, ds, d_opacity, out);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_vertex_scales_backward(int32_t method, int32_t n_frames, int32_t V, int32_t M, int32_t K, const int32_t *nbr_idx, const float *nbr_w,
                                const float *ds, const float *d_opacity, const int32_t *node_csr_offsets, const int32_t *node_csr_items,
                                const float *dL_dout, float *dL_dds, float *dL_ddo, dm4d_stream_t stream)
{
    if (method != 0 && method != 2) { set_error("vertex scales: method must be lbs or hybrid"); return DM4D_ERR_UNSUPPORTED; }
    if (n_frames < 0 || V < 0 || M <= 0 || K <= 0) { set_error("vertex scales: bad sizes"); return DM4D_ERR_INVALID; }
    if (n_frames == 0) return DM4D_OK;
    const int hybrid = method == 2;
    if (!nbr_idx || !nbr_w || !ds || !node_csr_offsets || !node_csr_items || !dL_dout || !dL_dds || (hybrid && !d_opacity)) { set_error("vertex scales backward: null tensor"); return DM4D_ERR_INVALID; }
    hipLaunchKernelGGL(k_vscale_bwd, dim3((M + kDsThreads - 1) / kDsThreads, n_frames), dim3(kDsThreads), 0, (hipStream_t)stream, V, K, M, hybrid,
                       nbr_idx, nbr_w, ds, d_opacity, node_csr_offsets, node_csr_items, dL_dout, dL_dds, dL_ddo);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_gaussian_scales_forward(int32_t n_frames, int32_t F, int32_t G, int32_t V, const int32_t *faces, const float *bary, const float *vertex_scales,
                                 const float *scaling, float *out, dm4d_stream_t stream)
{
    if (n_frames < 0 || F < 0 || G <= 0 || G > 6 || V <= 0) { set_error("gaussian scales: bad sizes"); return DM4D_ERR_INVALID; }
    if (n_frames == 0 || F == 0) return DM4D_OK;
    if (!faces || !bary || !vertex_scales || !scaling || !out) { set_error("gaussian scales: null tensor"); return DM4D_ERR_INVALID; }
    hipLaunchKernelGGL(k_gscale_fwd, dim3((F * G + kDsThreads - 1) / kDsThreads, n_frames), dim3(kDsThreads), 0, (hipStream_t)stream, F, G, V, faces, bary,
                       vertex_scales, scaling, out);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_gaussian_scales_backward(int32_t n_frames, int32_t F, int32_t G, int32_t V, const int32_t *faces, const float *bary, const float *vertex_scales,
                                  const float *scaling, const int32_t *vert_csr_offsets, const int32_t *vert_csr_items, const float *dL_dout,
                                  float *dL_dvertex_scales, float *dL_dscaling, dm4d_stream_t stream)
{
    if (n_frames < 0 || F < 0 || G <= 0 || G > 6 || V <= 0) { set_error("gaussian scales: bad sizes"); return DM4D_ERR_INVALID; }
    if (n_frames == 0 || F == 0) return DM4D_OK;
    if (!faces || !bary || !vertex_scales || !scaling || !vert_csr_offsets || !vert_csr_items || !dL_dout) { set_error("gaussian scales backward: null tensor"); return DM4D_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    if (dL_dvertex_scales)
        hipLaunchKernelGGL(k_gscale_bwd_vertex, dim3((V + kDsThreads - 1) / kDsThreads, n_frames), dim3(kDsThreads), 0, st, F, G, V, bary, scaling,
                           vert_csr_offsets, vert_csr_items, dL_dout, dL_dvertex_scales);
    if (dL_dscaling)
        hipLaunchKernelGGL(k_gscale_bwd_scaling, dim3((F * G + kDsThreads - 1) / kDsThreads), dim3(kDsThreads), 0, st, n_frames, F, G, V, faces, bary,
                           vertex_scales, dL_dout, dL_dscaling);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // extern "C"
