// deform_mlp.hip -- the MLP of the deformation network, fused (gfx950).
//
// After the HexPlane features (hexplane.hip) the reference runs, per query point
// (custom/threestudio-dreammesh4d/geometry/deformation.py:285-305,430-436,507-512):
//     h     = W0 feat + b0                                   Linear(IN, 64)          (feature_out)
//     x     = relu(h)
//     y_k   = x + W1_k x + b1_k                              residual Linear(64, 64) (heads: pos, scales, rot, opacity)
//     out_k = W2_k y_k + b2_k                                Linear(64, {3, 6, 4, 1})
// as ~12 GEMV-sized linears plus ~25 elementwise kernels forward and ~45 backward per step -- for
// 4000 rows.  Here: 2 launches forward, 2 backward, activations kept in LDS.
//
//   k_mlp_pack : transposes the weights once per call into [in][out] order (the forward streams a
//                weight row per input with lanes along `out`: coalesced, L1-resident)
//   k_mlp_fwd  : one workgroup = 32 rows; thread (o, row group) keeps 8 rows of output o in
//                registers; saves h and y_k for the backward
//   k_mlp_bwd  : one workgroup = 32 rows; dy_k, dx, dh, d feat and the workgroup's PARTIAL weight
//                gradients (fixed summation order)
//   k_mlp_reduce: sums the partials over the workgroups in order -> deterministic parameter
//                gradients, no floating-point atomics
//
// Arithmetic is float32 FMA chains in a fixed order (results agree with rocBLAS to rounding).
#include <string.h>

#include "common.h"
#include "../../include/dm4d.h"

namespace dm4d {

constexpr int kW = 64;        // hidden width (DeformationNetwork(net_width=64))
constexpr int kRT = 32;       // rows per workgroup
constexpr int kLd = 36;       // LDS row stride in floats for [k][row] tiles (16-B aligned, 32 rows + pad)
constexpr int kMaxHeads = 4;
constexpr int kMaxOut = 8;
constexpr int kMaxIn = 256;

struct MlpDesc {
    int P, IN, n_heads;
    int out_dim[kMaxHeads];
    const float *W0, *b0;
    const float *W1[kMaxHeads], *b1[kMaxHeads], *W2[kMaxHeads], *b2[kMaxHeads];
    float *W0T;                 // [IN][64]
    float *W1T[kMaxHeads];      // [64][64]
    float *W2T[kMaxHeads];      // [64][out]
};

struct MlpGrads {
    float *W0, *b0;
    float *W1[kMaxHeads], *b1[kMaxHeads], *W2[kMaxHeads], *b2[kMaxHeads];
};

__host__ __device__ static inline size_t partial_floats(const MlpDesc &d)
{
    size_t n = (size_t)kW * d.IN + kW;
    for (int k = 0; k < d.n_heads; ++k) n += (size_t)kW * kW + kW + (size_t)d.out_dim[k] * kW + d.out_dim[k];
    return n;
}

// ------------------------------------------------------------------------------------------ pack
__global__ void k_mlp_pack(MlpDesc d)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    int base = 0;
    if (gid < d.IN * kW) {   // W0 [64][IN] -> W0T [IN][64]
        const int i = gid / kW, o = gid % kW;
        d.W0T[gid] = d.W0[(size_t)o * d.IN + i];
        return;
    }
    base = d.IN * kW;
    for (int k = 0; k < d.n_heads; ++k) {
        if (gid < base + kW * kW) {
            const int e = gid - base, i = e / kW, o = e % kW;
            d.W1T[k][e] = d.W1[k][o * kW + i];
            return;
        }
        base += kW * kW;
        const int od = d.out_dim[k];
        if (gid < base + kW * od) {
            const int e = gid - base, i = e / od, c = e % od;
            d.W2T[k][e] = d.W2[k][c * kW + i];
            return;
        }
        base += kW * od;
    }
}

// ------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void k_mlp_fwd(MlpDesc d, const float *__restrict__ feat, float *__restrict__ Hs,
                                                 float *__restrict__ Ys, float *out0, float *out1, float *out2,
                                                 float *out3)
{
    __shared__ __attribute__((aligned(16))) float s_a[kMaxIn * kLd];   // feat^T [IN][row]
    __shared__ __attribute__((aligned(16))) float s_x[kW * kLd];       // relu(h)^T [64][row]
    __shared__ __attribute__((aligned(16))) float s_y[kW * kLd];       // y_k^T [64][row]
    const int tid = threadIdx.x, o = tid & 63, rg = tid >> 6;
    const int row0 = blockIdx.x * kRT;
    const int IN = d.IN;
    for (int e = tid; e < kRT * IN; e += 256) {
        const int r = e / IN, i = e % IN;
        s_a[i * kLd + r] = (row0 + r < d.P) ? feat[(size_t)(row0 + r) * IN + i] : 0.f;
    }
    __syncthreads();
    float acc[8];
    {
        const float b = d.b0[o];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = b;
#pragma unroll 8
        for (int i = 0; i < IN; ++i) {
            const float w = d.W0T[i * kW + o];
            const float4 f0 = *reinterpret_cast<const float4 *>(&s_a[i * kLd + rg * 8]);
            const float4 f1 = *reinterpret_cast<const float4 *>(&s_a[i * kLd + rg * 8 + 4]);
            acc[0] = __builtin_fmaf(w, f0.x, acc[0]); acc[1] = __builtin_fmaf(w, f0.y, acc[1]);
            acc[2] = __builtin_fmaf(w, f0.z, acc[2]); acc[3] = __builtin_fmaf(w, f0.w, acc[3]);
            acc[4] = __builtin_fmaf(w, f1.x, acc[4]); acc[5] = __builtin_fmaf(w, f1.y, acc[5]);
            acc[6] = __builtin_fmaf(w, f1.z, acc[6]); acc[7] = __builtin_fmaf(w, f1.w, acc[7]);
        }
    }
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = row0 + rg * 8 + j;
        if (r < d.P) Hs[(size_t)r * kW + o] = acc[j];
        x[j] = fmaxf(acc[j], 0.f);
        s_x[o * kLd + rg * 8 + j] = x[j];
    }
    __syncthreads();
    float *outs[kMaxHeads] = {out0, out1, out2, out3};
    for (int k = 0; k < d.n_heads; ++k) {
        const float b = d.b1[k][o];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = x[j] + b;
        const float *__restrict__ W1T = d.W1T[k];
#pragma unroll 8
        for (int i = 0; i < kW; ++i) {
            const float w = W1T[i * kW + o];
            const float4 f0 = *reinterpret_cast<const float4 *>(&s_x[i * kLd + rg * 8]);
            const float4 f1 = *reinterpret_cast<const float4 *>(&s_x[i * kLd + rg * 8 + 4]);
            acc[0] = __builtin_fmaf(w, f0.x, acc[0]); acc[1] = __builtin_fmaf(w, f0.y, acc[1]);
            acc[2] = __builtin_fmaf(w, f0.z, acc[2]); acc[3] = __builtin_fmaf(w, f0.w, acc[3]);
            acc[4] = __builtin_fmaf(w, f1.x, acc[4]); acc[5] = __builtin_fmaf(w, f1.y, acc[5]);
            acc[6] = __builtin_fmaf(w, f1.z, acc[6]); acc[7] = __builtin_fmaf(w, f1.w, acc[7]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = row0 + rg * 8 + j;
            if (r < d.P) Ys[((size_t)k * d.P + r) * kW + o] = acc[j];
            s_y[o * kLd + rg * 8 + j] = acc[j];
        }
        __syncthreads();
        const int od = d.out_dim[k];
        const int r = tid & 31, c = tid >> 5;    // 8 output columns x 32 rows
        if (c < od && row0 + r < d.P) {
            float a = d.b2[k][c];
            const float *__restrict__ W2T = d.W2T[k];
            for (int i = 0; i < kW; ++i) a = __builtin_fmaf(W2T[i * od + c], s_y[i * kLd + r], a);
            outs[k][(size_t)(row0 + r) * od + c] = a;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ backward
// LDS tiles, all [k][row] with stride kLd unless noted
__global__ __launch_bounds__(256) void k_mlp_bwd(MlpDesc d, const float *__restrict__ feat, const float *__restrict__ Hs,
                                                 const float *__restrict__ Ys, const float *g0, const float *g1,
                                                 const float *g2, const float *g3, float *__restrict__ g_feat,
                                                 float *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) float s_a[kRT * (kMaxIn + 4)];  // feat, row-major [row][IN + 4]  (dW0)
    __shared__ __attribute__((aligned(16))) float s_xr[kRT * (kW + 4)];     // relu(h)   [row][64 + 4]       (dW1)
    __shared__ __attribute__((aligned(16))) float s_dy[kW * kLd];           // dy_k^T    [64][row]           (dx, dW1, db1)
    __shared__ __attribute__((aligned(16))) float s_dh[kW * kLd];           // dh^T      [64][row]           (dW0, db0, d feat)
    __shared__ __attribute__((aligned(16))) float s_g[kMaxOut * kLd];       // g_out_k^T [out][row]
    const int tid = threadIdx.x, o = tid & 63, rg = tid >> 6;
    const int row0 = blockIdx.x * kRT;
    const int IN = d.IN, lda = IN + 4, ldx = kW + 4;
    float *__restrict__ part = partial + (size_t)blockIdx.x * partial_floats(d);
    const float *gs[kMaxHeads] = {g0, g1, g2, g3};

    for (int e = tid; e < kRT * IN; e += 256) {
        const int r = e / IN, i = e % IN;
        s_a[r * lda + i] = (row0 + r < d.P) ? feat[(size_t)(row0 + r) * IN + i] : 0.f;
    }
    float hpre[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = row0 + rg * 8 + j;
        hpre[j] = (r < d.P) ? Hs[(size_t)r * kW + o] : 0.f;
        s_xr[(rg * 8 + j) * ldx + o] = fmaxf(hpre[j], 0.f);
    }
    float dx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dx[j] = 0.f;
    size_t poff = (size_t)kW * IN + kW;    // head partials follow dW0, db0
    __syncthreads();

    for (int k = 0; k < d.n_heads; ++k) {
        const int od = d.out_dim[k];
        // ---- g_out tile ----
        for (int e = tid; e < kRT * od; e += 256) {
            const int r = e / od, c = e % od;
            s_g[c * kLd + r] = (gs[k] && row0 + r < d.P) ? gs[k][(size_t)(row0 + r) * od + c] : 0.f;
        }
        __syncthreads();
        // ---- dy = W2^T g  (thread: output o, 8 rows) ----
        float dy[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) dy[j] = 0.f;
        for (int c = 0; c < od; ++c) {
            const float w = d.W2[k][c * kW + o];
            const float4 f0 = *reinterpret_cast<const float4 *>(&s_g[c * kLd + rg * 8]);
            const float4 f1 = *reinterpret_cast<const float4 *>(&s_g[c * kLd + rg * 8 + 4]);
            dy[0] = __builtin_fmaf(w, f0.x, dy[0]); dy[1] = __builtin_fmaf(w, f0.y, dy[1]);
            dy[2] = __builtin_fmaf(w, f0.z, dy[2]); dy[3] = __builtin_fmaf(w, f0.w, dy[3]);
            dy[4] = __builtin_fmaf(w, f1.x, dy[4]); dy[5] = __builtin_fmaf(w, f1.y, dy[5]);
            dy[6] = __builtin_fmaf(w, f1.z, dy[6]); dy[7] = __builtin_fmaf(w, f1.w, dy[7]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s_dy[o * kLd + rg * 8 + j] = dy[j];
            dx[j] += dy[j];                        // y = x + W1 x + b1: the identity branch
        }
        __syncthreads();
        // ---- dx += W1^T dy ----
        {
            const float *__restrict__ W1 = d.W1[k];
#pragma unroll 8
            for (int jn = 0; jn < kW; ++jn) {
                const float w = W1[jn * kW + o];
                const float4 f0 = *reinterpret_cast<const float4 *>(&s_dy[jn * kLd + rg * 8]);
                const float4 f1 = *reinterpret_cast<const float4 *>(&s_dy[jn * kLd + rg * 8 + 4]);
                dx[0] = __builtin_fmaf(w, f0.x, dx[0]); dx[1] = __builtin_fmaf(w, f0.y, dx[1]);
                dx[2] = __builtin_fmaf(w, f0.z, dx[2]); dx[3] = __builtin_fmaf(w, f0.w, dx[3]);
                dx[4] = __builtin_fmaf(w, f1.x, dx[4]); dx[5] = __builtin_fmaf(w, f1.y, dx[5]);
                dx[6] = __builtin_fmaf(w, f1.z, dx[6]); dx[7] = __builtin_fmaf(w, f1.w, dx[7]);
            }
        }
        // ---- partial dW1[jn][i] = sum_r dy[r][jn] x[r][i]: thread (i = o, jn = rg*16 .. +15) ----
        {
            float a[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = 0.f;
#pragma unroll 4
            for (int r = 0; r < kRT; ++r) {
                const float xv = s_xr[r * ldx + o];
#pragma unroll
                for (int q = 0; q < 16; ++q) a[q] = __builtin_fmaf(s_dy[(rg * 16 + q) * kLd + r], xv, a[q]);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) part[poff + (size_t)(rg * 16 + q) * kW + o] = a[q];
        }
        // ---- partial db1[o] (wave 0), dW2[c][o], db2[c] ----
        if (rg == 0) {
            float a = 0.f;
            for (int r = 0; r < kRT; ++r) a += s_dy[o * kLd + r];
            part[poff + (size_t)kW * kW + o] = a;
        }
        for (int c = rg; c < od; c += 4) {
            float a = 0.f;
            for (int r = 0; r < kRT; ++r) {
                const float yv = (row0 + r < d.P) ? Ys[((size_t)k * d.P + row0 + r) * kW + o] : 0.f;
                a = __builtin_fmaf(s_g[c * kLd + r], yv, a);
            }
            part[poff + (size_t)kW * kW + kW + (size_t)c * kW + o] = a;
        }
        if (tid < od) {
            float a = 0.f;
            for (int r = 0; r < kRT; ++r) a += s_g[tid * kLd + r];
            part[poff + (size_t)kW * kW + kW + (size_t)od * kW + tid] = a;
        }
        poff += (size_t)kW * kW + kW + (size_t)od * kW + od;
        __syncthreads();
    }
    // ---- dh = dx * (h > 0) ----
#pragma unroll
    for (int j = 0; j < 8; ++j) s_dh[o * kLd + rg * 8 + j] = hpre[j] > 0.f ? dx[j] : 0.f;
    __syncthreads();
    // ---- partial dW0[oo][i] = sum_r dh[r][oo] feat[r][i]: thread i (IN / 64 columns each... ) ----
    // thread (i = tid % IN_T, og): IN may be 64..256; each thread owns column(s) i and a slice of outputs
    {
        const int cols = IN;                       // columns
        const int tpc = 256 / min(256, cols);      // threads per column group (1, 2 or 4)
        const int per = kW / tpc;                  // outputs per thread (64, 32 or 16)
        for (int i = tid % (256 / tpc); i < cols; i += 256 / tpc) {
            const int og = tid / (256 / tpc);
            for (int o0 = 0; o0 < per; o0 += 16) {
                float a[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) a[q] = 0.f;
#pragma unroll 4
                for (int r = 0; r < kRT; ++r) {
                    const float fv = s_a[r * lda + i];
#pragma unroll
                    for (int q = 0; q < 16; ++q) a[q] = __builtin_fmaf(s_dh[(og * per + o0 + q) * kLd + r], fv, a[q]);
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) part[(size_t)(og * per + o0 + q) * IN + i] = a[q];
            }
        }
    }
    if (rg == 0) {
        float a = 0.f;
        for (int r = 0; r < kRT; ++r) a += s_dh[o * kLd + r];
        part[(size_t)kW * IN + o] = a;
    }
    // ---- d feat[r][i] = sum_o W0[o][i] dh[r][o]: thread (i, 16 or 8 rows) ----
    if (g_feat) {
        const int tpc = 256 / min(256, IN);        // row groups
        const int rows = kRT / tpc;                // rows per thread: 32, 16 or 8
        for (int i = tid % (256 / tpc); i < IN; i += 256 / tpc) {
            const int rg2 = tid / (256 / tpc);
            for (int r0 = 0; r0 < rows; r0 += 8) {
                float a[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) a[q] = 0.f;
#pragma unroll 8
                for (int oo = 0; oo < kW; ++oo) {
                    const float w = d.W0[(size_t)oo * IN + i];
                    const float4 f0 = *reinterpret_cast<const float4 *>(&s_dh[oo * kLd + rg2 * rows + r0]);
                    const float4 f1 = *reinterpret_cast<const float4 *>(&s_dh[oo * kLd + rg2 * rows + r0 + 4]);
                    a[0] = __builtin_fmaf(w, f0.x, a[0]); a[1] = __builtin_fmaf(w, f0.y, a[1]);
                    a[2] = __builtin_fmaf(w, f0.z, a[2]); a[3] = __builtin_fmaf(w, f0.w, a[3]);
                    a[4] = __builtin_fmaf(w, f1.x, a[4]); a[5] = __builtin_fmaf(w, f1.y, a[5]);
                    a[6] = __builtin_fmaf(w, f1.z, a[6]); a[7] = __builtin_fmaf(w, f1.w, a[7]);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = row0 + rg2 * rows + r0 + q;
                    if (r < d.P) g_feat[(size_t)r * IN + i] = a[q];
                }
            }
        }
    }
}

// sum of the workgroup partials in workgroup order
__global__ void k_mlp_reduce(MlpDesc d, MlpGrads g, const float *__restrict__ partial, int n_wg)
{
    const size_t n = partial_floats(d);
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    // fixed order: four interleaved running sums (loads of different workgroups in flight), combined at the end
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int w = 0;
    for (; w + 4 <= n_wg; w += 4) {
        a0 += partial[(size_t)w * n + e];
        a1 += partial[(size_t)(w + 1) * n + e];
        a2 += partial[(size_t)(w + 2) * n + e];
        a3 += partial[(size_t)(w + 3) * n + e];
    }
    for (; w < n_wg; ++w) a0 += partial[(size_t)w * n + e];
    const float a = (a0 + a1) + (a2 + a3);
    size_t base = 0;
    if (e < (size_t)kW * d.IN) { if (g.W0) g.W0[e] = a; return; }
    base = (size_t)kW * d.IN;
    if (e < base + kW) { if (g.b0) g.b0[e - base] = a; return; }
    base += kW;
    for (int k = 0; k < d.n_heads; ++k) {
        const int od = d.out_dim[k];
        if (e < base + kW * kW) { if (g.W1[k]) g.W1[k][e - base] = a; return; }
        base += kW * kW;
        if (e < base + kW) { if (g.b1[k]) g.b1[k][e - base] = a; return; }
        base += kW;
        if (e < base + (size_t)od * kW) { if (g.W2[k]) g.W2[k][e - base] = a; return; }
        base += (size_t)od * kW;
        if (e < base + od) { if (g.b2[k]) g.b2[k][e - base] = a; return; }
        base += od;
    }
}

static int fill_mlp(MlpDesc &d, int P, const dm4d_mlp_weights *w, void *scratch)
{
    if (!w || P < 0) { set_error("deform_mlp: null weights / negative P"); return DM4D_ERR_INVALID; }
    if (w->width != kW) { set_error("deform_mlp: width %d not supported (64 only)", w->width); return DM4D_ERR_UNSUPPORTED; }
    if (w->in_dim <= 0 || w->in_dim > kMaxIn || (w->in_dim % 64) != 0) {
        set_error("deform_mlp: in_dim %d not supported (64, 128, 192, 256)", w->in_dim);
        return DM4D_ERR_UNSUPPORTED;
    }
    if (w->n_heads < 1 || w->n_heads > kMaxHeads || !w->W0 || !w->b0) { set_error("deform_mlp: bad heads / W0"); return DM4D_ERR_INVALID; }
    memset(&d, 0, sizeof(d));
    d.P = P; d.IN = w->in_dim; d.n_heads = w->n_heads;
    d.W0 = w->W0; d.b0 = w->b0;
    float *s = (float *)scratch;
    d.W0T = s; s += (size_t)kW * d.IN;
    for (int k = 0; k < d.n_heads; ++k) {
        if (w->out_dim[k] < 1 || w->out_dim[k] > kMaxOut || !w->W1[k] || !w->b1[k] || !w->W2[k] || !w->b2[k]) {
            set_error("deform_mlp: head %d incomplete or out_dim %d > %d", k, w->out_dim[k], kMaxOut);
            return DM4D_ERR_INVALID;
        }
        d.out_dim[k] = w->out_dim[k];
        d.W1[k] = w->W1[k]; d.b1[k] = w->b1[k]; d.W2[k] = w->W2[k]; d.b2[k] = w->b2[k];
        d.W1T[k] = s; s += kW * kW;
        d.W2T[k] = s; s += kW * kMaxOut;
    }
    return DM4D_OK;
}

}  // namespace dm4d

using namespace dm4d;

extern "C" {

/* scratch: transposed weights + per-workgroup partial weight gradients */
size_t dm4d_deform_mlp_scratch_bytes(int32_t P, int32_t in_dim, int32_t n_heads)
{
    const size_t wg = (size_t)((P > 0 ? P : 1) + kRT - 1) / kRT;
    const size_t per_wg = (size_t)kW * in_dim + kW + (size_t)n_heads * (kW * kW + kW + kMaxOut * kW + kMaxOut);
    const size_t wt = (size_t)kW * in_dim + (size_t)n_heads * (kW * kW + kW * kMaxOut);
    return (wt + wg * per_wg) * sizeof(float) + 256;
}

int dm4d_deform_mlp_forward(int32_t P, const float *feat, const dm4d_mlp_weights *w, float *h_save, float *y_save,
                            float *const *out, void *scratch, dm4d_stream_t stream)
{
    MlpDesc d;
    int rc = fill_mlp(d, P, w, scratch);
    if (rc) return rc;
    if (P == 0) return DM4D_OK;
    if (!feat || !h_save || !y_save || !out || !scratch) { set_error("deform_mlp: null tensor"); return DM4D_ERR_INVALID; }
    float *o[kMaxHeads] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < d.n_heads; ++k) {
        o[k] = out[k];
        if (!o[k]) { set_error("deform_mlp: null output %d", k); return DM4D_ERR_INVALID; }
    }
    hipStream_t st = (hipStream_t)stream;
    int n_w = d.IN * kW;
    for (int k = 0; k < d.n_heads; ++k) n_w += kW * kW + kW * d.out_dim[k];
    hipLaunchKernelGGL(k_mlp_pack, dim3((n_w + 255) / 256), dim3(256), 0, st, d);
    DM4D_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_mlp_fwd, dim3((P + kRT - 1) / kRT), dim3(256), 0, st, d, feat, h_save, y_save, o[0], o[1], o[2], o[3]);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_deform_mlp_backward(int32_t P, const float *feat, const dm4d_mlp_weights *w, const float *h_save,
                             const float *y_save, const float *const *g_out, float *g_feat,
                             const dm4d_mlp_weights_grad *gw, void *scratch, dm4d_stream_t stream)
{
    MlpDesc d;
    int rc = fill_mlp(d, P, w, scratch);
    if (rc) return rc;
    if (!gw) { set_error("deform_mlp: null gradient struct"); return DM4D_ERR_INVALID; }
    if (P == 0) return DM4D_OK;
    if (!feat || !h_save || !y_save || !g_out || !scratch) { set_error("deform_mlp: null tensor"); return DM4D_ERR_INVALID; }
    const float *g[kMaxHeads] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < d.n_heads; ++k) g[k] = g_out[k];   // NULL = zero gradient for that head
    MlpGrads mg;
    memset(&mg, 0, sizeof(mg));
    mg.W0 = gw->W0; mg.b0 = gw->b0;
    for (int k = 0; k < d.n_heads; ++k) { mg.W1[k] = gw->W1[k]; mg.b1[k] = gw->b1[k]; mg.W2[k] = gw->W2[k]; mg.b2[k] = gw->b2[k]; }
    hipStream_t st = (hipStream_t)stream;
    const int n_wg = (P + kRT - 1) / kRT;
    size_t wt = (size_t)kW * d.IN + (size_t)d.n_heads * (kW * kW + kW * kMaxOut);
    float *partial = (float *)scratch + wt;
    hipLaunchKernelGGL(k_mlp_bwd, dim3(n_wg), dim3(256), 0, st, d, feat, h_save, y_save, g[0], g[1], g[2], g[3], g_feat, partial);
    DM4D_HIP_CHECK(hipGetLastError());
    const size_t n = partial_floats(d);
    hipLaunchKernelGGL(k_mlp_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, mg, (const float *)partial, n_wg);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // extern "C"
