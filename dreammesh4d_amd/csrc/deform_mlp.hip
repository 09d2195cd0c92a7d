// deform_mlp.hip -- the MLP of the deformation network on the FP32 matrix cores (gfx950).
//
// After the HexPlane features (hexplane.hip) the reference runs, per query point
// (custom/threestudio-dreammesh4d/geometry/deformation.py:285-305,430-436,507-512):
//     h     = W0 feat + b0                                   Linear(IN, 64)          (feature_out)
//     x     = relu(h)
//     y_k   = x + W1_k x + b1_k                              residual Linear(64, 64) (heads: pos, scales, rot, opacity)
//     out_k = W2_k y_k + b2_k                                Linear(64, {3, 6, 4, 1})
// as ~12 GEMV-sized linears plus ~25 elementwise kernels forward and ~45 backward per step -- for
// 4000 rows.  Here every product is a chain of v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: exact
// float32, bitwise an fmaf chain), 1 launch forward and 3 backward.
//
// Everything is computed TRANSPOSED: features are the M dimension of the MFMA, the 16 data rows of a
// workgroup its N dimension.  The D tile of one layer -- lane l holds features 4 (l >> 4) .. + 3 of data row
// l & 15 -- is then exactly the B operand the next layer wants (B[k = l >> 4][n = l & 15], one MFMA per
// register), so activations go from layer to layer without a transpose; only the four 16-feature tiles of a
// 64-wide layer (one per wave) are exchanged through 4 KB of LDS.  The A operands are rows of the weights
// in their nn.Linear layout ([out][in]: lane l reads W[16 w + (l & 15)][16 s + 4 (l >> 4) .. + 3] as one
// float4); the backward's transposed products read [in][out] copies that the forward writes on the side.
//
//   k_mlp_fwd   : workgroup = 16 rows x 4 waves (wave = 16 of the 64 features); saves h and y_k
//   k_mlp_bwd   : workgroup = 16 rows: dy_k, dx, dh (kept for k_mlp_wgrad), d feat
//   k_mlp_wgrad : every parameter gradient is sum_rows L[row][m] R[row][n]; one workgroup per 16x16 output
//                 tile and row slice (split-K), rows in a fixed order
//   k_mlp_reduce: sums the row-slice partials in order -> deterministic, no floating-point atomics
#include <string.h>

#include "common.h"
#include "../../include/dm4d.h"

namespace dm4d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kW = 64;        // hidden width (DeformationNetwork(net_width=64))
constexpr int kRT = 16;       // rows per workgroup == N of the MFMA
constexpr int kMaxHeads = 4;
constexpr int kMaxOut = 8;
constexpr int kMaxIn = 256;
constexpr int kXs = 17;       // float4 stride of a feature quad in the LDS exchange tiles (16 rows + pad)
constexpr int kKSplit = 8;    // row slices of the parameter-gradient products
constexpr int kMaxTickets = 4 * (kMaxIn / 16 + 1) + 25 * kMaxHeads + 4;      // parameter-gradient tiles (rounded up to a multiple of 4)

struct MlpDesc {
    int P, IN, n_heads;
    int out_dim[kMaxHeads];
    const float *W0, *b0;
    const float *W1[kMaxHeads], *b1[kMaxHeads], *W2[kMaxHeads], *b2[kMaxHeads];
    float *W0T;                 // [IN][64]          transposed copies for the backward (written by the forward)
    float *W1T[kMaxHeads];      // [64 in][64 out]
    float *DY;                  // [n_heads][P][64]  dL/dy_k   (backward)
    float *DH;                  // [P][64]           dL/dh     (backward)
    float *partial;             // [kKSplit][partial_floats]
    unsigned *tickets;          // [kMaxTickets]
};

struct MlpGrads {
    float *W0, *b0;
    float *W1[kMaxHeads], *b1[kMaxHeads], *W2[kMaxHeads], *b2[kMaxHeads];
};

// partial layout: W0 [64][IN] | b0 [64] | per head: W1 [64][64] | b1 [64] | W2 [od][64] | b2 [od]
__host__ __device__ static inline size_t partial_floats(const MlpDesc &d)
{
    size_t n = (size_t)kW * d.IN + kW;
    for (int k = 0; k < d.n_heads; ++k) n += (size_t)kW * kW + kW + (size_t)d.out_dim[k] * kW + d.out_dim[k];
    return n;
}

__device__ __forceinline__ f32x4 mfma4(const float4 a, const float4 b, f32x4 c)
{
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 to4(const f32x4 v) { return make_float4(v.x, v.y, v.z, v.w); }

// ------------------------------------------------------------------------------------------ forward
// the workgroup's 16 rows x IN feature tile is in s_f ([feature quad][row], written by the caller; this function starts
// with the barrier that publishes it); bx / nbx: the workgroup's index / count among the forward's workgroups
__device__ __forceinline__ void mlp_fwd_block(const MlpDesc &d, const int bx, const int nbx, const float4 *s_f, float *__restrict__ Hs,
                                              float *__restrict__ Ys, float *out0, float *out1, float *out2, float *out3)
{
    __shared__ float4 s_x[(kW / 4) * kXs];              // relu(h)
    __shared__ float4 s_y[kMaxHeads][(kW / 4) * kXs];   // y_k
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, kq = lane >> 4;
    const int row0 = bx * kRT, IN = d.IN;
    const int row = row0 + i;                            // this lane's data row (N index of every tile)
    const int fo = 16 * w + 4 * kq;                      // first of this lane's 4 output features
    const float4 b0 = ld4(d.b0 + fo);
    f32x4 acc0 = {b0.x, b0.y, b0.z, b0.w}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // every A operand of the wave (its 16 rows of W0 and of the heads' W1) is requested up front: one memory
    // round trip for the whole kernel instead of one per 16 inputs
    float4 wa[kMaxIn / 16], wh[kMaxHeads][4];
    {
        const float *__restrict__ Wrow = d.W0 + (size_t)(16 * w + i) * IN + 4 * kq;
#pragma unroll
        for (int s = 0; s < kMaxIn / 16; ++s)
            if (16 * s < IN) wa[s] = ld4(Wrow + 16 * s);
#pragma unroll
        for (int k = 0; k < kMaxHeads; ++k)
            if (k < d.n_heads) {
                const float *__restrict__ W1row = d.W1[k] + (size_t)(16 * w + i) * kW + 4 * kq;
#pragma unroll
                for (int s = 0; s < 4; ++s) wh[k][s] = ld4(W1row + 16 * s);
            }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kMaxIn / 16; s += 2) {           // two accumulators: the chains do not wait on each other
        if (16 * s < IN) {
            acc0 = mfma4(wa[s], s_f[(4 * s + kq) * kXs + i], acc0);
            acc1 = mfma4(wa[s + 1], s_f[(4 * s + 4 + kq) * kXs + i], acc1);
        }
    }
    const f32x4 h = acc0 + acc1;
    if (row < d.P) *reinterpret_cast<float4 *>(Hs + (size_t)row * kW + fo) = to4(h);
    const f32x4 x = {fmaxf(h.x, 0.f), fmaxf(h.y, 0.f), fmaxf(h.z, 0.f), fmaxf(h.w, 0.f)};
    s_x[(4 * w + kq) * kXs + i] = to4(x);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kMaxHeads; ++k) {
        if (k >= d.n_heads) break;
        const float4 b1 = ld4(d.b1[k] + fo);
        f32x4 a0 = {x.x + b1.x, x.y + b1.y, x.z + b1.z, x.w + b1.w}, a1 = {0.f, 0.f, 0.f, 0.f};
        a0 = mfma4(wh[k][0], s_x[kq * kXs + i], a0);
        a1 = mfma4(wh[k][1], s_x[(4 + kq) * kXs + i], a1);
        a0 = mfma4(wh[k][2], s_x[(8 + kq) * kXs + i], a0);
        a1 = mfma4(wh[k][3], s_x[(12 + kq) * kXs + i], a1);
        const f32x4 y = a0 + a1;
        if (row < d.P) *reinterpret_cast<float4 *>(Ys + ((size_t)k * d.P + row) * kW + fo) = to4(y);
        s_y[k][(4 * w + kq) * kXs + i] = to4(y);
    }
    __syncthreads();
    // out_k = W2_k y_k + b2_k: wave k takes head k (M = the <= 8 outputs, padded to one 16-row tile)
    if (w < d.n_heads) {
        const int k = w, od = d.out_dim[k];
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * kq + j < od) a[j] = d.b2[k][4 * kq + j];
        const float *__restrict__ Wrow = d.W2[k] + (size_t)i * kW + 4 * kq;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int s = 0; s < 4; ++s) a = mfma4(i < od ? ld4(Wrow + 16 * s) : z, s_y[k][(4 * s + kq) * kXs + i], a);
        float *outs[kMaxHeads] = {out0, out1, out2, out3};
        if (row < d.P) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * kq + j < od) outs[k][(size_t)row * od + 4 * kq + j] = a[j];
        }
    }
    // side job: this workgroup's slice of the [in][out] copies of W0 and W1_k the backward reads
    if (d.W0T) {
        const int n_w = IN * kW + d.n_heads * kW * kW, per = (n_w + nbx - 1) / nbx;
        const int e1 = min(n_w, (bx + 1) * per);
        for (int e = bx * per + tid; e < e1; e += 256) {
            if (e < IN * kW) {
                d.W0T[e] = d.W0[(size_t)(e % kW) * IN + e / kW];
            } else {
                const int r = e - IN * kW, k = r / (kW * kW), q = r % (kW * kW);
                d.W1T[k][q] = d.W1[k][(q % kW) * kW + q / kW];
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_mlp_fwd(MlpDesc d, const float *__restrict__ feat, float *__restrict__ Hs,
                                                 float *__restrict__ Ys, float *out0, float *out1, float *out2,
                                                 float *out3)
{
    __shared__ float4 s_f[(kMaxIn / 4) * kXs];          // feat tile   [feature quad][row]
    const int tid = threadIdx.x, row0 = blockIdx.x * kRT, IN = d.IN, nq = IN / 4;
    for (int e = tid; e < kRT * nq; e += 256) {
        const int r = e / nq, q = e % nq;
        s_f[q * kXs + r] = (row0 + r < d.P) ? ld4(feat + (size_t)(row0 + r) * IN + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    mlp_fwd_block(d, (int)blockIdx.x, (int)gridDim.x, s_f, Hs, Ys, out0, out1, out2, out3);
}

// ------------------------------------------------------------------------------------------ backward (activations)
// `feat_sink(row, first feature, d feat of the lane's 4 consecutive features)`: what happens to a tile of dL/dfeat -- k_mlp_bwd stores
// it; the fused node network (nodenet.hip, round 5) turns it into the HexPlane's per-point plane-product gradients on the spot
struct StoreFeatGrad {
    float *g_feat;
    int IN;
    __device__ __forceinline__ void operator()(const int row, const int col, const float4 v) const
    {
        *reinterpret_cast<float4 *>(g_feat + (size_t)row * IN + col) = v;
    }
};
template <typename FeatSink>
__device__ __forceinline__ void mlp_bwd_block(const MlpDesc &d, const int bx, const float *__restrict__ Hs, const float *g0, const float *g1,
                                              const float *g2, const float *g3, const bool want_feat, const FeatSink &feat_sink)
{
    __shared__ float4 s_dy[(kW / 4) * kXs];
    __shared__ float4 s_dh[(kW / 4) * kXs];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, kq = lane >> 4;
    const int row0 = bx * kRT, IN = d.IN;
    const int row = row0 + i;
    const int fo = 16 * w + 4 * kq;
    const float *gs[kMaxHeads] = {g0, g1, g2, g3};
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 hpre = (row < d.P) ? ld4(Hs + (size_t)row * kW + fo) : z;
    f32x4 dx = {0.f, 0.f, 0.f, 0.f};
    // the wave's rows of every W1T (A[m = i][k] = W1[k][16 w + i] = W1T[16 w + i][k]), requested up front
    float4 wh[kMaxHeads][4];
#pragma unroll
    for (int k = 0; k < kMaxHeads; ++k)
        if (k < d.n_heads) {
            const float *__restrict__ Wrow = d.W1T[k] + (size_t)(16 * w + i) * kW + 4 * kq;
#pragma unroll
            for (int s = 0; s < 4; ++s) wh[k][s] = ld4(Wrow + 16 * s);
        }
#pragma unroll
    for (int k = 0; k < kMaxHeads; ++k) {
        if (k >= d.n_heads) break;
        const int od = d.out_dim[k];
        // dy[f][r] = sum_c W2[c][f] g[r][c]      (A[m = i][k = kq] = W2[4 t + kq][16 w + i], B[kq][n = i] = g[row][4 t + kq])
        f32x4 dy = {0.f, 0.f, 0.f, 0.f};
        if (gs[k]) {
            for (int t = 0; 4 * t < od; ++t) {
                const int c = 4 * t + kq;
                const float a = (c < od) ? d.W2[k][c * kW + 16 * w + i] : 0.f;
                const float b = (c < od && row < d.P) ? gs[k][(size_t)row * od + c] : 0.f;
                dy = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, dy, 0, 0, 0);
            }
        }
        if (row < d.P) *reinterpret_cast<float4 *>(d.DY + ((size_t)k * d.P + row) * kW + fo) = to4(dy);
        __syncthreads();                                   // the previous head's readers are done with s_dy
        s_dy[(4 * w + kq) * kXs + i] = to4(dy);
        __syncthreads();
        // dx += dy + W1^T dy
        f32x4 a1 = dy;
        dx = mfma4(wh[k][0], s_dy[kq * kXs + i], dx);
        a1 = mfma4(wh[k][1], s_dy[(4 + kq) * kXs + i], a1);
        dx = mfma4(wh[k][2], s_dy[(8 + kq) * kXs + i], dx);
        a1 = mfma4(wh[k][3], s_dy[(12 + kq) * kXs + i], a1);
        dx = dx + a1;
    }
    const f32x4 dh = {hpre.x > 0.f ? dx.x : 0.f, hpre.y > 0.f ? dx.y : 0.f, hpre.z > 0.f ? dx.z : 0.f, hpre.w > 0.f ? dx.w : 0.f};
    if (row < d.P) *reinterpret_cast<float4 *>(d.DH + (size_t)row * kW + fo) = to4(dh);
    s_dh[(4 * w + kq) * kXs + i] = to4(dh);
    __syncthreads();
    // d feat[r][in] = sum_o W0[o][in] dh[r][o]   (M tiles of 16 inputs, wave w takes tiles w, w + 4, ...)
    if (want_feat) {
        for (int mt = w; mt < IN / 16; mt += 4) {
            const float *__restrict__ Wrow = d.W0T + (size_t)(16 * mt + i) * kW + 4 * kq;   // A[m = i][k = o] = W0[o][16 mt + i]
            float4 wc[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) wc[s] = ld4(Wrow + 16 * s);
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            a0 = mfma4(wc[0], s_dh[kq * kXs + i], a0);
            a1 = mfma4(wc[1], s_dh[(4 + kq) * kXs + i], a1);
            a0 = mfma4(wc[2], s_dh[(8 + kq) * kXs + i], a0);
            a1 = mfma4(wc[3], s_dh[(12 + kq) * kXs + i], a1);
            if (row < d.P) feat_sink(row, 16 * mt + 4 * kq, to4(a0 + a1));
        }
    }
}

__global__ __launch_bounds__(256) void k_mlp_bwd(MlpDesc d, const float *__restrict__ Hs, const float *g0, const float *g1,
                                                 const float *g2, const float *g3, float *__restrict__ g_feat)
{
    mlp_bwd_block(d, (int)blockIdx.x, Hs, g0, g1, g2, g3, g_feat != nullptr, StoreFeatGrad{g_feat, d.IN});
}

// ------------------------------------------------------------------------------------------ backward (parameters)
// out[m][n] = sum_rows L[row][m] R[row][n] for one 16x16 tile of one parameter and one row slice.  Tiles, in
// blockIdx.x order:  dW0 (L = dh, R = feat) 4 x (IN/16 + 1)  |  per head: dW1 (L = dy_k, R = relu(h)) 4 x 5,
// dW2 (L = g_k, R = y_k) 1 x 5.  The last column tile of every group is the bias: R = 1.
// `last` (optional, round 5): the reduction over the row slices inside this launch -- every slice block publishes its partial tile
// (agent-scope release), draws a ticket of its tile, and the block that draws the last one sums the kKSplit partials of the tile IN
// SLICE ORDER (the additions of mlp_reduce_block: bit-identical) straight into the gradient tensors.  8 KB of partials per tile: the
// case cdna_hip_programming.md's split-K recipe calls worth it; it removes the reduce launch from the end of the step's chain.
// DM4D_WGRAD_FINISH: 2 = write-through (sc1) partial stores + sc1 loads, no fence (the recipe's cheaper form); 1 = plain stores + one
// agent-scope release per slice block and one acquire in the last arriver; 0 = no in-launch reduction (the caller launches k_mlp_reduce)
#ifndef DM4D_WGRAD_FINISH
#define DM4D_WGRAD_FINISH 2
#endif
// Variant 2 is NOT a HIP-memory-model program: it relies on gfx942 / gfx950 hardware rules -- an sc1 store is written through to the
// agent-coherent L2 before `s_waitcnt vmcnt(0)` retires it (vmcnt covers stores on gfx9), the ticket atomic executes at the L2 behind
// it, and sc1 loads by the last arriver bypass its L1 (cdna_hip_programming.md, split-K recipe, "write-through" form).  Any other
// target must build the fenced form.
#if DM4D_WGRAD_FINISH == 2 && defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "DM4D_WGRAD_FINISH=2 (write-through split-K partials without a fence) is only valid on gfx942 / gfx950: build with -DDM4D_WGRAD_FINISH=1"
#endif
struct WgradFinish { unsigned *tickets; const MlpGrads *g; };      // tickets[tile]: zeroed by an EARLIER launch of the stream
__device__ __forceinline__ void mlp_wgrad_block(const MlpDesc &d, const int bx, const int by, const float *__restrict__ feat,
                                                const float *__restrict__ Hs, const float *__restrict__ Ys, const float *g0,
                                                const float *g1, const float *g2, const float *g3, const WgradFinish *last = nullptr)
{
    __shared__ float s_part[3][64][5];      // ([0][0][4]: the "this block reduces" flag of the in-launch reduction -- one LDS object, not two)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, kq = lane >> 4;
    const float *gs[kMaxHeads] = {g0, g1, g2, g3};
    const int IN = d.IN, P = d.P;
    // ---- decode the tile ----
    int t = bx;
    const float *L = nullptr, *R = nullptr;
    int ldL = kW, ldR = kW, Mdim = kW, Ndim = kW, mt = 0, nt = 0;
    bool relu = false;
    size_t w_off = 0, b_off = 0;          // offsets of the weight / bias block inside a partial
    const int n0 = 4 * (IN / 16 + 1);
    if (t < n0) {
        mt = t / (IN / 16 + 1); nt = t % (IN / 16 + 1);
        L = d.DH; R = feat; ldR = IN; Ndim = IN;
        w_off = 0; b_off = (size_t)kW * IN;
    } else {
        t -= n0;
        const int k = t / 25, u = t % 25;
        size_t off = (size_t)kW * IN + kW;
        for (int kk = 0; kk < k; ++kk) off += (size_t)kW * kW + kW + (size_t)d.out_dim[kk] * kW + d.out_dim[kk];
        if (u < 20) {
            mt = u / 5; nt = u % 5;
            L = d.DY + (size_t)k * P * kW; R = Hs; relu = true;
            w_off = off; b_off = off + (size_t)kW * kW;
        } else {
            mt = 0; nt = u - 20;
            const int od = d.out_dim[k];
            L = gs[k]; ldL = od; Mdim = od;
            R = Ys + (size_t)k * P * kW;
            w_off = off + (size_t)kW * kW + kW; b_off = w_off + (size_t)od * kW;
        }
    }
    const bool bias = nt * 16 >= Ndim;
    // ---- this workgroup's row slice, in steps of 4 rows (the K of one MFMA); wave w takes steps w, w + 4, ... ----
    const int steps = (P + 3) / 4, per = (steps + kKSplit - 1) / kKSplit;
    const int s0 = by * per, s1 = min(steps, s0 + per);
    const int mcol = 16 * mt + i, ncol = 16 * nt + i;
    const bool mok = L != nullptr && mcol < Mdim;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int kU = 8;                  // steps whose operands are requested together
    for (int sb = s0 + w; sb < s1; sb += 4 * kU) {
        float a[kU], b[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int s = sb + 4 * u, r = 4 * s + kq;
            const bool rok = s < s1 && r < P;
            a[u] = (mok && rok) ? L[(size_t)r * ldL + mcol] : 0.f;
            b[u] = !rok ? 0.f : bias ? 1.f : R[(size_t)r * ldR + ncol];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], relu ? fmaxf(b[u], 0.f) : b[u], acc, 0, 0, 0);
    }
    // ---- waves 1..3 -> LDS, wave 0 adds them in order and writes the partial ----
    if (w > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s_part[w - 1][lane][j] = acc[j];
    }
    __syncthreads();
    if (w == 0) {
        float *part = d.partial + (size_t)by * partial_floats(d);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = ((acc[j] + s_part[0][lane][j]) + s_part[1][lane][j]) + s_part[2][lane][j];
            const int m = 16 * mt + 4 * kq + j;     // D: row = 4 (lane >> 4) + j, column = lane & 15
            if (m >= Mdim) continue;
            // (in-launch reduction: the partial leaves as a WRITE-THROUGH store -- an agent-scope relaxed atomic store lowers to
            //  `global_store_dword ... sc1` -- so that no release fence is needed before the ticket)
            float *dst = bias ? part + b_off + m : part + w_off + (size_t)m * Ndim + ncol;
            if (bias && i != 0) continue;
            if (last && DM4D_WGRAD_FINISH == 2) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst = v;
        }
    }
    if (!last) return;
    // ---- in-launch reduction: publish, ticket, the last arriver sums (cdna_hip_programming.md, split-K recipe / Guideline 16) ----
    if (w == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            if (DM4D_WGRAD_FINISH == 1) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const unsigned ticket = __hip_atomic_fetch_add(last->tickets + bx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool is_last = ticket == (unsigned)(kKSplit - 1);
            if (is_last && DM4D_WGRAD_FINISH == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            s_part[0][0][4] = is_last ? 1.f : 0.f;
        }
    }
    __syncthreads();
    if (s_part[0][0][4] == 0.f || w != 0) return;
    {
        const MlpGrads &g = *last->g;
        const size_t n = partial_floats(d);
        // the tile's parameter: the decode above, as pointers
        float *Wg = nullptr, *Bg = nullptr;
        if (bx < n0) { Wg = g.W0; Bg = g.b0; }
        else {
            const int k = (bx - n0) / 25, u = (bx - n0) % 25;
            if (u < 20) { Wg = g.W1[k]; Bg = g.b1[k]; }
            else { Wg = g.W2[k]; Bg = g.b2[k]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = 16 * mt + 4 * kq + j;
            if (m >= Mdim || (bias && i != 0)) continue;
            const size_t e = bias ? b_off + m : w_off + (size_t)m * Ndim + ncol;
            float a = 0.f;
            if (DM4D_WGRAD_FINISH == 2) {      // write-through partials are read back with sc1 loads (agent-scope relaxed atomic loads): past the L1
                float pv[kKSplit];
#pragma unroll
                for (int sl = 0; sl < kKSplit; ++sl) pv[sl] = __hip_atomic_load(d.partial + (size_t)sl * n + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int sl = 0; sl < kKSplit; ++sl) a += pv[sl];
            } else {
                for (int sl = 0; sl < kKSplit; ++sl) a += d.partial[(size_t)sl * n + e];
            }
            if (bias) { if (Bg) Bg[m] = a; }
            else if (Wg) Wg[(size_t)m * Ndim + ncol] = a;
        }
    }
}

__global__ __launch_bounds__(256) void k_mlp_wgrad(MlpDesc d, const float *__restrict__ feat, const float *__restrict__ Hs,
                                                   const float *__restrict__ Ys, const float *g0, const float *g1,
                                                   const float *g2, const float *g3)
{
    mlp_wgrad_block(d, (int)blockIdx.x, (int)blockIdx.y, feat, Hs, Ys, g0, g1, g2, g3);
}

// sum of the row-slice partials in slice order
__device__ __forceinline__ void mlp_reduce_block(const MlpDesc &d, const MlpGrads &g, int n_part, const unsigned bx)
{
    const size_t n = partial_floats(d);
    const size_t e = (size_t)bx * 256 + threadIdx.x;
    if (e >= n) return;
    const float *__restrict__ partial = d.partial;
    float a = 0.f;
    for (int w = 0; w < n_part; ++w) a += partial[(size_t)w * n + e];
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

__global__ __launch_bounds__(256) void k_mlp_reduce(MlpDesc d, MlpGrads g, int n_part) { mlp_reduce_block(d, g, n_part, blockIdx.x); }

// scratch: W0T | W1T x heads | DY | DH | partials   (floats; every block a multiple of 4 floats)
static size_t scratch_floats(int P, int in_dim, int n_heads, size_t *dy_off, size_t *dh_off, size_t *part_off)
{
    size_t n = (size_t)kW * in_dim + (size_t)n_heads * kW * kW;
    if (dy_off) *dy_off = n;
    n += (size_t)n_heads * P * kW;
    if (dh_off) *dh_off = n;
    n += (size_t)P * kW;
    if (part_off) *part_off = n;
    n += (size_t)kKSplit * ((size_t)kW * in_dim + kW + (size_t)n_heads * (kW * kW + kW + kMaxOut * kW + kMaxOut));
    n += kMaxTickets;        // the tiles' arrival tickets of the in-launch reduction (uint32; behind the partials' upper bound)
    return n;
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
    size_t dy_off, dh_off, part_off;
    scratch_floats(P, d.IN, d.n_heads, &dy_off, &dh_off, &part_off);
    d.W0T = s;
    for (int k = 0; k < d.n_heads; ++k) {
        d.W1T[k] = s ? s + (size_t)kW * d.IN + (size_t)k * kW * kW : nullptr;
        if (w->out_dim[k] < 1 || w->out_dim[k] > kMaxOut || !w->W1[k] || !w->b1[k] || !w->W2[k] || !w->b2[k]) {
            set_error("deform_mlp: head %d incomplete or out_dim %d > %d", k, w->out_dim[k], kMaxOut);
            return DM4D_ERR_INVALID;
        }
        d.out_dim[k] = w->out_dim[k];
        d.W1[k] = w->W1[k]; d.b1[k] = w->b1[k]; d.W2[k] = w->W2[k]; d.b2[k] = w->b2[k];
    }
    d.DY = s ? s + dy_off : nullptr;
    d.DH = s ? s + dh_off : nullptr;
    d.partial = s ? s + part_off : nullptr;
    d.tickets = s ? reinterpret_cast<unsigned *>(s + scratch_floats(P, d.IN, d.n_heads, nullptr, nullptr, nullptr) - kMaxTickets) : nullptr;
    return DM4D_OK;
}

static bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace dm4d

using namespace dm4d;

extern "C" {

/* scratch: transposed weights + the backward's dy_k / dh + the row-slice partial parameter gradients */
size_t dm4d_deform_mlp_scratch_bytes(int32_t P, int32_t in_dim, int32_t n_heads)
{
    return scratch_floats(P > 0 ? P : 0, in_dim, n_heads, nullptr, nullptr, nullptr) * sizeof(float) + 256;
}

int dm4d_deform_mlp_forward(int32_t P, const float *feat, const dm4d_mlp_weights *w, float *h_save, float *y_save,
                            float *const *out, void *scratch, dm4d_stream_t stream)
{
    MlpDesc d;
    int rc = fill_mlp(d, P, w, scratch);
    if (rc) return rc;
    if (P == 0) return DM4D_OK;
    if (!feat || !h_save || !y_save || !out || !scratch) { set_error("deform_mlp: null tensor"); return DM4D_ERR_INVALID; }
    bool al = aligned16(feat) && aligned16(h_save) && aligned16(y_save) && aligned16(scratch) && aligned16(w->W0) && aligned16(w->b0);
    float *o[kMaxHeads] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < d.n_heads; ++k) {
        o[k] = out[k];
        if (!o[k]) { set_error("deform_mlp: null output %d", k); return DM4D_ERR_INVALID; }
        al = al && aligned16(w->W1[k]) && aligned16(w->b1[k]) && aligned16(w->W2[k]);
    }
    if (!al) { set_error("deform_mlp: tensors must be 16-byte aligned"); return DM4D_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
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
    if (!(aligned16(feat) && aligned16(h_save) && aligned16(y_save) && aligned16(g_feat) && aligned16(scratch))) {
        set_error("deform_mlp: tensors must be 16-byte aligned");
        return DM4D_ERR_INVALID;
    }
    const float *g[kMaxHeads] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < d.n_heads; ++k) g[k] = g_out[k];   // NULL = zero gradient for that head
    MlpGrads mg;
    memset(&mg, 0, sizeof(mg));
    mg.W0 = gw->W0; mg.b0 = gw->b0;
    for (int k = 0; k < d.n_heads; ++k) { mg.W1[k] = gw->W1[k]; mg.b1[k] = gw->b1[k]; mg.W2[k] = gw->W2[k]; mg.b2[k] = gw->b2[k]; }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_mlp_bwd, dim3((P + kRT - 1) / kRT), dim3(256), 0, st, d, h_save, g[0], g[1], g[2], g[3], g_feat);
    DM4D_HIP_CHECK(hipGetLastError());
    const int n_tiles = 4 * (d.IN / 16 + 1) + 25 * d.n_heads;
    hipLaunchKernelGGL(k_mlp_wgrad, dim3(n_tiles, kKSplit), dim3(256), 0, st, d, feat, h_save, y_save, g[0], g[1], g[2], g[3]);
    DM4D_HIP_CHECK(hipGetLastError());
    const size_t n = partial_floats(d);
    hipLaunchKernelGGL(k_mlp_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, mg, kKSplit);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // extern "C"
