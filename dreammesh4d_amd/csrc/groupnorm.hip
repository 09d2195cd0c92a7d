// groupnorm.hip -- GroupNorm (+ SiLU) over NHWC activations, forward and input-gradient (gfx950).
//
// Where it sits: the Zero123 SDS step.  Its UNet forward and VAE encoder forward + backward call
// GroupNorm ~100 times per step (extern/ldm_zero123/modules/diffusionmodules/util.py:242-244 GroupNorm32,
// openaimodel.py ResBlock in_layers / out_layers "GroupNorm, SiLU, conv", model.py Normalize + nonlinearity), and
// profiles/r02_zero123.md has the library path at 18 % of the step for the normalisation, 11 % for the NCHW <-> NHWC
// transposes MIOpen wraps around its NHWC convolution kernels, and the SiLU launches on top.  With the activations kept
// NHWC end to end the convolutions need no transposes, and this file is the normalisation for that layout:
//     y = silu?( (x + add[n, c] - mean[n, g]) * rstd[n, g] * gamma[c] + beta[c] )
// `add` (optional; a constant as far as backward is concerned) is the per-(sample, channel) term between a ResBlock's first
// convolution and its second GroupNorm -- the timestep embedding (openaimodel.py:259-275) and that convolution's bias, or the
// bias alone (stride 0 over the samples) in the VAE encoder: folded in here instead of launches of their own.
//
// HBM-bound: an activation is read twice (statistics, apply) and written once; statistics are float32 whatever the
// storage type.  One workgroup owns a contiguous slab of rows (pixels) of one sample -- in NHWC the slab is one
// contiguous range of memory, read with 16-byte loads by consecutive lanes -- and a thread keeps the same 16-byte
// channel column for all its rows, so per-channel partial sums live in registers; channels -> groups goes through LDS
// once per workgroup, slabs -> sample through a [N, splits, G, 2] scratch that the apply kernel's prologue sums in a
// fixed order (no atomics: results are reproducible run to run).  Two launches per call.
//
// Backward (frozen gamma / beta: the guidance models are not trained, only dL/dx is needed): with z = xhat gamma + beta,
// dz = dy silu'(z),  s1 = sum_group dz gamma,  s2 = sum_group dz gamma xhat,  m = elements per group:
//     dx = rstd (dz gamma - (s1 + xhat s2) / m)
// -- the same two-pass structure.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "dm4d.h"

namespace dm4d {

constexpr int kGnThreads = 256;
constexpr int kGnUnroll = 4;
enum GnMode { kGnFwdStats = 0, kGnFwdApply, kGnBwdStats, kGnBwdApply };

template <typename T> struct GnVec;
template <> struct GnVec<_Float16> { static constexpr int n = 8; typedef _Float16 type __attribute__((ext_vector_type(8))); };
template <> struct GnVec<float> { static constexpr int n = 4; typedef float type __attribute__((ext_vector_type(4))); };

struct GnArgs {
    int N, HW, C, G, splits, rows_per_split, silu, add_stride;
    float eps;
    const void *x, *add, *gamma, *beta, *dy;
    void *out;          // y (forward apply) | dx (backward apply)
    float *stats;       // [N, G, 2] mean, rstd
    float *partial;     // [N, splits, G, 2]
    const void *dx_add; // backward apply: dx += dx_add (the gradient that reaches x beside the norm: a residual connection), or NULL
};

__device__ __forceinline__ float gn_sigmoid(float z) { return 1.0f / (1.0f + __expf(-z)); }

template <typename T, int MODE>
__global__ __launch_bounds__(kGnThreads) void k_groupnorm(GnArgs a)
{
    using V = typename GnVec<T>::type;
    constexpr int VEC = GnVec<T>::n;
    constexpr bool kStats = MODE == kGnFwdStats || MODE == kGnBwdStats;
    constexpr bool kBwd = MODE == kGnBwdStats || MODE == kGnBwdApply;
    extern __shared__ float lds[];
    const int n = blockIdx.y, split = blockIdx.x, t = threadIdx.x;
    const int C = a.C, G = a.G, cpg = C / G, cv = C / VEC;
    const int cw = cv < kGnThreads ? cv : kGnThreads, rpp = kGnThreads / cw;     // columns, rows per pass of the workgroup
    const int row0 = split * a.rows_per_split, row1 = min(a.HW, row0 + a.rows_per_split);
    const int tc = t % cw, tr = t / cw;
    const bool row_thread = tr < rpp;
    float *g_a = lds, *g_b = lds + G;                 // per group: mean, rstd
    float *g_c = lds + 2 * G, *g_d = lds + 3 * G;     // per group: s1 / m, s2 / m (backward apply)
    float *red = lds + 4 * G;                         // [256][2] cross-lane sums of the prologue / epilogue
    float *chan = red + 2 * kGnThreads;               // [rpp][C][2] per-channel partial sums (statistics kernels)
    const size_t sample = (size_t)n * a.HW * C;
    const T *__restrict__ x = (const T *)a.x + sample;
    const T *__restrict__ dy = kBwd ? (const T *)a.dy + sample : nullptr;
    const T *__restrict__ dxa = (MODE == kGnBwdApply && a.dx_add) ? (const T *)a.dx_add + sample : nullptr;
    const T *__restrict__ add = a.add ? (const T *)a.add + (size_t)n * a.add_stride : nullptr;

    // ---- prologue: the sample's group statistics, from the slabs' partial sums.  J lanes per group sum every J-th slab, lane 0
    //      of the group adds the J results: a fixed order, and one round trip to the scratch instead of `splits` of them
    const int J = kGnThreads / G, pg = t / J, pj = t % J;
    if (MODE != kGnFwdStats) {
        const bool own = MODE != kGnBwdStats;                  // this kernel sums partials (forward apply, backward apply)
        float s = 0.f, q = 0.f;
        if (own && pg < G) {
            const float *p = a.partial + ((size_t)n * a.splits * G + pg) * 2;
#pragma unroll 4
            for (int i = pj; i < a.splits; i += J) { s += p[(size_t)i * G * 2]; q += p[(size_t)i * G * 2 + 1]; }
        }
        red[2 * t] = s;
        red[2 * t + 1] = q;
        __syncthreads();
        if (t < G) {
            s = 0.f; q = 0.f;
            for (int j = 0; j < J; ++j) { s += red[2 * (t * J + j)]; q += red[2 * (t * J + j) + 1]; }
            const float inv_m = 1.0f / ((float)cpg * (float)a.HW);
            float mean, rstd;
            if (MODE == kGnFwdApply) {
                const float dm = s * inv_m;                                      // mean of the shifted values (see kGnFwdStats)
                mean = (float)x[(size_t)t * cpg] + dm;
                const float var = fmaxf(q * inv_m - dm * dm, 0.f);
                rstd = 1.0f / sqrtf(var + a.eps);
                if (split == 0) { a.stats[((size_t)n * G + t) * 2] = mean; a.stats[((size_t)n * G + t) * 2 + 1] = rstd; }
            } else {
                mean = a.stats[((size_t)n * G + t) * 2];
                rstd = a.stats[((size_t)n * G + t) * 2 + 1];
            }
            g_a[t] = mean;
            g_b[t] = rstd;
            if (MODE == kGnBwdApply) { g_c[t] = s * inv_m; g_d[t] = q * inv_m; }
        }
        __syncthreads();
    }

    for (int c0 = 0; c0 < cv; c0 += cw) {
        const int col = c0 + tc;
        const bool act = row_thread && col < cv;
        const int cb = col * VEC;
        // per-channel constants of this thread's column
        float k0[VEC], k1[VEC], k2[VEC], k3[VEC], k4[VEC], k5[VEC], acc1[VEC], acc2[VEC];
        if (act) {
            const V gv = (MODE == kGnFwdStats) ? V{} : *reinterpret_cast<const V *>((const T *)a.gamma + cb);
            const V bv = (MODE == kGnFwdStats) ? V{} : *reinterpret_cast<const V *>((const T *)a.beta + cb);
            V av = V{};
            if (add) av = *reinterpret_cast<const V *>(add + cb);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                acc1[i] = 0.f; acc2[i] = 0.f;
                const float ad = (float)av[i];
                if (MODE == kGnFwdStats) {
                    // SHIFTED sums: x + ad - K with K = the group's first element of the sample, so that the variance below is
                    // sum d^2 / m - (sum d / m)^2 of values centred within ~a standard deviation of 0 -- the plain two-moment
                    // form cancels catastrophically once |mean| >> std (float32: |mean| / std = 1e3 loses every digit)
                    k0[i] = ad - (float)x[(size_t)((cb + i) / cpg) * cpg];
                    k1[i] = k2[i] = k3[i] = k4[i] = k5[i] = 0.f;
                    continue;
                }
                const int g = (cb + i) / cpg;
                const float mean = g_a[g], rstd = g_b[g], gam = (float)gv[i], bet = (float)bv[i];
                if (MODE == kGnFwdApply) {           // z = (x + ad) A + B
                    k0[i] = rstd * gam;
                    k1[i] = __builtin_fmaf(ad - mean, k0[i], bet);
                    k2[i] = k3[i] = k4[i] = k5[i] = 0.f;
                } else {                             // xhat = x rstd + (-mean rstd);  z = xhat gam + bet
                    k0[i] = rstd; k1[i] = (ad - mean) * rstd; k2[i] = gam; k3[i] = bet;
                    k4[i] = (MODE == kGnBwdApply) ? g_c[g] : 0.f;
                    k5[i] = (MODE == kGnBwdApply) ? g_d[g] : 0.f;
                }
            }
            auto body = [&](const V xv, const V dv, const V rv, T *o) {
                V ov;
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float v = (float)xv[i];
                    if (MODE == kGnFwdStats) {
                        const float w = v + k0[i];
                        acc1[i] += w;
                        acc2[i] = __builtin_fmaf(w, w, acc2[i]);
                    } else if (MODE == kGnFwdApply) {
                        float z = __builtin_fmaf(v, k0[i], k1[i]);
                        if (a.silu) z = z * gn_sigmoid(z);
                        ov[i] = (T)z;
                    } else {
                        const float xh = __builtin_fmaf(v, k0[i], k1[i]);
                        float dz = (float)dv[i];
                        if (a.silu) {
                            const float z = __builtin_fmaf(xh, k2[i], k3[i]);
                            const float s = gn_sigmoid(z);
                            dz = dz * (s * __builtin_fmaf(z, 1.0f - s, 1.0f));
                        }
                        if (MODE == kGnBwdStats) {
                            acc1[i] += dz;
                            acc2[i] = __builtin_fmaf(dz, xh, acc2[i]);
                        } else {
                            const float corr = __builtin_fmaf(xh, k5[i], k4[i]);
                            ov[i] = (T)(k0[i] * __builtin_fmaf(dz, k2[i], -corr) + (float)rv[i]);
                        }
                    }
                }
                if (!kStats) *reinterpret_cast<V *>(o) = ov;
            };
            T *out = kStats ? nullptr : (T *)a.out + sample;
            int r = row0 + tr;
            for (; r + (kGnUnroll - 1) * rpp < row1; r += kGnUnroll * rpp) {
                V xv[kGnUnroll], dv[kGnUnroll], rv[kGnUnroll];
#pragma unroll
                for (int u = 0; u < kGnUnroll; ++u) {
                    const size_t off = (size_t)(r + u * rpp) * C + cb;
                    xv[u] = *reinterpret_cast<const V *>(x + off);
                    dv[u] = kBwd ? *reinterpret_cast<const V *>(dy + off) : V{};
                    rv[u] = dxa ? *reinterpret_cast<const V *>(dxa + off) : V{};
                }
#pragma unroll
                for (int u = 0; u < kGnUnroll; ++u) body(xv[u], dv[u], rv[u], kStats ? nullptr : out + (size_t)(r + u * rpp) * C + cb);
            }
            for (; r < row1; r += rpp) {
                const size_t off = (size_t)r * C + cb;
                body(*reinterpret_cast<const V *>(x + off), kBwd ? *reinterpret_cast<const V *>(dy + off) : V{},
                     dxa ? *reinterpret_cast<const V *>(dxa + off) : V{}, kStats ? nullptr : out + off);
            }
            if (kStats) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float w = (MODE == kGnBwdStats) ? k2[i] : 1.0f;        // backward sums carry gamma
                    chan[((size_t)tr * C + cb + i) * 2] = acc1[i] * w;
                    chan[((size_t)tr * C + cb + i) * 2 + 1] = acc2[i] * w;
                }
            }
        }
    }
    if (kStats) {          // channels -> groups: J lanes per group over the group's rpp * cpg channel sums, then lane 0 of the group
        __syncthreads();
        float s = 0.f, q = 0.f;
        if (pg < G) {
            const int E = rpp * cpg;
            for (int e = pj; e < E; e += J) {
                const int r = e / cpg, c = pg * cpg + e % cpg;
                s += chan[((size_t)r * C + c) * 2];
                q += chan[((size_t)r * C + c) * 2 + 1];
            }
        }
        red[2 * t] = s;
        red[2 * t + 1] = q;
        __syncthreads();
        if (t < G) {
            s = 0.f; q = 0.f;
            for (int j = 0; j < J; ++j) { s += red[2 * (t * J + j)]; q += red[2 * (t * J + j) + 1]; }
            float *p = a.partial + (((size_t)n * a.splits + split) * G + t) * 2;
            p[0] = s;
            p[1] = q;
        }
    }
}

// ---------------------------------------------------------------------------------------- forward, slab-resident (ONE launch)
// The UNet's activations at batch 8 are 0.3 .. 10 MB: both launches above are at the launch floor (5-8 us each, 61 GroupNorms per
// forward = 0.9 ms of the 6.5 ms UNet).  Here a workgroup owns ALL pixels of one sample for a block of `gb` consecutive groups
// (gb cpg channels = `ppp` 16-byte pieces per pixel; gb = the smallest count that makes the block a whole number of pieces), keeps
// its slab in REGISTERS (<= kSlabPPT pieces per thread), and does statistics, normalisation, SiLU and the store from them: one
// read, one write, one launch.  The slab's pieces are dealt piece-fastest (consecutive lanes read consecutive pieces of a pixel)
// with a thread count that is a multiple of ppp, so a thread keeps ONE channel column and its constants for all its pixels.
// Sums: shifted like the two-pass kernel's (x + add - K, K = the group's first element), per thread and channel, then through
// LDS in a fixed order (deterministic).  Chosen by gn_slab_plan when the slab fits; everything else takes the two launches.
constexpr int kSlabPPT = 24;
constexpr int kSlabThreads = 256;
__global__ __launch_bounds__(kSlabThreads) void k_groupnorm_slab_f16(GnArgs a, int gb, int ppp)
{
    typedef _Float16 V __attribute__((ext_vector_type(8)));
    __shared__ float part[kSlabThreads][17];          // per thread: 8 channel sums, 8 channel sums of squares (+1: bank spread)
    __shared__ float red[kSlabThreads][2];
    __shared__ float gstat[16][2];                    // mean, rstd of the block's groups
    const int n = blockIdx.y, t = threadIdx.x, nt = blockDim.x;
    const int C = a.C, cpg = C / a.G, g0 = blockIdx.x * gb, c0 = g0 * cpg, nch = gb * cpg;
    const int piece = t % ppp, R = nt / ppp;          // rows (pixels) in flight; this thread: pixels t / ppp, + R, ...
    const size_t sample = (size_t)n * a.HW * C;
    const _Float16 *__restrict__ x = (const _Float16 *)a.x + sample + c0 + 8 * piece;
    const _Float16 *__restrict__ add = a.add ? (const _Float16 *)a.add + (size_t)n * a.add_stride + c0 + 8 * piece : nullptr;
    V v[kSlabPPT];
    const int p0 = t / ppp;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < kSlabPPT; ++k) {
        const int px = p0 + k * R;
        if (px < a.HW) { v[k] = *reinterpret_cast<const V *>(x + (size_t)px * C); cnt = k + 1; }
    }
    float ad[8], sh[8];
    {
        V av = V{};
        if (add) av = *reinterpret_cast<const V *>(add);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ad[i] = (float)av[i];
            const int g = (8 * piece + i) / cpg;                                   // group within the block
            sh[i] = ad[i] - (float)((const _Float16 *)a.x)[sample + (size_t)(g0 + g) * cpg];
        }
    }
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
#pragma unroll
    for (int k = 0; k < kSlabPPT; ++k) {
        if (k < cnt) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float w = (float)v[k][i] + sh[i];
                s1[i] += w;
                s2[i] = __builtin_fmaf(w, w, s2[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { part[t][i] = s1[i]; part[t][8 + i] = s2[i]; }
    __syncthreads();
    // channels -> J lanes per channel over the R threads of its piece column -> groups
    const int J = nt / nch > 0 ? nt / nch : 1;
    {
        const int ch = t / J, j = t % J;
        float s = 0.f, q = 0.f;
        if (ch < nch) {
            const int pc = ch >> 3, i = ch & 7;
            for (int r = j; r < R; r += J) { s += part[pc + ppp * r][i]; q += part[pc + ppp * r][8 + i]; }
        }
        red[t][0] = s; red[t][1] = q;
    }
    __syncthreads();
    if (t < gb) {
        float s = 0.f, q = 0.f;
        for (int e = 0; e < cpg * J; ++e) { s += red[t * cpg * J + e][0]; q += red[t * cpg * J + e][1]; }
        const float inv_m = 1.0f / ((float)cpg * (float)a.HW);
        const float dm = s * inv_m;
        const float mean = (float)((const _Float16 *)a.x)[sample + (size_t)(g0 + t) * cpg] + dm;
        const float var = fmaxf(q * inv_m - dm * dm, 0.f);
        const float rstd = 1.0f / sqrtf(var + a.eps);
        gstat[t][0] = mean; gstat[t][1] = rstd;
        a.stats[((size_t)n * a.G + g0 + t) * 2] = mean;
        a.stats[((size_t)n * a.G + g0 + t) * 2 + 1] = rstd;
    }
    __syncthreads();
    float k0[8], k1[8];
    {
        const V gv = *reinterpret_cast<const V *>((const _Float16 *)a.gamma + c0 + 8 * piece);
        const V bv = *reinterpret_cast<const V *>((const _Float16 *)a.beta + c0 + 8 * piece);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int g = (8 * piece + i) / cpg;
            k0[i] = gstat[g][1] * (float)gv[i];
            k1[i] = __builtin_fmaf(ad[i] - gstat[g][0], k0[i], (float)bv[i]);
        }
    }
    _Float16 *__restrict__ out = (_Float16 *)a.out + sample + c0 + 8 * piece;
#pragma unroll
    for (int k = 0; k < kSlabPPT; ++k) {
        if (k < cnt) {
            V o;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float z = __builtin_fmaf((float)v[k][i], k0[i], k1[i]);
                if (a.silu) z = z * gn_sigmoid(z);
                o[i] = (_Float16)z;
            }
            *reinterpret_cast<V *>(out + (size_t)(p0 + k * R) * C) = o;
        }
    }
}

// groups per workgroup, pieces per pixel and threads of the slab kernel for this shape; false: take the two-pass kernels
static bool gn_slab_plan(const GnArgs &a, int &gb, int &ppp, int &threads)
{
    static const int off = [] { const char *e = getenv("DM4D_GN_SLAB"); return e && atoi(e) == 0 ? 1 : 0; }();      // (A/B switch)
    if (off) return false;
    const int cpg = a.C / a.G;
    int g = 8, x = cpg;                 // gb = 8 / gcd(cpg, 8)
    while (x % 2 == 0 && g > 1) { x /= 2; g /= 2; }
    gb = g;
    if (a.G % gb != 0 || gb > 16) return false;
    ppp = gb * cpg / 8;
    if (ppp < 1 || ppp > kSlabThreads / 2) return false;
    threads = kSlabThreads / ppp * ppp;
    if (gb * cpg > threads) return false;                                       // (the channel -> group stage has a lane per channel)
    const int R = threads / ppp;
    // measured per shape (tools/gn_shapes.py): 1.4-2.9x faster than the two launches at <= 16 x 16 pixels per sample (7-9 us against
    // 10-23), 10 % slower at 32 x 32 (8 samples x 8 group blocks = 64 workgroups of 20 pieces per thread) and at the VAE's 64 x 64
    static const int max_hw = [] { const char *e = getenv("DM4D_GN_SLAB_MAX_HW"); return e ? atoi(e) : 256; }();      // (A/B switch)
    return a.HW <= max_hw && (a.HW + R - 1) / R <= kSlabPPT;
}

static size_t gn_lds_bytes(int C, int G, int vec, bool stats)
{
    const int cv = C / vec, cw = cv < kGnThreads ? cv : kGnThreads, rpp = kGnThreads / cw;
    return sizeof(float) * (4 * (size_t)G + 2 * kGnThreads + (stats ? 2 * (size_t)rpp * C : 0));
}

template <typename T>
static int gn_launch(GnArgs a, bool backward, hipStream_t st)
{
    constexpr int VEC = GnVec<T>::n;
    const dim3 grid(a.splits, a.N), block(kGnThreads);
    const size_t lds_s = gn_lds_bytes(a.C, a.G, VEC, true), lds_a = gn_lds_bytes(a.C, a.G, VEC, false);
    if (lds_s > 64 * 1024) { set_error("groupnorm: C = %d needs %zu bytes of LDS", a.C, lds_s); return DM4D_ERR_INVALID; }
    if (!backward) {
        if constexpr (std::is_same<T, _Float16>::value) {
            int gb, ppp, threads;
            if (gn_slab_plan(a, gb, ppp, threads)) {
                hipLaunchKernelGGL(k_groupnorm_slab_f16, dim3(a.G / gb, a.N), dim3(threads), 0, st, a, gb, ppp);
                DM4D_HIP_CHECK(hipGetLastError());
                return DM4D_OK;
            }
        }
        hipLaunchKernelGGL((k_groupnorm<T, kGnFwdStats>), grid, block, lds_s, st, a);
        hipLaunchKernelGGL((k_groupnorm<T, kGnFwdApply>), grid, block, lds_a, st, a);
    } else {
        hipLaunchKernelGGL((k_groupnorm<T, kGnBwdStats>), grid, block, lds_s, st, a);
        hipLaunchKernelGGL((k_groupnorm<T, kGnBwdApply>), grid, block, lds_a, st, a);
    }
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

static int gn_check(int N, int HW, int C, int G, int dtype, int splits, const void *x, const void *gamma, const void *beta,
                    const void *out, const float *stats, const float *scratch)
{
    const int vec = dtype == DM4D_GN_F16 ? 8 : 4;
    if (dtype != DM4D_GN_F16 && dtype != DM4D_GN_F32) { set_error("groupnorm: dtype %d", dtype); return DM4D_ERR_INVALID; }
    if (N < 0 || HW <= 0 || C <= 0 || G <= 0 || G > kGnThreads || C % G || C % vec || splits < 1 || splits > DM4D_GN_MAX_SPLITS) {
        set_error("groupnorm: N %d HW %d C %d G %d splits %d (C must be a multiple of G and of %d)", N, HW, C, G, splits, vec);
        return DM4D_ERR_INVALID;
    }
    if (N && (!x || !gamma || !beta || !out || !stats || !scratch)) { set_error("groupnorm: null pointer"); return DM4D_ERR_INVALID; }
    return DM4D_OK;
}

}  // namespace dm4d

using namespace dm4d;

extern "C" int dm4d_groupnorm_nhwc_forward(int32_t N, int32_t HW, int32_t C, int32_t G, int32_t dtype, const void *x,
                                           const void *add, int32_t add_stride, const void *gamma, const void *beta, float eps,
                                           int32_t silu, void *y, float *stats, float *scratch, int32_t splits, dm4d_stream_t stream)
{
    int rc = gn_check(N, HW, C, G, dtype, splits, x, gamma, beta, y, stats, scratch);
    if (rc != DM4D_OK || N == 0) return rc;
    if (add && add_stride != 0 && (add_stride < C || add_stride % 8 != 0)) { set_error("groupnorm: add_stride must be 0 or >= C and a multiple of 8"); return DM4D_ERR_INVALID; }
    GnArgs a{N, HW, C, G, splits, (HW + splits - 1) / splits, silu, add_stride, eps, x, add, gamma, beta, nullptr, y, stats, scratch, nullptr};
    return dtype == DM4D_GN_F16 ? gn_launch<_Float16>(a, false, (hipStream_t)stream) : gn_launch<float>(a, false, (hipStream_t)stream);
}

extern "C" int dm4d_groupnorm_nhwc_backward_add(int32_t N, int32_t HW, int32_t C, int32_t G, int32_t dtype, const void *x,
                                                const void *add, int32_t add_stride, const void *gamma, const void *beta,
                                                const float *stats, int32_t silu, const void *dy, const void *dx_add, void *dx,
                                                float *scratch, int32_t splits, dm4d_stream_t stream)
{
    int rc = gn_check(N, HW, C, G, dtype, splits, x, gamma, beta, dx, stats, scratch);
    if (rc != DM4D_OK || N == 0) return rc;
    if (!dy) { set_error("groupnorm: null pointer"); return DM4D_ERR_INVALID; }
    if (add && add_stride != 0 && (add_stride < C || add_stride % 8 != 0)) { set_error("groupnorm: add_stride must be 0 or >= C and a multiple of 8"); return DM4D_ERR_INVALID; }
    GnArgs a{N, HW, C, G, splits, (HW + splits - 1) / splits, silu, add_stride, 0.f, x, add, gamma, beta, dy, dx, const_cast<float *>(stats), scratch, dx_add};
    return dtype == DM4D_GN_F16 ? gn_launch<_Float16>(a, true, (hipStream_t)stream) : gn_launch<float>(a, true, (hipStream_t)stream);
}

extern "C" int dm4d_groupnorm_nhwc_backward(int32_t N, int32_t HW, int32_t C, int32_t G, int32_t dtype, const void *x,
                                            const void *add, int32_t add_stride, const void *gamma, const void *beta,
                                            const float *stats, int32_t silu,
                                            const void *dy, void *dx, float *scratch, int32_t splits, dm4d_stream_t stream)
{
    return dm4d_groupnorm_nhwc_backward_add(N, HW, C, G, dtype, x, add, add_stride, gamma, beta, stats, silu, dy, nullptr, dx, scratch, splits, stream);
}
