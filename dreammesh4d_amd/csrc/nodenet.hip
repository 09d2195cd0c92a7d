// nodenet.hip -- the deformation network of the graph nodes as ONE operator: HexPlane query + MLP forward in one launch,
// their backward in three (the separate operators of hexplane.hip / deform_mlp.hip take 2 and 5, with a torch kernel for
// 2 t - 1 in front).  Same kernels' bodies, regrouped:
//
//   forward   k_nodenet_fwd   workgroup = 16 (frame, node) rows: the 16 x 128 feature tile is computed straight into the LDS
//                              tile the MLP's first layer reads (and written once to HBM for the backward), then the MLP
//   backward  k_mlp_bwd       dy_k, dx, dh, d feat                                                         (deform_mlp.hip)
//             k_nodenet_bwd2  independent jobs side by side: per-point plane-product gradients | parameter-gradient tiles of
//                              the MLP (split-K partials) | zero fill of the time planes
//             k_nodenet_bwd3  time-plane columns | spatial texels | reduction of the MLP partials
// Every step of the per-iteration chain is latency-bound at 4000 rows ("the step is the sum of its
// chain"): what this buys is four launches and two dependent round trips, not bandwidth.
//
// Reference: C/geometry/deformation.py:88-305,430-436 queried by C/geometry/dynamic_sugar.py:420-431.
#include <stdlib.h>

#include "hexplane.hip"
#include "deform_mlp.hip"

// Probe builds (tools/build_variant.sh <name> "-DDM4D_PROBE_NODENET=n" nodenet.hip; never the product): 1 = no plane sampling (constant
// features), 2 = no MLP, 3 = no [in][out] weight copies, 4 / 5 / 6 = bwd3 without its time columns / spatial texels / split-K reduction,
// 7 / 8 / 9 = bwd2 without its parameter-gradient tiles / per-point products / zero fill: what each phase of the four launches costs.
#ifndef DM4D_PROBE_NODENET
#define DM4D_PROBE_NODENET 0
#endif

namespace dm4d {

// 1024 threads: the query is a chain of dependent gathers (node -> 24 texels per feature), so the 2048 features of the tile are
// spread over 16 waves (two features per thread; with 4 waves and eight per thread the kernel took 41 us -- more than the two
// operators it replaces); the MLP is the 4-wave code of deform_mlp.hip, waves 4..15 leave after the barrier that publishes
// the tile (a wave that has ended no longer counts at s_barrier).
constexpr int kNodeFwdThreads = 1024;
__global__ __launch_bounds__(kNodeFwdThreads) void k_nodenet_fwd(HexDesc hd, MlpDesc d_, const float *__restrict__ nodes, const float *__restrict__ times,
                                                                 float *__restrict__ feat, float *__restrict__ samples, float *__restrict__ Hs,
                                                                 float *__restrict__ Ys, float *out0, float *out1, float *out2, float *out3)
{
    __shared__ float4 s_f[(kMaxIn / 4) * kXs];          // feat tile   [feature quad][row]
    MlpDesc d = d_;
    const int tid = threadIdx.x, row0 = blockIdx.x * kRT, IN = d.IN;     // IN == hd.S * 32
    float *sf = reinterpret_cast<float *>(s_f);
    // 32 consecutive lanes = the 32 channels of one (row, scale) query: one 128-byte line per texel (channels-last planes)
    for (int e = tid; e < kRT * IN; e += kNodeFwdThreads) {
        const int r = e / IN, col = e % IN, row = row0 + r;
        float v = 0.f;
        if (DM4D_PROBE_NODENET == 1) v = 1.0f + 1e-3f * (float)col;
        else if (row < d.P) v = hex_feature(hd, nodes, times, row / hd.M, row % hd.M, col / kHexCh, col % kHexCh, feat, samples);
        sf[((col >> 2) * kXs + r) * 4 + (col & 3)] = v;
    }
    if (DM4D_PROBE_NODENET == 2) return;
    if (DM4D_PROBE_NODENET == 3) d.W0T = nullptr;
    if (tid >= 256) { __syncthreads(); return; }        // (their share of the barrier inside mlp_fwd_block that publishes s_f)
    mlp_fwd_block(d, (int)blockIdx.x, (int)gridDim.x, s_f, Hs, Ys, out0, out1, out2, out3);
}

struct NodeBwdJobs { unsigned point_blocks, wgrad_tiles, wgrad_blocks, zero_blocks, plane_blocks, reduce_blocks, mlp_blocks; };

// ---- round 5: the backward as TWO launches (the three below stay: DM4D_NODENET_BWD3=1, the A/B switch) ----
//   k_nodenet_bwdA   MLP backward of 16 rows (dy_k, dx, dh) whose dL/dfeat tile goes straight on to the HexPlane's per-point
//                    plane-product gradients G (the lane that holds four consecutive channels of a row reads its six saved samples as
//                    float4s and rewrites them in place: hex_bwd_point's products in its order) | zero fill of the time planes | the
//                    parameter-gradient tiles' arrival tickets cleared
//   k_nodenet_bwdB   parameter-gradient tiles with the row-slice reduction inside the launch (mlp_wgrad_block's last-arriving slice) |
//                    time-plane columns | spatial texels
// Measured on the bench scene (tools/build_variant.sh probes, profiles/r05_nodenet_probe.txt): of the 67 us of k_mlp_bwd (12) ->
// bwd2 (21) -> bwd3 (34) the per-point products and the split-K reduction cost ~4 and ~0 us of kernel time but each holds a launch
// boundary and a dependent round trip on the step's serial chain; bwd3's 34 us are its two gathers (9 + 11) on a ~14 us floor.
struct PointGrad {
    float *G;           // [B][M][S][6][32]: the forward's samples, rewritten in place
    int S;
    __device__ __forceinline__ void operator()(const int row, const int col, const float4 g) const
    {
        const int s = col / kHexCh, c = col % kHexCh;
        float *o = G + (((size_t)row * S + s) * kHexPlanes) * kHexCh + c;
        float4 v[kHexPlanes], pre[kHexPlanes], suf[kHexPlanes];
#pragma unroll
        for (int p = 0; p < kHexPlanes; ++p) v[p] = *reinterpret_cast<const float4 *>(o + (size_t)p * kHexCh);
        pre[0] = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
        for (int p = 1; p < kHexPlanes; ++p)
            pre[p] = make_float4(pre[p - 1].x * v[p - 1].x, pre[p - 1].y * v[p - 1].y, pre[p - 1].z * v[p - 1].z, pre[p - 1].w * v[p - 1].w);
        suf[kHexPlanes - 1] = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
        for (int p = kHexPlanes - 2; p >= 0; --p)
            suf[p] = make_float4(suf[p + 1].x * v[p + 1].x, suf[p + 1].y * v[p + 1].y, suf[p + 1].z * v[p + 1].z, suf[p + 1].w * v[p + 1].w);
#pragma unroll
        for (int p = 0; p < kHexPlanes; ++p)
            *reinterpret_cast<float4 *>(o + (size_t)p * kHexCh) =
                make_float4(g.x * (pre[p].x * suf[p].x), g.y * (pre[p].y * suf[p].y), g.z * (pre[p].z * suf[p].z), g.w * (pre[p].w * suf[p].w));
    }
};

__global__ __launch_bounds__(256) void k_nodenet_bwdA(HexDesc hd, MlpDesc d, NodeBwdJobs jb, float *__restrict__ G, HexGrads hg,
                                                      const float *__restrict__ Hs, const float *g0, const float *g1, const float *g2,
                                                      const float *g3)
{
    unsigned b = blockIdx.x;
    if (b < jb.mlp_blocks) { mlp_bwd_block(d, (int)b, Hs, g0, g1, g2, g3, true, PointGrad{G, hd.S}); return; }
    b -= jb.mlp_blocks;
    if (b < jb.zero_blocks) { hex_zero(hg, b); return; }
    for (int t = threadIdx.x; t < kMaxTickets; t += 256) d.tickets[t] = 0u;
}

__global__ __launch_bounds__(256) void k_nodenet_bwdB(HexDesc hd, MlpDesc d, NodeBwdJobs jb, const float *__restrict__ nodes,
                                                      const float *__restrict__ times, HexPlan pl, const float *__restrict__ G, HexGrads hg,
                                                      MlpGrads mg, const float *__restrict__ feat, const float *__restrict__ Hs,
                                                      const float *__restrict__ Ys, const float *g0, const float *g1, const float *g2,
                                                      const float *g3)
{
    unsigned b = blockIdx.x;
    if (b < jb.wgrad_blocks) {
        // slice-major: the eight slices of a tile are eight blocks apart in launch order only by the tile count, i.e. they start together
        const WgradFinish fin = {d.tickets, &mg};
        mlp_wgrad_block(d, (int)(b % jb.wgrad_tiles), (int)(b / jb.wgrad_tiles), feat, Hs, Ys, g0, g1, g2, g3, DM4D_WGRAD_FINISH ? &fin : nullptr);
        return;
    }
    b -= jb.wgrad_blocks;
    if (b < (unsigned)pl.n_time) {
        hex_bwd_time(hd, b, nodes, times, pl.n_time, pl.tp_scale, pl.tp_plane, pl.tp_col, pl.tp_off, pl.tp_item, G, hg);
        return;
    }
    b -= (unsigned)pl.n_time;
    hex_bwd_spatial(hd, b, nodes, pl.n_spatial, pl.sp_scale, pl.sp_plane, pl.sp_texel, pl.sp_off, pl.sp_item, G, hg);
}

__global__ __launch_bounds__(256) void k_nodenet_bwd2(HexDesc hd, MlpDesc d, NodeBwdJobs jb, const float *__restrict__ g_feat, float *__restrict__ G,
                                                      HexGrads hg, const float *__restrict__ feat, const float *__restrict__ Hs,
                                                      const float *__restrict__ Ys, const float *g0, const float *g1, const float *g2,
                                                      const float *g3)
{
    unsigned b = blockIdx.x;
    if (b < jb.wgrad_blocks) { if (DM4D_PROBE_NODENET != 7) mlp_wgrad_block(d, (int)(b % jb.wgrad_tiles), (int)(b / jb.wgrad_tiles), feat, Hs, Ys, g0, g1, g2, g3); return; }
    b -= jb.wgrad_blocks;
    if (b < jb.point_blocks) { if (DM4D_PROBE_NODENET != 8) hex_bwd_point(hd, b, g_feat, G); return; }
    if (DM4D_PROBE_NODENET != 9) hex_zero(hg, b - jb.point_blocks);
}

__global__ __launch_bounds__(256) void k_nodenet_bwd3(HexDesc hd, MlpDesc d, NodeBwdJobs jb, const float *__restrict__ nodes,
                                                      const float *__restrict__ times, HexPlan pl, const float *__restrict__ G, HexGrads hg,
                                                      MlpGrads mg)
{
    unsigned b = blockIdx.x;
    if (b < (unsigned)pl.n_time) {
        if (DM4D_PROBE_NODENET != 4) hex_bwd_time(hd, b, nodes, times, pl.n_time, pl.tp_scale, pl.tp_plane, pl.tp_col, pl.tp_off, pl.tp_item, G, hg);
        return;
    }
    b -= (unsigned)pl.n_time;
    if (b < jb.plane_blocks) {
        if (DM4D_PROBE_NODENET != 5) hex_bwd_spatial(hd, b, nodes, pl.n_spatial, pl.sp_scale, pl.sp_plane, pl.sp_texel, pl.sp_off, pl.sp_item, G, hg);
        return;
    }
    if (DM4D_PROBE_NODENET != 6) mlp_reduce_block(d, mg, kKSplit, b - jb.plane_blocks);
}

}  // namespace dm4d

using namespace dm4d;

extern "C" {

size_t dm4d_nodenet_scratch_bytes(int32_t S, int32_t M, int32_t B, int32_t n_heads)
{
    return dm4d_deform_mlp_scratch_bytes(B * M, S * kHexCh, n_heads);
}

int dm4d_nodenet_forward(int32_t S, int32_t M, int32_t B, const int32_t *res, const float *const *planes, int32_t flags,
                         const float *aabb_host, const float *nodes, const float *times, const dm4d_mlp_weights *w, float *feat,
                         void *samples, float *h_save, float *y_save, float *const *out, void *scratch, dm4d_stream_t stream)
{
    HexDesc hd;
    int rc = fill_desc(hd, S, M, B, res, planes, aabb_host, flags & DM4D_HEX_CHANNELS_LAST);
    if (rc) return rc;
    hd.t01 = (flags & DM4D_HEX_TIMES_01) ? 1 : 0;
    MlpDesc d;
    const int P = B * M;
    if ((rc = fill_mlp(d, P, w, scratch))) return rc;
    if (w->in_dim != S * kHexCh) { set_error("nodenet: the MLP's in_dim %d != %d scales x 32 channels", w->in_dim, S); return DM4D_ERR_INVALID; }
    if (!planes || !nodes || !times || !feat || !h_save || !y_save || !out || !scratch) { set_error("nodenet: null tensor"); return DM4D_ERR_INVALID; }
    bool al = aligned16(feat) && aligned16(h_save) && aligned16(y_save) && aligned16(scratch) && aligned16(w->W0) && aligned16(w->b0);
    float *o[kMaxHeads] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < d.n_heads; ++k) {
        o[k] = out[k];
        if (!o[k]) { set_error("nodenet: null output %d", k); return DM4D_ERR_INVALID; }
        al = al && aligned16(w->W1[k]) && aligned16(w->b1[k]) && aligned16(w->W2[k]);
    }
    if (!al) { set_error("nodenet: tensors must be 16-byte aligned"); return DM4D_ERR_INVALID; }
    hipLaunchKernelGGL(k_nodenet_fwd, dim3((P + kRT - 1) / kRT), dim3(kNodeFwdThreads), 0, (hipStream_t)stream, hd, d, nodes, times, feat,
                       (float *)samples, h_save, y_save, o[0], o[1], o[2], o[3]);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_nodenet_backward(int32_t S, int32_t M, int32_t B, const int32_t *res, const float *const *planes, int32_t flags,
                          const float *aabb_host, const float *nodes, const float *times, const dm4d_mlp_weights *w,
                          const float *feat, void *samples, const float *h_save, const float *y_save, const float *const *g_out,
                          int32_t n_spatial, const int32_t *sp_scale, const int32_t *sp_plane, const int32_t *sp_texel,
                          const int32_t *sp_off, const int32_t *sp_item, int32_t n_time, const int32_t *tp_scale,
                          const int32_t *tp_plane, const int32_t *tp_col, const int32_t *tp_off, const int32_t *tp_item,
                          float *g_feat, float *const *g_planes, const dm4d_mlp_weights_grad *gw, void *scratch, dm4d_stream_t stream)
{
    HexDesc hd;
    int rc = fill_desc(hd, S, M, B, res, planes, aabb_host, flags & DM4D_HEX_CHANNELS_LAST);
    if (rc) return rc;
    hd.t01 = (flags & DM4D_HEX_TIMES_01) ? 1 : 0;
    MlpDesc d;
    const int P = B * M;
    if ((rc = fill_mlp(d, P, w, scratch))) return rc;
    if (!planes || !nodes || !times || !feat || !samples || !h_save || !y_save || !g_out || !g_feat || !g_planes || !gw || !scratch) {
        set_error("nodenet: null tensor");
        return DM4D_ERR_INVALID;
    }
    if (!(aligned16(feat) && aligned16(h_save) && aligned16(y_save) && aligned16(g_feat) && aligned16(scratch))) {
        set_error("nodenet: tensors must be 16-byte aligned");
        return DM4D_ERR_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    const float *g[kMaxHeads] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < d.n_heads; ++k) g[k] = g_out[k];
    MlpGrads mg;
    memset(&mg, 0, sizeof(mg));
    mg.W0 = gw->W0; mg.b0 = gw->b0;
    for (int k = 0; k < d.n_heads; ++k) { mg.W1[k] = gw->W1[k]; mg.b1[k] = gw->b1[k]; mg.W2[k] = gw->W2[k]; mg.b2[k] = gw->b2[k]; }
    HexGrads hg;
    memset(&hg, 0, sizeof(hg));
    hg.n = S * kHexPlanes;
    unsigned long long run = 0;
    for (int s = 0; s < S; ++s)
        for (int p = 0; p < kHexPlanes; ++p) {
            const int k = s * kHexPlanes + p;
            hg.g[k] = g_planes[k];
            if (!hg.g[k]) { set_error("nodenet: null gradient plane %d", k); return DM4D_ERR_INVALID; }
            run += (size_t)kHexCh * hd.res[s][c_axis0_host[p]] * hd.res[s][c_axis1_host[p]] / 4;
            hg.end4[k] = run;
        }
    if (flags & DM4D_HEX_KEEP_SPATIAL)
        for (int s = 0; s < S; ++s)
            for (int p : {0, 1, 3}) hg.keep_mask |= 1ull << (s * kHexPlanes + p);
    static const bool three_launches = getenv("DM4D_NODENET_BWD3") && atoi(getenv("DM4D_NODENET_BWD3")) != 0;      // (A/B switch: rounds 3-4's chain)
    if (!three_launches) {
        NodeBwdJobs jb;
        memset(&jb, 0, sizeof(jb));
        jb.mlp_blocks = (unsigned)((P + kRT - 1) / kRT);
        jb.zero_blocks = kHexZeroBlocks * (unsigned)hg.n;
        jb.wgrad_tiles = (unsigned)(4 * (d.IN / 16 + 1) + 25 * d.n_heads);
        jb.wgrad_blocks = jb.wgrad_tiles * kKSplit;
        jb.plane_blocks = (unsigned)(((size_t)(n_spatial > 0 ? n_spatial : 0) * kHexCh + 255) / 256);
        if ((int)jb.wgrad_tiles > kMaxTickets) { set_error("nodenet: %u parameter-gradient tiles", jb.wgrad_tiles); return DM4D_ERR_UNSUPPORTED; }
        hipLaunchKernelGGL(k_nodenet_bwdA, dim3(jb.mlp_blocks + jb.zero_blocks + 1), dim3(256), 0, st, hd, d, jb, (float *)samples, hg, h_save,
                           g[0], g[1], g[2], g[3]);
        DM4D_HIP_CHECK(hipGetLastError());
        HexPlan pl = {n_spatial, n_time < 0 ? 0 : n_time, sp_scale, sp_plane, sp_texel, sp_off, sp_item, tp_scale, tp_plane, tp_col, tp_off, tp_item};
        hipLaunchKernelGGL(k_nodenet_bwdB, dim3(jb.wgrad_blocks + (unsigned)pl.n_time + jb.plane_blocks), dim3(256), 0, st, hd, d, jb, nodes, times,
                           pl, (const float *)samples, hg, mg, feat, h_save, y_save, g[0], g[1], g[2], g[3]);
        DM4D_HIP_CHECK(hipGetLastError());
        if (!DM4D_WGRAD_FINISH) {
            const size_t np = partial_floats(d);
            hipLaunchKernelGGL(k_mlp_reduce, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, d, mg, kKSplit);
            DM4D_HIP_CHECK(hipGetLastError());
        }
        return DM4D_OK;
    }
    // 1: dL/dy_k, dL/dh, dL/dfeat
    hipLaunchKernelGGL(k_mlp_bwd, dim3((P + kRT - 1) / kRT), dim3(256), 0, st, d, h_save, g[0], g[1], g[2], g[3], g_feat);
    DM4D_HIP_CHECK(hipGetLastError());
    // 2: MLP parameter-gradient tiles | per-point plane products | zero fill
    NodeBwdJobs jb;
    jb.wgrad_tiles = (unsigned)(4 * (d.IN / 16 + 1) + 25 * d.n_heads);
    jb.wgrad_blocks = jb.wgrad_tiles * kKSplit;
    jb.point_blocks = (unsigned)(((size_t)B * M * S * kHexCh + 255) / 256);
    jb.zero_blocks = kHexZeroBlocks * (unsigned)hg.n;
    jb.plane_blocks = (unsigned)(((size_t)(n_spatial > 0 ? n_spatial : 0) * kHexCh + 255) / 256);
    jb.reduce_blocks = (unsigned)((partial_floats(d) + 255) / 256);
    hipLaunchKernelGGL(k_nodenet_bwd2, dim3(jb.wgrad_blocks + jb.point_blocks + jb.zero_blocks), dim3(256), 0, st, hd, d, jb, g_feat,
                       (float *)samples, hg, feat, h_save, y_save, g[0], g[1], g[2], g[3]);
    DM4D_HIP_CHECK(hipGetLastError());
    // 3: the planes' gathers | the reduction of the MLP partials
    HexPlan pl = {n_spatial, n_time < 0 ? 0 : n_time, sp_scale, sp_plane, sp_texel, sp_off, sp_item, tp_scale, tp_plane, tp_col, tp_off, tp_item};
    hipLaunchKernelGGL(k_nodenet_bwd3, dim3((unsigned)pl.n_time + jb.plane_blocks + jb.reduce_blocks), dim3(256), 0, st, hd, d, jb, nodes, times,
                       pl, (const float *)samples, hg, mg);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // extern "C"
