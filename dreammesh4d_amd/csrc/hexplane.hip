// hexplane.hip -- fused multi-scale HexPlane feature query (forward + grid gradients) for the
// deformation-graph nodes (gfx950).  Replaces, per training step, the 24 F.grid_sample launches
// (+ 24 atomics-based grid_sampler backward launches, 2.5 ms/step measured through PyTorch-ROCm) of
//   custom/threestudio-dreammesh4d/geometry/deformation.py:88-113 (grid_sample_wrapper),
//   :141-174 (interpolate_ms_features), :226-240 (HexPlaneField.get_density)
// as queried by DynamicSuGaRModel._get_timed_dg_attributes (geometry/dynamic_sugar.py:420-431):
// M static graph nodes x B timestamps per step.
//
//   feat[f, m, s*32 + c] = prod_{p in 6 planes} bilinear(plane[s][p][c], coords(m, t_f))
// (align_corners=True, padding_mode='border', aabb = [[+b],[-b]] so x -> -x/b).
//
// Structure exploited: node positions are static, only the time coordinate changes.  The backward
// therefore never scatters with atomics: a plan built once per node set lists, for every touched
// texel (spatial planes) or touched column (time planes), the (node, corner) pairs that reach it, and
// the gradient kernels GATHER over those lists in fixed order and write each touched texel exactly
// once -- deterministic, no float atomics.  The caller supplies zero-filled dense gradient planes
// (parameter layout [1, 32, H, W] is kept: checkpoints of the reference load unchanged).
#include <string.h>
#include "common.h"
#include "raster.h"

namespace dm4d {

constexpr int kHexCh = 32;        // output_coordinate_dim (deformation.py:64)
constexpr int kHexPlanes = 6;     // (x,y) (x,z) (x,t) (y,z) (y,t) (z,t)
constexpr int kHexMaxScales = 8;
constexpr int kHexMaxFrames = 16;   // frames (distinct timestamps) per call
__constant__ int c_axis0[6] = {0, 0, 0, 1, 1, 2};
__constant__ int c_axis1[6] = {1, 2, 3, 2, 3, 3};
static const int c_axis0_host[6] = {0, 0, 0, 1, 1, 2}, c_axis1_host[6] = {1, 2, 3, 2, 3, 3};

struct HexDesc {
    int S, M, B;
    int cl;                                       // plane storage: 0 = [32][H][W], 1 = [H][W][32] (channels-last)
    int t01;                                      // times are timestamps in [0, 1]; the kernels map them to 2 t - 1
    int res[kHexMaxScales][4];                    // resolution of axes x, y, z, t at scale s
    const float *plane[kHexMaxScales][kHexPlanes]; // [32][res[a1]][res[a0]]
    float lo[3], inv[3];                          // x_n = (p - lo) * inv - 1
};
// gradient planes of one backward call, passed by value (no device-side pointer table to upload)
struct HexGrads {
    float *g[kHexMaxScales * kHexPlanes];
    unsigned long long end4[kHexMaxScales * kHexPlanes];   // running end (in float4) of the planes, for the zero fill
    int n;
    unsigned long long keep_mask;                          // planes the zero fill leaves alone (DM4D_HEX_KEEP_SPATIAL)
};

// grid_sample coordinate (align_corners=True, border padding): index of the lower texel and the
// weight of the upper one.  i0 + 1 may equal n (then its weight is exactly 0 and it is skipped).
__device__ __forceinline__ void texel_coord(float xn, int n, int &i0, float &w1)
{
    float ix = ((xn + 1.f) * 0.5f) * (float)(n - 1);
    ix = fminf((float)(n - 1), fmaxf(ix, 0.f));
    const float fl = floorf(ix);
    i0 = (int)fl;
    w1 = ix - fl;
}

struct Sample { int i00, i01, i10, i11; float w00, w01, w10, w11; };   // (row, col): 0 = lower, 1 = upper
__device__ __forceinline__ Sample plane_sample(const HexDesc &d, int s, int p, const float xn[4])
{
    const int a0 = c_axis0[p], a1 = c_axis1[p];
    const int W = d.res[s][a0], H = d.res[s][a1];
    int x0, y0;
    float wx, wy;
    texel_coord(xn[a0], W, x0, wx);
    texel_coord(xn[a1], H, y0, wy);
    const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);   // clamped duplicates carry weight 0
    Sample q;
    q.i00 = y0 * W + x0; q.i01 = y0 * W + x1; q.i10 = y1 * W + x0; q.i11 = y1 * W + x1;
    q.w00 = (1.f - wx) * (1.f - wy); q.w01 = wx * (1.f - wy); q.w10 = (1.f - wx) * wy; q.w11 = wx * wy;
    return q;
}
// element (channel c, texel) of a plane with HW texels
__device__ __forceinline__ size_t plane_elem(const HexDesc &d, int c, size_t HW, size_t texel)
{
    return d.cl ? texel * kHexCh + c : (size_t)c * HW + texel;
}
// base = offset of the channel, ts = stride between texels (channels-last: the 32 channels of a texel are one
// 128-byte line, so the 32 lanes of a query read 4 lines per plane instead of 128 scattered words)
__device__ __forceinline__ float sample_value(const float *__restrict__ pl, size_t base, size_t ts, const Sample &q)
{
    return ((pl[base + q.i00 * ts] * q.w00 + pl[base + q.i01 * ts] * q.w01) + pl[base + q.i10 * ts] * q.w10) +
           pl[base + q.i11 * ts] * q.w11;
}
__device__ __forceinline__ void node_coords(const HexDesc &d, const float *__restrict__ nodes,
                                            const float *__restrict__ times, int f, int m, float xn[4])
{
#pragma unroll
    for (int a = 0; a < 3; ++a) xn[a] = (nodes[3 * (size_t)m + a] - d.lo[a]) * d.inv[a] - 1.0f;
    xn[3] = d.t01 ? times[f] * 2.0f - 1.0f : times[f];      // DM4D_HEX_TIMES_01: 2 t - 1 (dynamic_sugar.py:431), exactly as torch rounds it
}

// ---------------------------------------------------------------------------------------- forward
// feature (frame f, node m, scale s, channel c): product of the six plane samples; the samples are kept for the backward
__device__ __forceinline__ float hex_feature(const HexDesc &d, const float *__restrict__ nodes, const float *__restrict__ times,
                                             const int f, const int m, const int s, const int c, float *__restrict__ feat,
                                             float *__restrict__ samples)
{
    float xn[4];
    node_coords(d, nodes, times, f, m, xn);
    float acc = 1.f;
    float *sv = samples ? samples + ((((size_t)f * d.M + m) * d.S + s) * kHexPlanes) * kHexCh + c : nullptr;
#pragma unroll
    for (int p = 0; p < kHexPlanes; ++p) {
        const Sample q = plane_sample(d, s, p, xn);
        const size_t HW = (size_t)d.res[s][c_axis0[p]] * d.res[s][c_axis1[p]];
        const float v = sample_value(d.plane[s][p], d.cl ? (size_t)c : (size_t)c * HW, d.cl ? (size_t)kHexCh : (size_t)1, q);
        if (sv) sv[(size_t)p * kHexCh] = v;      // kept for the backward (the channel-major planes make every
        acc = acc * v;                           // sample 4 scattered 4-byte reads: not worth repeating)
    }
    feat[((size_t)f * d.M + m) * (d.S * kHexCh) + s * kHexCh + c] = acc;
    return acc;
}
// one thread per (frame, node, scale, channel); 32 consecutive lanes = the 32 channels of one query
__global__ __launch_bounds__(256) void k_hex_fwd(HexDesc d, const float *__restrict__ nodes,
                                                 const float *__restrict__ times, float *__restrict__ feat,
                                                 float *__restrict__ samples /* [B][M][S][6][32] or nullptr */)
{
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)d.B * d.M * d.S * kHexCh;
    if (gid >= total) return;
    const int c = (int)(gid % kHexCh);
    const int s = (int)((gid / kHexCh) % d.S);
    const int m = (int)((gid / ((size_t)kHexCh * d.S)) % d.M);
    const int f = (int)(gid / ((size_t)kHexCh * d.S * d.M));
    hex_feature(d, nodes, times, f, m, s, c, feat, samples);
}

// ---------------------------------------------------------------------------------------- backward 1
// G[f][m][s][p][c] = dL/dfeat * prod_{p' != p} sample_{p'}, computed IN PLACE over the samples the forward saved
__device__ __forceinline__ void hex_bwd_point(const HexDesc &d, const unsigned bid, const float *__restrict__ g_feat, float *__restrict__ G)
{
    const size_t gid = (size_t)bid * 256 + threadIdx.x;
    const size_t total = (size_t)d.B * d.M * d.S * kHexCh;
    if (gid >= total) return;
    const int c = (int)(gid % kHexCh);
    const int s = (int)((gid / kHexCh) % d.S);
    const int m = (int)((gid / ((size_t)kHexCh * d.S)) % d.M);
    const int f = (int)(gid / ((size_t)kHexCh * d.S * d.M));
    float *o = G + ((((size_t)f * d.M + m) * d.S + s) * kHexPlanes) * kHexCh + c;
    float v[kHexPlanes];
#pragma unroll
    for (int p = 0; p < kHexPlanes; ++p) v[p] = o[(size_t)p * kHexCh];
    const float g = g_feat[((size_t)f * d.M + m) * (d.S * kHexCh) + s * kHexCh + c];
    float pre[kHexPlanes], suf[kHexPlanes];
    pre[0] = 1.f;
#pragma unroll
    for (int p = 1; p < kHexPlanes; ++p) pre[p] = pre[p - 1] * v[p - 1];
    suf[kHexPlanes - 1] = 1.f;
#pragma unroll
    for (int p = kHexPlanes - 2; p >= 0; --p) suf[p] = suf[p + 1] * v[p + 1];
#pragma unroll
    for (int p = 0; p < kHexPlanes; ++p) o[(size_t)p * kHexCh] = g * (pre[p] * suf[p]);
}

// ---------------------------------------------------------------------------------------- backward 2
// Spatial planes: one thread per (touched texel, channel).  Plan arrays (built once per node set):
//   sp_scale[u], sp_plane[u], sp_texel[u]  for the U touched texels
//   sp_off[U+1], sp_item[]                 item = node * 4 + corner  (corner = 2*row + col)
__device__ __forceinline__ void hex_bwd_spatial(const HexDesc &d, const unsigned bid, const float *__restrict__ nodes, int U,
                                                const int32_t *__restrict__ sp_scale,
                                                const int32_t *__restrict__ sp_plane,
                                                const int32_t *__restrict__ sp_texel,
                                                const int32_t *__restrict__ sp_off,
                                                const int32_t *__restrict__ sp_item,
                                                const float *__restrict__ G, const HexGrads &hg)
{
    const size_t gid = (size_t)bid * 256 + threadIdx.x;
    if (gid >= (size_t)U * kHexCh) return;
    const int c = (int)(gid % kHexCh), u = (int)(gid / kHexCh);
    const int s = sp_scale[u], p = sp_plane[u];
    float acc = 0.f;
    for (int e = sp_off[u]; e < sp_off[u + 1]; ++e) {
        const int m = sp_item[e] >> 2, corner = sp_item[e] & 3;
        float xn[4];
#pragma unroll
        for (int a = 0; a < 3; ++a) xn[a] = (nodes[3 * (size_t)m + a] - d.lo[a]) * d.inv[a] - 1.0f;
        xn[3] = 0.f;
        const Sample q = plane_sample(d, s, p, xn);
        const float w = corner == 0 ? q.w00 : corner == 1 ? q.w01 : corner == 2 ? q.w10 : q.w11;
        // the frames' values are REQUESTED together and added in frame order afterwards: as a loop with a run-time trip count this
        // was one global round trip per frame, one after the other (the compiler issues a load, waits, adds) -- the whole cost of
        // this job (profiles/r05_nodenet_probe.txt: 11 us with the gathers, 0 without)
        float gv[kHexMaxFrames];
#pragma unroll
        for (int f = 0; f < kHexMaxFrames; ++f)
            gv[f] = f < d.B ? G[((((size_t)f * d.M + m) * d.S + s) * kHexPlanes + p) * kHexCh + c] : 0.f;
        float gs = 0.f;
#pragma unroll
        for (int f = 0; f < kHexMaxFrames; ++f)
            if (f < d.B) gs += gv[f];
        acc += w * gs;
    }
    const size_t HW = (size_t)d.res[s][c_axis0[p]] * d.res[s][c_axis1[p]];
    hg.g[s * kHexPlanes + p][plane_elem(d, c, HW, (size_t)sp_texel[u])] = acc;
}

// Time planes: one workgroup per touched column.  A column's gradient at time row y is
//   sum_f w(y, f) * sum_items wcol(item) * G[f][node(item)]
// -- the inner sum does not depend on the row, so a thread (channel, item lane q) accumulates ONE value per frame
// over the items e = e0 + q, e0 + q + 8, ... (every G element is read once), the 8 item lanes are added in lane
// order, and the column's distinct time rows (<= 2 B: the two time texels of every frame, merged in a fixed
// order) are then combined from the per-frame sums.  Deterministic.
//   tp_scale[u], tp_plane[u], tp_col[u]; tp_off[U+1], tp_item[] = node * 2 + corner (column corner)
__device__ __forceinline__ void hex_bwd_time(const HexDesc &d, const unsigned bid, const float *__restrict__ nodes,
                                             const float *__restrict__ times, int U,
                                             const int32_t *__restrict__ tp_scale,
                                             const int32_t *__restrict__ tp_plane,
                                             const int32_t *__restrict__ tp_col,
                                             const int32_t *__restrict__ tp_off,
                                             const int32_t *__restrict__ tp_item,
                                             const float *__restrict__ G, const HexGrads &hg)
{
    __shared__ int s_rows[2 * kHexMaxFrames], s_r0[kHexMaxFrames], s_r1[kHexMaxFrames], s_nrows;
    __shared__ float s_wy[kHexMaxFrames];
    __shared__ float s_part[8][kHexMaxFrames][kHexCh];
    const int tid = threadIdx.x;
    const int u = (int)bid;
    const int s = tp_scale[u], p = tp_plane[u];
    const int a0 = c_axis0[p];
    const int W = d.res[s][a0], H = d.res[s][3];
    if (tid == 0) {
        int nrows = 0;
        for (int f = 0; f < d.B; ++f) {
            int y0;
            float wy;
            texel_coord(d.t01 ? times[f] * 2.0f - 1.0f : times[f], H, y0, wy);
            const int yy[2] = {y0, min(y0 + 1, H - 1)};
            int slot[2];
            for (int k = 0; k < 2; ++k) {
                int r = 0;
                while (r < nrows && s_rows[r] != yy[k]) ++r;
                if (r == nrows) s_rows[nrows++] = yy[k];
                slot[k] = r;
            }
            s_r0[f] = slot[0];
            s_r1[f] = slot[1];
            s_wy[f] = wy;
        }
        s_nrows = nrows;
    }
    const int c = tid & (kHexCh - 1), q = tid >> 5;
    float acc[kHexMaxFrames];
#pragma unroll
    for (int f = 0; f < kHexMaxFrames; ++f) acc[f] = 0.f;
    for (int e = tp_off[u] + q; e < tp_off[u + 1]; e += 8) {
        const int m = tp_item[e] >> 1, corner = tp_item[e] & 1;
        const float xa = (nodes[3 * (size_t)m + a0] - d.lo[a0]) * d.inv[a0] - 1.0f;
        int x0;
        float wx;
        texel_coord(xa, W, x0, wx);
        const float wcol = corner ? wx : (1.f - wx);
        const float *__restrict__ Gm = G + (((size_t)m * d.S + s) * kHexPlanes + p) * kHexCh + c;
        const size_t fstride = (size_t)d.M * d.S * kHexPlanes * kHexCh;
#pragma unroll
        for (int f = 0; f < kHexMaxFrames; ++f)
            if (f < d.B) acc[f] += Gm[f * fstride] * wcol;
    }
#pragma unroll
    for (int f = 0; f < kHexMaxFrames; ++f)
        if (f < d.B) s_part[q][f][c] = acc[f];
    __syncthreads();
    // thread (c, q): time-row slots q, q + 8, ...
    const size_t HW = (size_t)W * H;
    for (int r = q; r < s_nrows; r += 8) {
        float t = 0.f;
        for (int f = 0; f < d.B; ++f) {
            float wf = 0.f;
            if (s_r0[f] == r) wf += 1.f - s_wy[f];
            if (s_r1[f] == r) wf += s_wy[f];     // both texels may coincide at the border
            if (wf != 0.f) {
                float sf = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) sf += s_part[k][f][c];
                t += sf * wf;
            }
        }
        hg.g[s * kHexPlanes + p][plane_elem(d, c, HW, (size_t)s_rows[r] * W + tp_col[u])] = t;
    }
}

// zero fill of all gradient planes in one launch: blockIdx.y = plane (float4 granularity; plane sizes are
// multiples of 4 floats)
constexpr unsigned kHexZeroBlocks = 256;   // per plane
__device__ __forceinline__ void hex_zero(const HexGrads &hg, const unsigned bid)
{
    const int k = (int)(bid / kHexZeroBlocks);
    if ((hg.keep_mask >> k) & 1u) return;      // a spatial plane the caller keeps zero outside its touched texels
    const unsigned bx = bid % kHexZeroBlocks;
    const unsigned long long n4 = hg.end4[k] - (k ? hg.end4[k - 1] : 0ull);
    float4 *dst = reinterpret_cast<float4 *>(hg.g[k]);
    for (unsigned long long i = (unsigned long long)bx * 256 + threadIdx.x; i < n4; i += (unsigned long long)kHexZeroBlocks * 256)
        dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// The backward is a chain of small, latency-bound launches on M = 1000 nodes; independent ones share a launch
// (different blocks do different jobs, side by side) instead of queueing behind each other:
//   launch A: the per-point products (blocks [0, point_blocks)) | the zero fill of the gradient planes (the rest)
//   launch B: the time planes' columns (blocks [0, n_time): the longer job first) | the spatial planes' texels
__global__ __launch_bounds__(256) void k_hex_bwd_point_zero(HexDesc d, const float *__restrict__ g_feat, float *__restrict__ G,
                                                            HexGrads hg, unsigned point_blocks)
{
    if (blockIdx.x < point_blocks) hex_bwd_point(d, blockIdx.x, g_feat, G);
    else hex_zero(hg, blockIdx.x - point_blocks);
}

struct HexPlan {
    int n_spatial, n_time;
    const int32_t *sp_scale, *sp_plane, *sp_texel, *sp_off, *sp_item;
    const int32_t *tp_scale, *tp_plane, *tp_col, *tp_off, *tp_item;
};
__global__ __launch_bounds__(256) void k_hex_bwd_planes(HexDesc d, const float *__restrict__ nodes, const float *__restrict__ times,
                                                        HexPlan pl, const float *__restrict__ G, HexGrads hg)
{
    if (blockIdx.x < (unsigned)pl.n_time)
        hex_bwd_time(d, blockIdx.x, nodes, times, pl.n_time, pl.tp_scale, pl.tp_plane, pl.tp_col, pl.tp_off, pl.tp_item, G, hg);
    else
        hex_bwd_spatial(d, blockIdx.x - (unsigned)pl.n_time, nodes, pl.n_spatial, pl.sp_scale, pl.sp_plane, pl.sp_texel, pl.sp_off,
                        pl.sp_item, G, hg);
}

// ---------------------------------------------------------------------------------------- plan helper
// per (scale, axis, node): lower texel index along that axis (device arithmetic == the kernels')
__global__ void k_hex_axis_index(HexDesc d, const float *__restrict__ nodes, int32_t *__restrict__ i0 /* [S][3][M] */)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= d.S * 3 * d.M) return;
    const int m = gid % d.M, a = (gid / d.M) % 3, s = gid / (3 * d.M);
    const float xn = (nodes[3 * (size_t)m + a] - d.lo[a]) * d.inv[a] - 1.0f;
    int x0;
    float w;
    texel_coord(xn, d.res[s][a], x0, w);
    i0[gid] = x0;
}

static int fill_desc(HexDesc &d, int S, int M, int B, const int32_t *res, const float *const *planes, const float *aabb,
                     int channel_last = 0)
{
    if (S <= 0 || S > kHexMaxScales || M <= 0 || B <= 0 || B > kHexMaxFrames) {
        set_error("hexplane: bad S/M/B (%d/%d/%d; S <= %d, B <= %d)", S, M, B, kHexMaxScales, kHexMaxFrames);
        return DM4D_ERR_INVALID;
    }
    if (!res || !aabb) { set_error("hexplane: null res/aabb"); return DM4D_ERR_INVALID; }
    memset(&d, 0, sizeof(d));
    d.S = S; d.M = M; d.B = B;
    d.cl = channel_last ? 1 : 0;
    for (int s = 0; s < S; ++s) {
        for (int a = 0; a < 4; ++a) {
            d.res[s][a] = res[s * 4 + a];
            if (d.res[s][a] < 2) { set_error("hexplane: resolution < 2"); return DM4D_ERR_INVALID; }
        }
        for (int p = 0; p < kHexPlanes; ++p) d.plane[s][p] = planes ? planes[s * kHexPlanes + p] : nullptr;
    }
    for (int a = 0; a < 3; ++a) {
        d.lo[a] = aabb[a];
        d.inv[a] = 2.0f / (aabb[3 + a] - aabb[a]);
    }
    return DM4D_OK;
}

}  // namespace dm4d

using namespace dm4d;

extern "C" {

int dm4d_hexplane_axis_index(int32_t S, int32_t M, const int32_t *res, const float *aabb_host, const float *nodes,
                             int32_t *i0, dm4d_stream_t stream)
{
    HexDesc d;
    int rc = fill_desc(d, S, M, 1, res, nullptr, aabb_host);
    if (rc) return rc;
    if (!nodes || !i0) { set_error("hexplane: null nodes/i0"); return DM4D_ERR_INVALID; }
    hipLaunchKernelGGL(k_hex_axis_index, dim3((S * 3 * M + 255) / 256), dim3(256), 0, (hipStream_t)stream, d, nodes, i0);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_hexplane_forward(int32_t S, int32_t M, int32_t B, const int32_t *res, const float *const *planes,
                          int32_t channel_last, const float *aabb_host, const float *nodes, const float *times,
                          float *feat, void *samples, dm4d_stream_t stream)
{
    HexDesc d;
    int rc = fill_desc(d, S, M, B, res, planes, aabb_host, channel_last & DM4D_HEX_CHANNELS_LAST);
    if (rc) return rc;
    d.t01 = (channel_last & DM4D_HEX_TIMES_01) ? 1 : 0;
    if (!planes || !nodes || !times || !feat) { set_error("hexplane: null tensor"); return DM4D_ERR_INVALID; }
    const size_t total = (size_t)B * M * S * kHexCh;
    hipLaunchKernelGGL(k_hex_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d, nodes, times, feat,
                       (float *)samples);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

size_t dm4d_hexplane_scratch_bytes(int32_t S, int32_t M, int32_t B) { return (size_t)B * M * S * kHexPlanes * kHexCh * 4; }

int dm4d_hexplane_backward(int32_t S, int32_t M, int32_t B, const int32_t *res, const float *const *planes,
                           int32_t channel_last, const float *aabb_host, const float *nodes, const float *times,
                           const float *g_feat,
                           int32_t n_spatial, const int32_t *sp_scale, const int32_t *sp_plane, const int32_t *sp_texel,
                           const int32_t *sp_off, const int32_t *sp_item, int32_t n_time, const int32_t *tp_scale,
                           const int32_t *tp_plane, const int32_t *tp_col, const int32_t *tp_off, const int32_t *tp_item,
                           void *scratch, float *const *g_planes, dm4d_stream_t stream)
{
    HexDesc d;
    int rc = fill_desc(d, S, M, B, res, planes, aabb_host, channel_last & DM4D_HEX_CHANNELS_LAST);
    if (rc) return rc;
    d.t01 = (channel_last & DM4D_HEX_TIMES_01) ? 1 : 0;
    if (!planes || !nodes || !times || !g_feat || !scratch || !g_planes) { set_error("hexplane: null tensor"); return DM4D_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    HexGrads hg;
    memset(&hg, 0, sizeof(hg));
    hg.n = S * kHexPlanes;
    // DM4D_HEX_KEEP_SPATIAL: the spatial gradient planes (xy, xz, yz: 134 of the 143 MB at the reference's resolutions) are
    // PERSISTENT buffers that hold zeros outside the texels this plan touches -- the nodes are static, so those texels are
    // the same every step and every one of them is overwritten below: nothing to clear.  The time planes' touched rows
    // move with the timestamps; they are small and cleared as before.
    if (channel_last & DM4D_HEX_KEEP_SPATIAL)
        for (int s = 0; s < S; ++s)
            for (int p : {0, 1, 3}) hg.keep_mask |= 1ull << (s * kHexPlanes + p);
    unsigned long long run = 0;
    for (int s = 0; s < S; ++s)
        for (int p = 0; p < kHexPlanes; ++p) {
            const int k = s * kHexPlanes + p;
            hg.g[k] = g_planes[k];
            if (!hg.g[k]) { set_error("hexplane: null gradient plane %d", k); return DM4D_ERR_INVALID; }
            const size_t n = (size_t)kHexCh * d.res[s][c_axis0_host[p]] * d.res[s][c_axis1_host[p]];   // 32 channels: multiple of 4
            run += n / 4;
            hg.end4[k] = run;
        }
    const size_t total = (size_t)B * M * S * kHexCh;
    const unsigned point_blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(k_hex_bwd_point_zero, dim3(point_blocks + kHexZeroBlocks * (unsigned)hg.n), dim3(256), 0, st, d, g_feat,
                       (float *)scratch, hg, point_blocks);
    DM4D_HIP_CHECK(hipGetLastError());
    HexPlan pl = {n_spatial, n_time, sp_scale, sp_plane, sp_texel, sp_off, sp_item, tp_scale, tp_plane, tp_col, tp_off, tp_item};
    const unsigned sp_blocks = (unsigned)(((size_t)(n_spatial > 0 ? n_spatial : 0) * kHexCh + 255) / 256);
    if (n_time < 0) pl.n_time = 0;
    if (sp_blocks + (unsigned)pl.n_time > 0) {
        hipLaunchKernelGGL(k_hex_bwd_planes, dim3((unsigned)pl.n_time + sp_blocks), dim3(256), 0, st, d, nodes, times, pl,
                           (const float *)scratch, hg);
        DM4D_HIP_CHECK(hipGetLastError());
    }
    return DM4D_OK;
}

}  // extern "C"
