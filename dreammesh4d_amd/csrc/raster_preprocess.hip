// raster_preprocess.hip -- per-Gaussian kernels of the tile rasterizer (gfx950).
//
//   K1 k_preprocess   : frustum cull, cov3D(scale,quat), EWA cov2D + 0.3 low-pass, conic,
//                       radius = ceil(3 sqrt(lambda_max)), pixel centre, tile rect, colour;
//                       per-workgroup duplicate sum and per-tile duplicate histogram.
//   K3 k_scatter      : duplicates -> per-tile segments (slot = tile_start + cursor++).
//   B2 k_gather_bwd   : deterministic gather of the per-duplicate partial gradients written
//                       by render_bwd, then the whole preprocess backward in registers
//                       (conic -> cov2D -> cov3D/mean -> scale/rotation; NDC and depth -> mean).
//
// Replaces preprocessCUDA / duplicateWithKeys / computeCov2DCUDA-bwd / preprocessCUDA-bwd of
// the un-vendored diff-gaussian-rasterization package used at
// custom/threestudio-dreammesh4d/renderer/diff_sugar_rasterizer_temporal.py:169-178,202-211.
//
// All HBM traffic is SoA and coalesced: a wave reads 64 consecutive Gaussians.  Arithmetic
// follows the contract of DESIGN.md (no FMA contraction) so radii / depth bits / tile rects
// compare bit-for-bit with the CPU checker.
#include "common.h"
#include "raster.h"

#ifndef DM4D_GREC
#define DM4D_GREC 128
#endif
namespace dm4d {

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 xform4x3(const f3 p, const float *M)
{
    f3 o;
    o.x = ((M[0] * p.x + M[4] * p.y) + M[8] * p.z) + M[12];
    o.y = ((M[1] * p.x + M[5] * p.y) + M[9] * p.z) + M[13];
    o.z = ((M[2] * p.x + M[6] * p.y) + M[10] * p.z) + M[14];
    return o;
}
__device__ __forceinline__ float4 xform4x4(const f3 p, const float *M)
{
    float4 o;
    o.x = ((M[0] * p.x + M[4] * p.y) + M[8] * p.z) + M[12];
    o.y = ((M[1] * p.x + M[5] * p.y) + M[9] * p.z) + M[13];
    o.z = ((M[2] * p.x + M[6] * p.y) + M[10] * p.z) + M[14];
    o.w = ((M[3] * p.x + M[7] * p.y) + M[11] * p.z) + M[15];
    return o;
}

__device__ __forceinline__ void quat_to_R(const float4 q /* w,x,y,z */, float R[9])
{
    float r = q.x, x = q.y, y = q.z, z = q.w;
    R[0] = 1.f - 2.f * (y * y + z * z);
    R[1] = 2.f * (x * y - r * z);
    R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z);
    R[4] = 1.f - 2.f * (x * x + z * z);
    R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y);
    R[7] = 2.f * (y * z + r * x);
    R[8] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = (R S)(R S)^T, upper triangle
__device__ __forceinline__ void cov3d_from_scale_rot(const f3 scale, float mod, const float4 q, float cov6[6])
{
    float R[9], M[9];
    quat_to_R(q, R);
    float s[3] = {mod * scale.x, mod * scale.y, mod * scale.z};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int j = 0; j < 3; ++j) M[a * 3 + j] = R[a * 3 + j] * s[j];
#define DM4D_SIG(a, b) ((M[a * 3 + 0] * M[b * 3 + 0] + M[a * 3 + 1] * M[b * 3 + 1]) + M[a * 3 + 2] * M[b * 3 + 2])
    cov6[0] = DM4D_SIG(0, 0);
    cov6[1] = DM4D_SIG(0, 1);
    cov6[2] = DM4D_SIG(0, 2);
    cov6[3] = DM4D_SIG(1, 1);
    cov6[4] = DM4D_SIG(1, 2);
    cov6[5] = DM4D_SIG(2, 2);
#undef DM4D_SIG
}

struct Cov2DAux {
    float T0[3], T1[3];
    float tz, tcx, tcy;
    bool xclamped, yclamped;
};

// cov2D = (J W) Sigma (J W)^T (without the low-pass); c = (c00, c01, c11)
__device__ __forceinline__ void cov2d(const f3 mean, const ViewParams &vp, const float *V, const float cov6[6],
                                      float c[3], Cov2DAux *aux)
{
    f3 t = xform4x3(mean, V);
    float limx = 1.3f * vp.tanfovx, limy = 1.3f * vp.tanfovy;
    float txtz = t.x / t.z, tytz = t.y / t.z;
    float cx = fminf(limx, fmaxf(-limx, txtz));
    float cy = fminf(limy, fmaxf(-limy, tytz));
    float tcx = cx * t.z, tcy = cy * t.z;
    float J00 = vp.focal_x / t.z;
    float J02 = -(vp.focal_x * tcx) / (t.z * t.z);
    float J11 = vp.focal_y / t.z;
    float J12 = -(vp.focal_y * tcy) / (t.z * t.z);
    float T0[3], T1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        T0[j] = J00 * V[j * 4 + 0] + J02 * V[j * 4 + 2];
        T1[j] = J11 * V[j * 4 + 1] + J12 * V[j * 4 + 2];
    }
    const float S[9] = {cov6[0], cov6[1], cov6[2], cov6[1], cov6[3], cov6[4], cov6[2], cov6[4], cov6[5]};
    float v0[3], v1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        v0[j] = (T0[0] * S[0 * 3 + j] + T0[1] * S[1 * 3 + j]) + T0[2] * S[2 * 3 + j];
        v1[j] = (T1[0] * S[0 * 3 + j] + T1[1] * S[1 * 3 + j]) + T1[2] * S[2 * 3 + j];
    }
    c[0] = (v0[0] * T0[0] + v0[1] * T0[1]) + v0[2] * T0[2];
    c[1] = (v0[0] * T1[0] + v0[1] * T1[1]) + v0[2] * T1[2];
    c[2] = (v1[0] * T1[0] + v1[1] * T1[1]) + v1[2] * T1[2];
    if (aux) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { aux->T0[j] = T0[j]; aux->T1[j] = T1[j]; }
        aux->tz = t.z;
        aux->tcx = tcx;
        aux->tcy = tcy;
        aux->xclamped = (txtz < -limx || txtz > limx);
        aux->yclamped = (tytz < -limy || tytz > limy);
    }
}

#define DM4D_SH_C0 0.28209479177387814f

__device__ __forceinline__ f3 load3(const float *p, int i) { return f3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }

// ---------------------------------------------------------------------------------------- K1
// One workgroup = kPreBlock (1024) consecutive Gaussians, 4 per thread (coalesced: a wave reads
// 64 consecutive Gaussians per step).  The per-tile duplicate histogram of the workgroup is
// accumulated with LDS atomics and written densely to hist[WG][tile]; K2 turns it into
// per-(WG, tile) bases.  No global atomics except one add per workgroup for D.
__global__ __launch_bounds__(kPreThreads) void k_preprocess(BatchDesc d)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist[];   // [T]
    const ViewCtx c = resolve(d, blockIdx.y);
    const ViewParams &vp = c.vp;
    const dm4d_raster_inputs &in = c.in;
    int32_t *__restrict__ radii = c.radii;
    const GeomPtrs &g = c.g;
    const int T = c.T;
    __shared__ float sV[16], sP[16];
    __shared__ uint32_t s_wsum[2][kPreThreads / 64];
    const int tid = threadIdx.x;
    if (tid < 16) { sV[tid] = vp.view[tid]; sP[tid] = vp.proj[tid]; }
    for (int t = tid; t < T; t += kPreThreads) s_hist[t] = 0u;
    __syncthreads();
    uint32_t touched_sum = 0, rec_sum = 0;
#pragma unroll 1
    for (int it = 0; it < kPreItems; ++it) {
        const int i = blockIdx.x * kPreBlock + it * kPreThreads + tid;
        if (i >= in.N) break;
        uint32_t touched = 0, recs = 0;
        int my_radius = 0;
        const f3 p = load3(in.means3D, i);
        const f3 pv = xform4x3(p, sV);
        if (pv.z > 0.2f) {
            const float4 ph = xform4x4(p, sP);
            const float pw = 1.0f / (ph.w + 0.0000001f);
            const float ppx = ph.x * pw, ppy = ph.y * pw;
            float cov6[6];
            if (in.cov3D_precomp) {
#pragma unroll
                for (int k = 0; k < 6; ++k) cov6[k] = in.cov3D_precomp[6 * (size_t)i + k];
            } else {
                const float4 q = reinterpret_cast<const float4 *>(in.rotations)[i];
                cov3d_from_scale_rot(load3(in.scales, i), vp.scale_modifier, q, cov6);
            }
            float c[3];
            cov2d(p, vp, sV, cov6, c, nullptr);
            c[0] += 0.3f;
            c[2] += 0.3f;
            const float det = c[0] * c[2] - c[1] * c[1];
            if (det != 0.0f) {
                const float det_inv = 1.f / det;
                const float mid = 0.5f * (c[0] + c[2]);
                const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
                const float lambda1 = mid + disc, lambda2 = mid - disc;
                const int r = f2i_sat(ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2))));
                const float px = ((ppx + 1.0f) * (float)vp.W - 1.0f) * 0.5f;
                const float py = ((ppy + 1.0f) * (float)vp.H - 1.0f) * 0.5f;
                const Rect rc = tile_rect(px, py, r, vp.gx, vp.gy);
                const int area = (rc.x1 - rc.x0) * (rc.y1 - rc.y0);
                if (area != 0) {
                    my_radius = r;
                    touched = (uint32_t)area;
                    g.xy[i] = make_float2(px, py);
                    g.depth[i] = pv.z;
                    const float4 co = make_float4(c[2] * det_inv, -c[1] * det_inv, c[0] * det_inv, in.opacities[i]);
                    g.conic_opacity[i] = co;
                    const Bands bd = cell_bands(px, py, co.x, co.y, co.z, co.w, rc);
                    recs = (uint32_t)(bd.nbx * bd.nby);
                    uint32_t dense = 1u;
                    const float cdet = co.x * co.z - co.y * co.y;
                    // (a block that is one cell wide or high is its own exact bound: only corners can be empty)
                    if (bd.nbx >= 2 && bd.nby >= 2 && recs <= 64u && cdet > 0.f && co.x > 0.f && co.z > 0.f) {
                        // small block: keep only the cells the ellipse really meets (a third of the cells of the
                        // axis-aligned bound of a thin diagonal splat are empty)
                        const float tau = __logf(255.0f * co.w) * 1.001f + 0.01f;
                        const EllipseRows er = ellipse_rows(co.x, co.y, co.z, tau);
                        uint64_t mask = 0ull;
                        for (int by = 0; by < bd.nby; ++by) {
                            int b0, b1;
                            row_span(er, px, py, (float)(4 * (bd.by0 + by)), b0, b1);
                            b0 = max(b0 - bd.bx0, 0);
                            b1 = min(b1 - bd.bx0, bd.nbx - 1);
                            if (b0 <= b1) mask |= ((~0ull >> (63 - (b1 - b0))) << b0) << (by * bd.nbx);
                        }
                        g.cellmask[i] = mask;
                        recs = (uint32_t)__popcll(mask);
                        dense = 0u;
                    }
                    g.cellinfo[i] = make_uint4((uint32_t)bd.bx0 | ((uint32_t)bd.by0 << 16),
                                               (uint32_t)bd.nbx | ((uint32_t)bd.nby << 16), 0u, dense);
                    if (d.tile_records) {      // tile-record mode: one backward record per tile the cell block spans
                        const TileSpan ts = tile_span((uint32_t)bd.bx0 | ((uint32_t)bd.by0 << 16), (uint32_t)bd.nbx | ((uint32_t)bd.nby << 16));
                        recs = (uint32_t)(ts.tnx * ts.tny);
                    }
                    if (in.shs) {
                        const float *sh = in.shs + (size_t)i * in.sh_coeffs * 3;
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            float v = DM4D_SH_C0 * sh[ch] + 0.5f;
                            g.clamped[3 * (size_t)i + ch] = (v < 0.f);
                            g.rgb[3 * (size_t)i + ch] = fmaxf(v, 0.f);
                        }
                    }
                    for (int y = rc.y0; y < rc.y1; ++y)
                        for (int x = rc.x0; x < rc.x1; ++x) atomicAdd(&s_hist[y * vp.gx + x], 1u);   // LDS
                }
            }
        }
        radii[i] = my_radius;
        g.tiles_touched[i] = touched;
        g.rec_touched[i] = recs;
        touched_sum += touched;
        rec_sum += recs;
    }
    const uint32_t ws = wave_sum_u32(touched_sum), wr = wave_sum_u32(rec_sum);
    if ((tid & 63) == 0) { s_wsum[0][tid >> 6] = ws; s_wsum[1][tid >> 6] = wr; }
    __syncthreads();
    uint32_t *row = g.hist + (size_t)blockIdx.x * T;
    for (int t = tid; t < T; t += kPreThreads) row[t] = s_hist[t];
    if (tid == 0) {
        uint32_t s = 0, r = 0;
#pragma unroll
        for (int w = 0; w < kPreThreads / 64; ++w) { s += s_wsum[0][w]; r += s_wsum[1][w]; }
        g.block_sums[blockIdx.x] = s;
        g.rec_block_sums[blockIdx.x] = r;
        if (s) atomicAdd(&g.counters[kCntD], s);   // two integer atomics per workgroup
        if (r) atomicAdd(&g.counters[kCntR], r);
    }
}

// ---------------------------------------------------------------------------------------- K3
// Same 1024-Gaussian workgroups as K1.  LDS holds cursor[t] = tile_start[t] + (duplicates of
// earlier workgroups in tile t); every duplicate takes its slot with one LDS atomic.
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *s_w /* [4] */, uint32_t *total)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t incl = wave_incl_scan_u32(v, lane);
    __syncthreads();
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kPreThreads / 64; ++w) {
        const uint32_t x = s_w[w];
        if (w < wv) pre += x;
        tot += x;
    }
    *total = tot;
    return pre + incl - v;
}

__global__ __launch_bounds__(kPreThreads) void k_scatter(BatchDesc d)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_cursor[];   // [T]
    const ViewCtx c = resolve(d, blockIdx.y);
    const ViewParams &vp = c.vp;
    const int N = c.in.N, T = c.T;
    const int32_t *__restrict__ radii = c.radii;
    const GeomPtrs &g = c.g;
    const BinPtrs &b = c.b;
    const uint32_t cap = c.cap;
    __shared__ uint32_t s_w[kPreThreads / 64];
    const int tid = threadIdx.x;
    // tile_start = exclusive scan of tile_count (each workgroup redoes this small scan in LDS;
    // workgroup 0 publishes it for the later kernels)
    {
        const int per = (T + kPreThreads - 1) / kPreThreads;
        const int t0 = tid * per, t1 = min(T, t0 + per);
        uint32_t loc = 0;
        for (int t = t0; t < t1; ++t) loc += g.tile_count[t];
        uint32_t total;
        uint32_t run = block_excl_scan_256(loc, s_w, &total);
        const uint32_t *row = g.hist + (size_t)blockIdx.x * T;
        for (int t = t0; t < t1; ++t) {
            if (blockIdx.x == 0) g.tile_start[t] = run;
            s_cursor[t] = run + row[t];
            run += g.tile_count[t];
        }
        if (blockIdx.x == 0 && tid == 0) g.tile_start[T] = total;
    }
    // Workgroup 0 also publishes the view's tiles by DESCENDING duplicate count (256-bucket counting sort):
    // the launch order of K4 and of the blend kernels (longest-processing-time first -- a tile's work is
    // serial in its list length, so the long silhouette tiles must start first to overlap with the rest).
    if (blockIdx.x == 0) {
        __shared__ uint32_t s_oh[256], s_oc[256], s_omax;
        if (tid == 0) s_omax = 1u;
        s_oh[tid] = 0u;
        __syncthreads();
        uint32_t wmax = 0;
        for (int t = tid; t < T; t += kPreThreads) wmax = max(wmax, g.tile_count[t]);
        wmax = wave_max_u32(wmax);
        if ((tid & 63) == 0) atomicMax(&s_omax, wmax);
        __syncthreads();
        const float oscale = 255.0f / (float)s_omax;
        for (int t = tid; t < T; t += kPreThreads)
            atomicAdd(&s_oh[255 - min(255, (int)((float)g.tile_count[t] * oscale))], 1u);
        __syncthreads();
        if (tid == 0) {
            uint32_t run = 0;
            for (int k = 0; k < 256; ++k) { s_oc[k] = run; run += s_oh[k]; }
        }
        __syncthreads();
        for (int t = tid; t < T; t += kPreThreads)
            g.order[atomicAdd(&s_oc[255 - min(255, (int)((float)g.tile_count[t] * oscale))], 1u)] = (uint32_t)t;
    }
    // first backward record of this workgroup = records of the earlier workgroups
    uint32_t rcarry;
    {
        uint32_t rloc = 0;
        for (int w = tid; w < (int)blockIdx.x; w += kPreThreads) rloc += g.rec_block_sums[w];
        uint32_t total;
        block_excl_scan_256(rloc, s_w, &total);
        rcarry = total;
    }
    __syncthreads();
    bool overflow = false, rec_overflow = false;
#pragma unroll 1
    for (int it = 0; it < kPreItems; ++it) {
        const int i = blockIdx.x * kPreBlock + it * kPreThreads + tid;
        const uint32_t touched = (i < N) ? g.tiles_touched[i] : 0u;
        uint32_t total;
        const uint32_t recs = (i < N) ? g.rec_touched[i] : 0u;
        const uint32_t r0 = rcarry + block_excl_scan_256(recs, s_w, &total);
        rcarry += total;
        if (i < N && touched != 0u) {
            reinterpret_cast<uint32_t *>(g.cellinfo + i)[2] = r0;   // first backward record of the Gaussian
            g.rec0[i] = r0;
            rec_overflow |= (recs > 0u) && ((uint64_t)r0 + recs > (uint64_t)c.rec_cap);
        }
        if (touched == 0) continue;
        const float2 xy = g.xy[i];
        const Rect rc = tile_rect(xy.x, xy.y, radii[i], vp.gx, vp.gy);
        const uint32_t dbits = __float_as_uint(g.depth[i]);
        uint32_t n = 0;
        for (int y = rc.y0; y < rc.y1; ++y)
            for (int x = rc.x0; x < rc.x1; ++x, ++n) {
                const uint32_t slot = atomicAdd(&s_cursor[y * vp.gx + x], 1u);   // LDS
                if (slot < cap) {
                    b.u_depth[slot] = dbits;
                    b.u_idx[slot] = (uint32_t)i;
                } else {
                    overflow = true;
                }
            }
    }
    if (overflow) g.counters[kCntOverflow] = 1u;
    if (rec_overflow) g.counters[kCntRecOverflow] = 1u;
}

// ---------------------------------------------------------------------------------------- B2
// What one Gaussian of one view gets back from B2: its gradients w.r.t. the 3D mean, the rotation (w, x, y, z) and the blended
// colour channels -- kept in registers by the fused gather + face kernel (gather_face.hip), which never writes them per view.
struct GatherOut { float dmean[3], drot[4], dcol[kMaxChannels]; };

// PARTS: float4 per record: 2 (lean without depth gradient), 3 (12-float records: 3 channels, or lean 6-channel) or 4 (16 floats).
// The body of B2 for Gaussian i of the view `c` (every lane of the wave calls it, for 64 CONSECUTIVE Gaussians of the SAME view:
// the record streaming is wave-cooperative).  sV / sP: the view's matrices in LDS; chunk: this wave's staging window.
// Records per LDS window of a wave: 128, and 96 for the 64-byte records of the full backward (4 x 7.7 KB per workgroup instead of
// 4 x 10.2: four workgroups per CU -- what its ~100 VGPRs allow -- instead of three)
constexpr int kGatherRecords(int parts) { return parts <= 3 ? DM4D_GREC : 96; }
// Per-view outputs (c.o.*) are written where their pointer is set.
template <int PARTS>
__device__ __forceinline__ void gather_gaussian(const BatchDesc &d, const ViewCtx &c, const int i, const bool live, const float *sV,
                                                const float *sP, float *chunk, GatherOut &res)
{
    const ViewParams &vp = c.vp;
    const dm4d_raster_inputs &in = c.in;
    const int32_t *__restrict__ radii = c.radii;
    const GeomPtrs &g = c.g;
    const float *__restrict__ dLt = c.dLq;
    const BwdOutputs &o = c.o;
    constexpr int kGRec = kGatherRecords(PARTS), kGStride = PARTS <= 3 ? 12 : 20;
    constexpr int NQ = kGRec * PARTS / 64;      // float4 a lane holds of a window in flight
    const int lane = threadIdx.x & 63;
    const size_t si = (size_t)i;
#pragma unroll
    for (int k = 0; k < 3; ++k) res.dmean[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) res.drot[k] = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxChannels; ++k) res.dcol[k] = 0.f;

    // acc: 0,1 dL/dmean2D (NDC) | 2,3,4 dL/dconic (A,B,C) | 5 dL/dopacity | 6 dL/ddepth | 7.. dL/dcolour
    float acc[7 + kMaxChannels];
#pragma unroll
    for (int k = 0; k < 7 + kMaxChannels; ++k) acc[k] = 0.f;
    const int r = live ? radii[i] : 0;
    const int C = vp.C;
    // the per-Gaussian inputs of the preprocess backward are requested BEFORE the records are streamed (round 6: their round trip used to follow
    // the record sum; k_gather_face_bwd<2> 287.5 -> 277.1 us per 20 views)
    f3 h_m = {0.f, 0.f, 0.f}, h_sc = {0.f, 0.f, 0.f};
    float4 h_q = make_float4(1.f, 0.f, 0.f, 0.f), h_co = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r > 0) {
        h_m = load3(in.means3D, i);
        if (!in.cov3D_precomp) { h_q = reinterpret_cast<const float4 *>(in.rotations)[i]; h_sc = load3(in.scales, i); }
        h_co = g.conic_opacity[i];
    }
    {
        // The records of a Gaussian are one contiguous block (its reached cells, K1), the blocks of consecutive
        // Gaussians follow each other (K3's scan), and B1 wrote every record -- real sums for the entries the
        // forward consumed, zeros for the rest.  So the 64 Gaussians of a wave own ONE contiguous range of
        // records: the wave streams it through LDS with fully coalesced 1 KB loads and every lane adds up its
        // own records from there, in record order.
        const bool lean = d.lean != 0;
        constexpr int RS = PARTS * 4;
        uint32_t rec0 = 0xFFFFFFFFu, end = 0u;
        if (r > 0) {
            const uint32_t cnt = g.rec_touched[i];
            if (cnt > 0u) {
                rec0 = g.cellinfo[i].z;
                end = min(rec0 + cnt, max(rec0, c.rec_cap));
            }
        }
        uint32_t R0 = rec0, R1 = end;
        R0 = wave_min_u32(R0);
        R1 = wave_max_u32(R1);
        // software pipeline: the next window's loads are in flight while this one is summed
        float4 pq[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) pq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#define DM4D_FETCH(BASE)                                                                                   \
        {                                                                                                  \
            const uint32_t n4_ = min((uint32_t)kGRec, R1 - (BASE)) * (uint32_t)PARTS;                      \
            const float4 *src4_ = reinterpret_cast<const float4 *>(dLt + (size_t)(BASE) * RS);             \
            _Pragma("unroll") for (int q = 0; q < NQ; ++q)                                                 \
                if ((uint32_t)lane + 64u * q < n4_) pq[q] = src4_[lane + 64 * q];                          \
        }
        if (R0 < R1) DM4D_FETCH(R0)
        for (uint32_t base = R0; base < R1; base += kGRec) {
            const uint32_t nrec = min((uint32_t)kGRec, R1 - base);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const uint32_t idx4 = (uint32_t)lane + 64u * q;
                if (idx4 < nrec * (uint32_t)PARTS)
                    *reinterpret_cast<float4 *>(chunk + (idx4 / (uint32_t)PARTS) * kGStride + (idx4 % (uint32_t)PARTS) * 4) = pq[q];
            }
            if (base + kGRec < R1) DM4D_FETCH(base + kGRec)
            __builtin_amdgcn_wave_barrier();
            const uint32_t lo = max(rec0, base), hi = min(end, base + nrec);
            for (uint32_t slot = lo; slot < hi; ++slot) {
                const float4 *rp = reinterpret_cast<const float4 *>(chunk + (slot - base) * kGStride);
                const float4 a0 = rp[0], a1 = rp[1];
                acc[0] += a0.x; acc[1] += a0.y; acc[2] += a0.z; acc[3] += a0.w;
                if (PARTS == 2) {   // mean2D 2 | conic 3 | colour channels 3..5
                    acc[4] += a1.x; acc[10] += a1.y; acc[11] += a1.z; acc[12] += a1.w;
                    continue;
                }
                const float4 a2 = rp[2];
                if (lean) {   // mean2D 2 | conic 3 | depth | colour channels 3..5
                    acc[4] += a1.x; acc[6] += a1.y; acc[10] += a1.z; acc[11] += a1.w; acc[12] += a2.x;
                    continue;
                }
                acc[4] += a1.x; acc[5] += a1.y; acc[6] += a1.z; acc[7] += a1.w;
                acc[8] += a2.x; acc[9] += a2.y;
                if (C > 3) {
                    const float4 a3 = rp[3];
                    acc[10] += a2.z; acc[11] += a2.w; acc[12] += a3.x;
                }
            }
        }
#undef DM4D_FETCH
    }
    {
        // B1's records carry the moments sum_pixels q (dx, dy, dx^2, dx dy, dy^2), q = dL/dG G (raster_render.hip):
        // dL/dmean2D (NDC) = -(A m0 + B m1, B m0 + C m1) (W/2, H/2); dL/dconic = (-m2 / 2, -m3, -m4 / 2)
        const float4 co = h_co;
        const float m0 = acc[0], m1 = acc[1];
        acc[0] = -(co.x * m0 + co.y * m1) * (0.5f * (float)vp.W);
        acc[1] = -(co.y * m0 + co.z * m1) * (0.5f * (float)vp.H);
        acc[2] = -0.5f * acc[2];
        acc[3] = -acc[3];
        acc[4] = -0.5f * acc[4];
    }
    if (!live) return;
    if (o.dL_dmeans2D) {
        o.dL_dmeans2D[3 * si + 0] = acc[0];
        o.dL_dmeans2D[3 * si + 1] = acc[1];
        o.dL_dmeans2D[3 * si + 2] = 0.f;
    }
#pragma unroll
    for (int ch = 0; ch < kMaxChannels; ++ch) res.dcol[ch] = acc[7 + ch];
    if (o.dL_dopacity) {
        // the records carry sum q = opacity * sum G dL/dalpha (B1's q already holds the opacity factor; a Gaussian with
        // opacity < 1/255 contributes nowhere, its sum is an exact 0)
        const float op_ = in.opacities[i];
        o.dL_dopacity[si] = (r > 0 && op_ > 0.f) ? acc[5] / op_ : 0.f;
    }
    if (o.dL_dcolors) {
        for (int ch = 0; ch < C; ++ch) o.dL_dcolors[(size_t)C * si + ch] = acc[7 + ch];
    }
    if (o.dL_dsh && in.shs) {
        const int M = in.sh_coeffs;
        for (int ch = 0; ch < 3; ++ch)
            o.dL_dsh[si * M * 3 + ch] = (r > 0 && !g.clamped[3 * si + ch]) ? DM4D_SH_C0 * acc[7 + ch] : 0.f;
        for (int k = 3; k < 3 * M; ++k) o.dL_dsh[si * M * 3 + k] = 0.f;
    }

    float dmean[3] = {0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f};
    float drot[4] = {0.f, 0.f, 0.f, 0.f};
    if (r > 0) {
        const f3 m = h_m;
        float cov6[6];
        float4 q = make_float4(1.f, 0.f, 0.f, 0.f);
        f3 sc = {0.f, 0.f, 0.f};
        if (in.cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; ++k) cov6[k] = in.cov3D_precomp[6 * si + k];
        } else {
            q = h_q; sc = h_sc;
            cov3d_from_scale_rot(sc, vp.scale_modifier, q, cov6);
        }
        // ---- conic -> cov2D -> (cov3D, T) ----
        float c[3];
        Cov2DAux aux;
        cov2d(m, vp, sV, cov6, c, &aux);
        const float a = c[0] + 0.3f, bb = c[1], cc = c[2] + 0.3f;
        const float denom = a * cc - bb * bb;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        const float gA = acc[2], gB = acc[3], gC = acc[4];
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        const float *T0 = aux.T0, *T1 = aux.T1;
        if (denom2inv != 0.f) {
            dL_da = denom2inv * (-cc * cc * gA + bb * cc * gB + (denom - a * cc) * gC);
            dL_dc = denom2inv * (-a * a * gC + a * bb * gB + (denom - a * cc) * gA);
            dL_db = denom2inv * (2 * bb * cc * gA - (denom + 2 * bb * bb) * gB + 2 * a * bb * gC);
            dcov[0] = T0[0] * T0[0] * dL_da + T0[0] * T1[0] * dL_db + T1[0] * T1[0] * dL_dc;
            dcov[3] = T0[1] * T0[1] * dL_da + T0[1] * T1[1] * dL_db + T1[1] * T1[1] * dL_dc;
            dcov[5] = T0[2] * T0[2] * dL_da + T0[2] * T1[2] * dL_db + T1[2] * T1[2] * dL_dc;
            dcov[1] = 2 * T0[0] * T0[1] * dL_da + (T0[0] * T1[1] + T0[1] * T1[0]) * dL_db + 2 * T1[0] * T1[1] * dL_dc;
            dcov[2] = 2 * T0[0] * T0[2] * dL_da + (T0[0] * T1[2] + T0[2] * T1[0]) * dL_db + 2 * T1[0] * T1[2] * dL_dc;
            dcov[4] = 2 * T0[2] * T0[1] * dL_da + (T0[1] * T1[2] + T0[2] * T1[1]) * dL_db + 2 * T1[1] * T1[2] * dL_dc;
        }
        {
            const float S[9] = {cov6[0], cov6[1], cov6[2], cov6[1], cov6[3], cov6[4], cov6[2], cov6[4], cov6[5]};
            float dT0[3], dT1[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float s0 = S[k * 3 + 0] * T0[0] + S[k * 3 + 1] * T0[1] + S[k * 3 + 2] * T0[2];
                const float s1 = S[k * 3 + 0] * T1[0] + S[k * 3 + 1] * T1[1] + S[k * 3 + 2] * T1[2];
                dT0[k] = 2 * s0 * dL_da + s1 * dL_db;
                dT1[k] = 2 * s1 * dL_dc + s0 * dL_db;
            }
            float dJ00 = 0, dJ02 = 0, dJ11 = 0, dJ12 = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                dJ00 += sV[k * 4 + 0] * dT0[k];
                dJ02 += sV[k * 4 + 2] * dT0[k];
                dJ11 += sV[k * 4 + 1] * dT1[k];
                dJ12 += sV[k * 4 + 2] * dT1[k];
            }
            const float tz = 1.f / aux.tz, tz2 = tz * tz, tz3 = tz2 * tz;
            const float xm = aux.xclamped ? 0.f : 1.f, ym = aux.yclamped ? 0.f : 1.f;
            const float fx = vp.focal_x, fy = vp.focal_y;
            const float dtx = xm * -fx * tz2 * dJ02;
            const float dty = ym * -fy * tz2 * dJ12;
            const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * aux.tcx) * tz3 * dJ02 +
                              (2 * fy * aux.tcy) * tz3 * dJ12;
#pragma unroll
            for (int j = 0; j < 3; ++j) dmean[j] = sV[j * 4 + 0] * dtx + sV[j * 4 + 1] * dty + sV[j * 4 + 2] * dtz;
        }
        // ---- NDC-space mean gradient through the projection ----
        {
            const float4 ph = xform4x4(m, sP);
            const float mw = 1.0f / (ph.w + 0.0000001f);
            const float mul1 = ph.x * mw * mw, mul2 = ph.y * mw * mw;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                dmean[j] += (sP[j * 4 + 0] * mw - sP[j * 4 + 3] * mul1) * acc[0] +
                            (sP[j * 4 + 1] * mw - sP[j * 4 + 3] * mul2) * acc[1];
        }
        // ---- depth channel -> mean ----
        {
            const float mul3 = sV[2] * m.x + sV[6] * m.y + sV[10] * m.z + sV[14];
#pragma unroll
            for (int j = 0; j < 3; ++j) dmean[j] += (sV[j * 4 + 2] - sV[j * 4 + 3] * mul3) * acc[6];
        }
        // ---- cov3D -> scale / rotation ----
        if (!in.cov3D_precomp) {
            float R[9];
            quat_to_R(q, R);
            const float mod = vp.scale_modifier;
            const float s[3] = {mod * sc.x, mod * sc.y, mod * sc.z};
            const float Gs[9] = {dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3],
                                 0.5f * dcov[4], 0.5f * dcov[2], 0.5f * dcov[4], dcov[5]};
            float dR[9];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float ds = 0.f;
#pragma unroll
                for (int aa = 0; aa < 3; ++aa) {
                    float a2 = 0.f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) a2 += Gs[aa * 3 + k] * (R[k * 3 + j] * s[j]);
                    const float dM = 2.f * a2;
                    ds += R[aa * 3 + j] * dM;
                    dR[aa * 3 + j] = dM * s[j];
                }
                dscale[j] = mod * ds;
            }
            const float rr = q.x, x = q.y, y = q.z, z = q.w;
            drot[0] = 2 * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
            drot[1] = 2 * (y * dR[1] + z * dR[2] + y * dR[3] - 2 * x * dR[4] - rr * dR[5] + z * dR[6] + rr * dR[7] - 2 * x * dR[8]);
            drot[2] = 2 * (-2 * y * dR[0] + x * dR[1] + rr * dR[2] + x * dR[3] + z * dR[5] - rr * dR[6] + z * dR[7] - 2 * y * dR[8]);
            drot[3] = 2 * (-2 * z * dR[0] - rr * dR[1] + x * dR[2] + rr * dR[3] - 2 * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) res.dmean[k] = dmean[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) res.drot[k] = drot[k];
    if (o.dL_dmeans3D) {
        o.dL_dmeans3D[3 * si + 0] = dmean[0];
        o.dL_dmeans3D[3 * si + 1] = dmean[1];
        o.dL_dmeans3D[3 * si + 2] = dmean[2];
    }
    if (o.dL_dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; ++k) o.dL_dcov3D[6 * si + k] = dcov[k];
    }
    if (o.dL_dscales) {
        o.dL_dscales[3 * si + 0] = dscale[0];
        o.dL_dscales[3 * si + 1] = dscale[1];
        o.dL_dscales[3 * si + 2] = dscale[2];
    }
    if (o.dL_drotations)
        reinterpret_cast<float4 *>(o.dL_drotations)[i] = make_float4(drot[0], drot[1], drot[2], drot[3]);
}

constexpr int kGatherChunkFloats(int parts) { return kGatherRecords(parts) * (parts <= 3 ? 12 : 20); }
template <int PARTS>
__global__ __launch_bounds__(kPreThreads) void k_gather_bwd(BatchDesc d)
{
    const ViewCtx c = resolve(d, blockIdx.y);
    __shared__ float sV[16], sP[16];
    // per-wave staging window of kGRec records.  In a window of n records only ~n / 8 lanes (Gaussians) have
    // anything to add and the wave loops for the longest of them, so a larger window means fewer, better filled
    // iterations.  Row stride: 12 floats for the 12-float records (16 lanes reading 16 consecutive records with
    // ds_read_b128 hit disjoint bank groups: 12 i mod 64 are 16 different multiples of 4), 20 for the 16-float ones.
    __shared__ __attribute__((aligned(16))) float s_chunk[kPreThreads / 64][kGatherChunkFloats(PARTS)];
    const int tid = threadIdx.x;
    if (tid < 16) { sV[tid] = c.vp.view[tid]; sP[tid] = c.vp.proj[tid]; }
    __syncthreads();
    const int i = blockIdx.x * kPreThreads + tid;
    GatherOut res;
    gather_gaussian<PARTS>(d, c, i, i < c.in.N, sV, sP, s_chunk[tid >> 6], res);
}

__global__ void k_mark_visible(int N, const float *__restrict__ means3D, const float *__restrict__ view,
                               uint8_t *__restrict__ present)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const f3 pv = xform4x3(load3(means3D, i), view);
    present[i] = pv.z > 0.2f;
}

// ---------------------------------------------------------------------------------------- launchers
__global__ void k_zero_counters(BatchDesc d)
{
    const int b = blockIdx.x;
    if (threadIdx.x < 64) reinterpret_cast<uint32_t *>(d.geom + (size_t)b * d.geom_stride)[threadIdx.x] = 0u;
}

int launch_zero_counters(const BatchDesc &d, hipStream_t st)
{
    hipLaunchKernelGGL(k_zero_counters, dim3(d.B), dim3(64), 0, st, d);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int launch_preprocess(const BatchDesc &d, hipStream_t st)
{
    const int nb = (d.N + kPreBlock - 1) / kPreBlock;
    if (nb == 0) return DM4D_OK;
    const int T = ((d.W + kTile - 1) / kTile) * ((d.H + kTile - 1) / kTile);
    if ((size_t)T * 4 > 32768)
        DM4D_HIP_CHECK(hipFuncSetAttribute((const void *)k_preprocess, hipFuncAttributeMaxDynamicSharedMemorySize, T * 4));
    ProfScope prof_(kKPreprocess, st);
    hipLaunchKernelGGL(k_preprocess, dim3(nb, d.B), dim3(kPreThreads), (size_t)T * 4, st, d);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int launch_scatter(const BatchDesc &d, hipStream_t st)
{
    // always launched (even with N == 0): workgroup 0 publishes tile_start for the render kernels
    const int nb = (d.N + kPreBlock - 1) / kPreBlock;
    const int T = ((d.W + kTile - 1) / kTile) * ((d.H + kTile - 1) / kTile);
    if ((size_t)T * 4 > 32768)
        DM4D_HIP_CHECK(hipFuncSetAttribute((const void *)k_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, T * 4));
    ProfScope prof_(kKScatter, st);
    hipLaunchKernelGGL(k_scatter, dim3(nb > 0 ? nb : 1, d.B), dim3(kPreThreads), (size_t)T * 4, st, d);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int launch_gather_bwd(const BatchDesc &d, hipStream_t st)
{
    const int nb = (d.N + kPreThreads - 1) / kPreThreads;
    if (nb == 0) return DM4D_OK;
    ProfScope prof_(kKGatherBwd, st);
    if (grad_stride(d.C, d.lean) == 8) hipLaunchKernelGGL(k_gather_bwd<2>, dim3(nb, d.B), dim3(kPreThreads), 0, st, d);
    else if (grad_stride(d.C, d.lean) == 12) hipLaunchKernelGGL(k_gather_bwd<3>, dim3(nb, d.B), dim3(kPreThreads), 0, st, d);
    else hipLaunchKernelGGL(k_gather_bwd<4>, dim3(nb, d.B), dim3(kPreThreads), 0, st, d);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int launch_mark_visible(int N, const float *means3D, const float *view, uint8_t *present, hipStream_t st)
{
    if (N <= 0) return DM4D_OK;
    hipLaunchKernelGGL(k_mark_visible, dim3((N + 255) / 256), dim3(256), 0, st, N, means3D, view, present);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // namespace dm4d
