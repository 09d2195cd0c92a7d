// raster_render.hip -- alpha-compositing kernels of the tile rasterizer (gfx950, wave64).
//
// Execution unit = ONE WAVE per 8x8-pixel quadrant of a 16x16 tile.  Tiles stay 16x16 (the
// reference's BLOCK_X x BLOCK_Y) so tile lists, keys, n_contrib and final_T keep their
// upstream meaning, but K4 splits every tile's depth-sorted list into four quadrant lists using
// the exact bound of each splat's alpha >= 1/255 ellipse (quadrant_mask, raster.h).  A
// mesh-bound splat is ~1 px wide, so a quadrant list holds under half of its tile's list, and
// culled (splat, pixel) pairs are exactly ones the reference `continue`s on: results are
// unchanged.  One wave per workgroup means no barriers at all, 4x more independent work items
// than tiles for the 256 CUs to balance, and early exit at 8x8 granularity.
//
// Forward  (K5): 64 list entries at a time are gathered by the 64 lanes (coalesced list read,
//                L2-resident attribute gathers), double-buffered through wave-private LDS; every
//                lane then walks the chunk with broadcast LDS reads.  Front-to-back blend;
//                the wave stops when all 64 pixels are saturated (ballot).
// Backward (B1): back-to-front over the entries the forward consumed.  The 64-pixel sums of the
//                10 (13 with 6 colour channels) per-entry gradients use a packed butterfly:
//                v_permlane32_swap and v_permlane16_swap fold 4 values into one register before
//                the in-row DPP steps (28 instead of 60 cross-lane instructions), and one
//                store writes the whole record.  Records are indexed by (quadrant, tile-list
//                position): no floating-point atomics, gradients are bit-reproducible.
//
// blockIdx -> (tile, quadrant) keeps the 4 quadrants of a tile on ONE XCD (blocks are dispatched
// round-robin over the 8 XCDs), so the attribute gathers of neighbouring quadrants share an L2.
//
// Replaces renderCUDA fwd/bwd of the un-vendored diff-gaussian-rasterization (ashawkey fork:
// extra depth and alpha channels) used at
// custom/threestudio-dreammesh4d/renderer/diff_sugar_rasterizer_temporal.py:169-178,202-211.
// C = 6 blends the RGB pass and the normal pass of one view together (same geometry, :202-211).
#include "common.h"
#include "raster.h"

namespace dm4d {

typedef unsigned int u2v __attribute__((ext_vector_type(2)));

template <int C> struct StagedN { static constexpr int kVec = (C <= 3) ? 3 : 4; };

// gather one list entry: a = (x, y, conic.x, conic.y)  b = (conic.z, opacity, depth, k bits)
//                        c = colours 0..3               d = colours 4..5
template <int C>
__device__ __forceinline__ void gather_entry(const uint2 qe, const GeomPtrs &g, const float *__restrict__ colors,
                                             float4 (&r)[4])
{
    const uint32_t gid = qe.x;
    const float2 xy = g.xy[gid];
    const float4 co = g.conic_opacity[gid];
    const float dep = g.depth[gid];
    const float *c = colors + (size_t)C * gid;
    r[0] = make_float4(xy.x, xy.y, co.x, co.y);
    r[1] = make_float4(co.z, co.w, dep, __uint_as_float(qe.y));
    if (C <= 3) {
        r[2] = make_float4(c[0], c[1], c[2], 0.f);
    } else {
        const float2 c01 = *reinterpret_cast<const float2 *>(c);
        const float2 c23 = *reinterpret_cast<const float2 *>(c + 2);
        const float2 c45 = *reinterpret_cast<const float2 *>(c + 4);
        r[2] = make_float4(c01.x, c01.y, c23.x, c23.y);
        r[3] = make_float4(c45.x, c45.y, 0.f, 0.f);
    }
}

__device__ __forceinline__ void block_to_quadrant(int b, int &tile, int &q)
{
    const int xcd = b & 7, r = b >> 3;
    q = r & 3;
    tile = (r >> 2) * 8 + xcd;
}

// ---------------------------------------------------------------------------------------- K5
template <int C>
__global__ __launch_bounds__(64) void k_render_fwd(BatchDesc d)
{
    constexpr int NV = StagedN<C>::kVec;
    // one wave-private staging buffer: the next chunk waits in registers (prefetched during the
    // blend loop) and is written after the loop -- same wave, program order, no hazard
    __shared__ float4 s_e[64][NV];
    const ViewCtx c = resolve(d, blockIdx.y);
    const ViewParams &vp = c.vp;
    const float *__restrict__ colors = c.colors;
    const GeomPtrs &g = c.g;
    const BinPtrs &b = c.b;
    const uint32_t cap = c.cap;
    const ImgPtrs &im = c.im;
    float *__restrict__ out_color = c.out_color, *__restrict__ out_depth = c.out_depth,
                       *__restrict__ out_alpha = c.out_alpha;
    const int T = c.T;
    int tile, q;
    block_to_quadrant(blockIdx.x, tile, q);
    if (tile >= T) return;
    const int lane = threadIdx.x;
    const int tx = tile % vp.gx, ty = tile / vp.gx;
    const int px = tx * kTile + (q & 1) * 8 + (lane & 7);
    const int py = ty * kTile + (q >> 1) * 8 + (lane >> 3);
    const bool inside = px < vp.W && py < vp.H;
    const float pxf = (float)px, pyf = (float)py;

    const uint32_t s = g.tile_start[tile];
    const uint32_t nq = (s < cap) ? g.qcount[tile * 4 + q] : 0u;
    const uint2 *__restrict__ list = b.qlist + (size_t)q * b.cap + s;

    float T_ = 1.0f, D = 0.f, Wt = 0.f;
    float Cacc[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) Cacc[ch] = 0.f;
    uint32_t last = 0, lastj = 0;
    bool done = !inside;

    float4 r[4];
    if ((uint32_t)lane < nq) gather_entry<C>(list[lane], g, colors, r);
    for (uint32_t c0 = 0; c0 < nq; c0 += 64) {
        const int cnt = (int)min(64u, nq - c0);
        __builtin_amdgcn_wave_barrier();
        if (lane < cnt) {
#pragma unroll
            for (int v = 0; v < NV; ++v) s_e[lane][v] = r[v];
        }
        if (c0 + 64 + (uint32_t)lane < nq) gather_entry<C>(list[c0 + 64 + lane], g, colors, r);   // prefetch
        __builtin_amdgcn_wave_barrier();
        if (__ballot(!done) == 0) break;   // every pixel of the quadrant is saturated
        int t = 0;
        do {
            // Branch-free body (selects, not exec-mask branches): lanes that do not take the entry
            // blend with weight 0, which leaves their accumulators bit-identical.
            const float4 ea = s_e[t][0], eb = s_e[t][1], ec = s_e[t][2];
            const float dx = ea.x - pxf, dy = ea.y - pyf;
            const float power = -0.5f * ((ea.z * dx) * dx + (eb.x * dy) * dy) - (ea.w * dx) * dy;
            const float alpha = fminf(0.99f, eb.y * det_expf(power));
            const float test_T = T_ * (1.0f - alpha);
            const bool valid = (!done) & (power <= 0.0f) & (alpha >= 1.0f / 255.0f);
            const bool stop = valid & (test_T < 0.0001f);
            const bool contrib = valid & (!stop);
            const float w = contrib ? alpha * T_ : 0.f;
            Cacc[0] = __builtin_fmaf(ec.x, w, Cacc[0]);
            Cacc[1] = __builtin_fmaf(ec.y, w, Cacc[1]);
            Cacc[2] = __builtin_fmaf(ec.z, w, Cacc[2]);
            if (C > 3) {
                const float4 ed = s_e[t][NV - 1];
                Cacc[3] = __builtin_fmaf(ec.w, w, Cacc[3]);
                Cacc[C > 4 ? 4 : 0] = __builtin_fmaf(ed.x, w, Cacc[C > 4 ? 4 : 0]);
                Cacc[C > 5 ? 5 : 0] = __builtin_fmaf(ed.y, w, Cacc[C > 5 ? 5 : 0]);
            }
            D = __builtin_fmaf(eb.z, w, D);
            Wt = Wt + w;
            T_ = contrib ? test_T : T_;
            last = contrib ? __float_as_uint(eb.w) + 1u : last;
            lastj = contrib ? c0 + (uint32_t)t + 1u : lastj;
            done = done | stop;
        } while (++t < cnt && __ballot(!done) != 0);
    }
    if (inside) {
        const size_t P = (size_t)vp.H * vp.W;
        const size_t pid = (size_t)py * vp.W + px;
        im.final_T[pid] = T_;
        im.n_contrib[pid] = last;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) out_color[ch * P + pid] = __builtin_fmaf(T_, vp.bg[ch], Cacc[ch]);
        out_depth[pid] = D;
        out_alpha[pid] = Wt;
    }
    const uint32_t wj = wave_max_u32(lastj), wk = wave_max_u32(last);
    if (lane == 0) {
        g.qdone[tile * 4 + q] = wj;
        g.qkmax[tile * 4 + q] = wk;
    }
}

// ---------------------------------------------------------------------------------------- B1
// Packed 64-lane sums of NV (12 or 16) per-lane values.  After it, register out[r], lanes of row
// rho (lane >> 4) all hold the total of value 4r + {0,2,1,3}[rho].
template <int NV>
__device__ __forceinline__ void wave_reduce_packed(const float (&v)[NV], float (&out)[NV / 4])
{
    float p[NV / 2];
#pragma unroll
    for (int m = 0; m < NV / 2; ++m) {
        const u2v x = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * m]), __float_as_uint(v[2 * m + 1]), false, false);
        p[m] = __uint_as_float(x.x) + __uint_as_float(x.y);
    }
#pragma unroll
    for (int r = 0; r < NV / 4; ++r) {
        const u2v x = __builtin_amdgcn_permlane16_swap(__float_as_uint(p[2 * r]), __float_as_uint(p[2 * r + 1]), false, false);
        float o = __uint_as_float(x.x) + __uint_as_float(x.y);
        o = dpp_add<0xB1>(o);    // quad_perm [1,0,3,2]
        o = dpp_add<0x4E>(o);    // quad_perm [2,3,0,1]
        o = dpp_add<0x141>(o);   // row_half_mirror
        o = dpp_add<0x140>(o);   // row_mirror
        out[r] = o;
    }
}

template <int C>
__global__ __launch_bounds__(64) void k_render_bwd(BatchDesc d)
{
    constexpr int NV = StagedN<C>::kVec;
    constexpr int GS = (C <= 3) ? 12 : 16;   // == grad_stride(C)
    __shared__ float4 s_e[64][NV];
    const ViewCtx c = resolve(d, blockIdx.y);
    const ViewParams &vp = c.vp;
    const float *__restrict__ colors = c.colors;
    const GeomPtrs &g = c.g;
    const BinPtrs &b = c.b;
    const uint32_t cap = c.cap;
    const ImgPtrs &im = c.im;
    const float *__restrict__ dL_dcolor = c.dL_dcolor, *__restrict__ dL_ddepth = c.dL_ddepth,
                             *__restrict__ dL_dalpha = c.dL_dalpha;
    float *__restrict__ dLq = c.dLq;
    const int T = c.T;
    int tile, q;
    block_to_quadrant(blockIdx.x, tile, q);
    if (tile >= T) return;
    const int lane = threadIdx.x;
    const int tx = tile % vp.gx, ty = tile / vp.gx;
    const int px = tx * kTile + (q & 1) * 8 + (lane & 7);
    const int py = ty * kTile + (q >> 1) * 8 + (lane >> 3);
    const bool inside = px < vp.W && py < vp.H;
    const float pxf = (float)px, pyf = (float)py;

    const uint32_t s = g.tile_start[tile];
    const uint32_t nd = (s < cap) ? g.qdone[tile * 4 + q] : 0u;
    if (nd == 0) return;
    const uint2 *__restrict__ list = b.qlist + (size_t)q * b.cap + s;
    float *__restrict__ rec_base = dLq + ((size_t)q * b.cap + s) * GS;

    const size_t P = (size_t)vp.H * vp.W;
    const size_t pid = (size_t)py * vp.W + px;
    float T_final = 0.f, gD = 0.f, gA = 0.f;
    float gCol[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) gCol[ch] = 0.f;
    uint32_t last = 0;
    if (inside) {
        T_final = im.final_T[pid];
        last = im.n_contrib[pid];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) gCol[ch] = dL_dcolor[ch * P + pid];
        if (dL_ddepth) gD = dL_ddepth[pid];
        if (dL_dalpha) gA = dL_dalpha[pid];
    }
    float bgdot = 0.f;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) bgdot += vp.bg[ch] * gCol[ch];
    const float Tb = T_final * bgdot;
    float T_ = T_final, S = 0.f;
    const float half_W = 0.5f * (float)vp.W, half_H = 0.5f * (float)vp.H;
    // which value this lane stores: register (lane & 15), memory slot 4*(lane&15) + {0,2,1,3}[row]
    const int my_reg = lane & 15, row = lane >> 4;
    const int my_slot = 4 * my_reg + ((row == 1) ? 2 : (row == 2) ? 1 : row);

    const int c_last = (int)((nd - 1) / 64) * 64;
    float4 r[4];
    if ((uint32_t)(c_last + lane) < nd) gather_entry<C>(list[c_last + lane], g, colors, r);
    for (int c0 = c_last; c0 >= 0; c0 -= 64) {
        const int cnt = (int)min(64u, nd - (uint32_t)c0);
        __builtin_amdgcn_wave_barrier();
        if (lane < cnt) {
#pragma unroll
            for (int v = 0; v < NV; ++v) s_e[lane][v] = r[v];
        }
        if (c0 >= 64) gather_entry<C>(list[c0 - 64 + lane], g, colors, r);   // prefetch the chunk in front
        __builtin_amdgcn_wave_barrier();
        for (int t = cnt - 1; t >= 0; --t) {
            const float4 ea = s_e[t][0], eb = s_e[t][1];
            const uint32_t k = __float_as_uint(eb.w);
            // Branch-free: lanes that do not take the entry contribute exact zeros.
            const float dx = ea.x - pxf, dy = ea.y - pyf;
            const float power = -0.5f * ((ea.z * dx) * dx + (eb.x * dy) * dy) - (ea.w * dx) * dy;
            const float Gr = det_expf(power);
            const float alpha = fminf(0.99f, eb.y * Gr);
            const bool contrib = (k < last) & (power <= 0.0f) & (alpha >= 1.0f / 255.0f);
            const float4 ec = s_e[t][2];
            float col[C];
            col[0] = ec.x; col[1] = ec.y; col[2] = ec.z;
            if (C > 3) {
                const float4 ed = s_e[t][NV - 1];
                col[3] = ec.w;
                col[C > 4 ? 4 : 0] = ed.x;
                col[C > 5 ? 5 : 0] = ed.y;
            }
            const float G = contrib ? Gr : 0.f;
            const float inv_om = __builtin_amdgcn_rcpf(1.f - alpha);
            const float Tn = T_ * inv_om;
            T_ = contrib ? Tn : T_;
            const float w = contrib ? alpha * Tn : 0.f;
            float V = gA + eb.z * gD;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) V = __builtin_fmaf(col[ch], gCol[ch], V);
            const float dL_da = contrib ? (Tn * V - (S + Tb) * inv_om) : 0.f;
            S = __builtin_fmaf(V, w, S);
            const float dL_dG = eb.y * dL_da;
            const float gdx = G * dx, gdy = G * dy;
            float v[GS];
            v[0] = dL_dG * (-gdx * ea.z - gdy * ea.w) * half_W;
            v[1] = dL_dG * (-gdy * eb.x - gdx * ea.w) * half_H;
            v[2] = -0.5f * gdx * dx * dL_dG;
            v[3] = -gdx * dy * dL_dG;
            v[4] = -0.5f * gdy * dy * dL_dG;
            v[5] = G * dL_da;
            v[6] = w * gD;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) v[7 + ch] = w * gCol[ch];
#pragma unroll
            for (int i = 7 + C; i < GS; ++i) v[i] = 0.f;
            float out[GS / 4];
            if (__ballot(contrib) != 0) {
                wave_reduce_packed<GS>(v, out);
            } else {
#pragma unroll
                for (int i = 0; i < GS / 4; ++i) out[i] = 0.f;
            }
            if (my_reg < GS / 4) {
                float val = out[0];
#pragma unroll
                for (int i = 1; i < GS / 4; ++i) val = (my_reg == i) ? out[i] : val;
                rec_base[(size_t)k * GS + my_slot] = val;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------- launchers
int launch_render_fwd(const BatchDesc &d, hipStream_t st)
{
    const int T = ((d.W + kTile - 1) / kTile) * ((d.H + kTile - 1) / kTile);
    if (T <= 0) return DM4D_OK;
    const int blocks = ((T + 7) / 8) * 8 * 4;
    ProfScope prof_(kKRenderFwd, st);
    if (d.C <= 3) hipLaunchKernelGGL(k_render_fwd<3>, dim3(blocks, d.B), dim3(64), 0, st, d);
    else hipLaunchKernelGGL(k_render_fwd<6>, dim3(blocks, d.B), dim3(64), 0, st, d);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int launch_render_bwd(const BatchDesc &d, hipStream_t st)
{
    const int T = ((d.W + kTile - 1) / kTile) * ((d.H + kTile - 1) / kTile);
    if (T <= 0) return DM4D_OK;
    const int blocks = ((T + 7) / 8) * 8 * 4;
    ProfScope prof_(kKRenderBwd, st);
    if (d.C <= 3) hipLaunchKernelGGL(k_render_bwd<3>, dim3(blocks, d.B), dim3(64), 0, st, d);
    else hipLaunchKernelGGL(k_render_bwd<6>, dim3(blocks, d.B), dim3(64), 0, st, d);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

// ---- self-test of the packed reduction (exported through dm4d_selftest_wave_reduce) ----------
__global__ void k_selftest_reduce(const float *__restrict__ in /* [16][64] */, float *__restrict__ out /* [16] */)
{
    const int lane = threadIdx.x;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = in[i * 64 + lane];
    float o[4];
    wave_reduce_packed<16>(v, o);
    const int my_reg = lane & 15, row = lane >> 4;
    const int my_slot = 4 * my_reg + ((row == 1) ? 2 : (row == 2) ? 1 : row);
    if (my_reg < 4) {
        float val = o[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) val = (my_reg == i) ? o[i] : val;
        out[my_slot] = val;
    }
}
int launch_selftest_reduce(const float *in, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_selftest_reduce, dim3(1), dim3(64), 0, st, in, out);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // namespace dm4d
