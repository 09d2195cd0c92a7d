// raster_render.hip -- alpha-compositing kernels of the tile rasterizer (gfx950, wave64).
//
// Execution unit = ONE WAVE per 8x8-pixel quadrant of a 16x16 tile; inside the wave every 16-lane
// DPP row owns one 4x4-pixel CELL and walks that cell's own depth-sorted list.  Tiles stay 16x16
// (the reference's BLOCK_X x BLOCK_Y) so tile lists, keys, n_contrib and final_T keep their upstream
// meaning, but K4 splits every tile's sorted list into sixteen cell lists using the exact bound of
// each splat's alpha >= 1/255 ellipse (cell_bands, raster.h).  A mesh-bound splat is ~1 px wide:
// measured on the bench scene, 41 % of the (entry, pixel) pairs a cell list visits contribute,
// against 20 % for 8x8 lists and 8 % for whole-tile lists, and the four rows of a wave finish
// within 11 % of each other -- 1.75x fewer wave iterations than one list per quadrant.  Culled
// (splat, pixel) pairs are exactly ones the reference `continue`s on: results are unchanged.
// One wave per workgroup means no barriers at all and early exit at cell granularity.
//
// Forward  (K5): each row gathers 16 entries of its list (coalesced list read, L2-resident
//                attribute gathers) into wave-private LDS; the lanes of the row then walk the
//                chunk with row-broadcast LDS reads (4 distinct addresses per instruction).
//                Front-to-back blend; the wave stops when all 64 pixels are saturated (ballot).
// Backward (B1): back-to-front over the entries the forward consumed.  The 16-pixel sums of the
//                10 (13 with 6 colour channels) per-entry gradients go through LDS TRANSPOSED:
//                every lane writes its values to [value][lane], lane i of a row reads the 16
//                floats of value i of its row and adds them -- 7 packed adds on the VALU instead
//                of a 52-step DPP butterfly -- and the row stores one contiguous record.
//                Records are indexed by (Gaussian, cell): no floating-point atomics, gradients are
//                bit-reproducible, and B2 reads every Gaussian's records as one contiguous block.
//
// blockIdx -> (tile, quadrant) keeps the 4 quadrants of a tile on ONE XCD (blocks are dispatched
// round-robin over the 8 XCDs), so the attribute gathers of neighbouring quadrants share an L2.
//
// Replaces renderCUDA fwd/bwd of the un-vendored diff-gaussian-rasterization (ashawkey fork:
// extra depth and alpha channels) used at
// custom/threestudio-dreammesh4d/renderer/diff_sugar_rasterizer_temporal.py:169-178,202-211.
// C = 6 blends the RGB pass and the normal pass of one view together (same geometry, :202-211).
#include <stdlib.h>

#include "common.h"
#include "raster.h"
#include "raster_fwd.h"       // the forward blend of one wave (shared with the tile sort: raster_bin.hip)

namespace dm4d {

// ---- optional per-wave trace (dm4d_debug_trace): {start, end} in 100 MHz wall-clock ticks, HW_ID, XCC_ID ----
__device__ uint64_t *g_trace = nullptr;
__device__ uint32_t g_min_work = 0;   // debug: waves with shorter lists exit at once (isolates the long ones)
struct WaveTrace {
    uint64_t *buf, t0;
    __device__ __forceinline__ WaveTrace() : buf(g_trace), t0(0) { if (buf) t0 = wall_clock64(); }
    __device__ __forceinline__ void done(uint32_t work) const
    {
        if (!buf || threadIdx.x != 0) return;
        uint64_t *r = buf + 4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        r[0] = t0;
        r[1] = wall_clock64();
        r[2] = ((uint64_t)xcc << 32) | hw;
        r[3] = work;
    }
};
int set_trace_buffer(void *dev_ptr, uint32_t min_work)
{
    uint64_t *p = (uint64_t *)dev_ptr;
    DM4D_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &p, sizeof(p)));
    DM4D_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_min_work), &min_work, sizeof(min_work)));
    return DM4D_OK;
}

// ---------------------------------------------------------------------------------------- K5
template <int C>
__global__ __launch_bounds__(64) void k_render_fwd(BatchDesc d)
{
    __shared__ __attribute__((aligned(16))) float s_p[4 * kRowFloats];
    const WaveTrace trace;
    int view, tile, q;
    if (!block_to_quadrant(d, blockIdx.x, view, tile, q)) { trace.done(0); return; }
    render_fwd_wave<C>(d, view, tile, q, (int)threadIdx.x, s_p, trace, g_min_work);
}

// ---------------------------------------------------------------------------------------- K5 (long cells of the large tiles)
// One wave per cell of `earlylist` (raster.h: the long cells of the tiles the LARGE sort variant handled -- the
// silhouette tiles, whose lists are the longest of the launch).  Lane = (entry slot r = lane >> 4, pixel p = lane & 15
// of the cell): the four rows evaluate the alphas of four CONSECUTIVE list entries for the same 16 pixels, hand them to
// row 0 through 256 bytes of LDS, and row 0 runs the sequential blend -- the same operations on the same values in the
// same order as k_render_fwd, so the image is bit-identical.  The point is WHEN it runs: the large sort variant is
// done ~90 us before the small one, so these cells -- the tail the regular kernel used to end with -- are blended
// beside the small variant's sort, on their own stream (regular kernel 252 -> 204 us; run beside the regular kernel
// instead, the same code gained nothing: 247 us against 219).
// Staging: 64 entries per chunk, one per lane: [0..5] x y A B C opacity | [8..13] colours | [14] depth | [15] 1
// Four cells per workgroup (one per wave, no workgroup barrier): the long-running waves then share few CUs instead of
// taking one SIMD on most of them, which slowed the barrier-coupled workgroups of K4's small variant running beside.
// x of row r (16 lanes) -> o[r] in every row, same lane of the row.  v_permlane16_swap exchanges the odd rows of its
// first operand with the even rows of the second, v_permlane32_swap the upper half of the first with the lower half
// of the second (gfx950).
__device__ __forceinline__ void rows_allgather(float x, float (&o)[4])
{
    const unsigned a = __float_as_uint(x);
    const auto s16 = __builtin_amdgcn_permlane16_swap(a, a, false, false);          // [x0 x0 x2 x2], [x1 x1 x3 x3]
    const auto e = __builtin_amdgcn_permlane32_swap(s16[0], s16[0], false, false);  // x0 everywhere, x2 everywhere
    const auto f = __builtin_amdgcn_permlane32_swap(s16[1], s16[1], false, false);  // x1, x3
    o[0] = __uint_as_float(e[0]);
    o[1] = __uint_as_float(f[0]);
    o[2] = __uint_as_float(e[1]);
    o[3] = __uint_as_float(f[1]);
}
template <int C>
__global__ __launch_bounds__(256) void k_render_fwd_long(BatchDesc d)
{
    __shared__ __attribute__((aligned(16))) float s_e_all[4][64 * 16];
    float *s_e = s_e_all[threadIdx.x >> 6];
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
    const int view = (int)(wave % (uint32_t)d.B);
    const uint32_t first = wave / (uint32_t)d.B, step = n_waves / (uint32_t)d.B;
    if (first >= step) return;                 // the few waves past the last full round of views
    const ViewCtx c = resolve(d, view);
    const ViewParams &vp = c.vp;
    const float *__restrict__ colors = c.colors;
    const GeomPtrs &g = c.g;
    const BinPtrs &b = c.b;
    const ImgPtrs &im = c.im;
    const int lane = threadIdx.x & 63, r = lane >> 4, p = lane & 15;
    const uint32_t n_long = min(g.counters[kCntLongEarly], (uint32_t)(c.T * kCells));
    if (first < n_long) __builtin_amdgcn_s_setprio(3);
    for (uint32_t it = first; it < n_long; it += step) {
        const uint32_t cellid = g.earlylist[it];
        const int tile = (int)(cellid / kCells), cell = (int)(cellid % kCells);
        const int tx = tile % vp.gx, ty = tile / vp.gx, q = cell >> 2, rw = cell & 3;
        const int px = tx * kTile + (q & 1) * 8 + (rw & 1) * 4 + (p & 3);
        const int py = ty * kTile + (q >> 1) * 8 + (rw >> 1) * 4 + (p >> 2);
        const bool inside = px < vp.W && py < vp.H;
        const float pxf = (float)px, pyf = (float)py;
        const uint32_t s = g.tile_start[tile];
        const uint32_t nr = g.ccount[cellid];
        const uint32_t *__restrict__ list = b.clist + (size_t)cell * b.cap + s;

        float T_ = 1.0f;
        f2v C01 = (f2v)(0.f), C23 = (f2v)(0.f), C45 = (f2v)(0.f), DW = (f2v)(0.f);
        uint32_t lastj = 0;
        bool done = !inside | (r != 0);      // the pixel state lives in row 0

        float4 e[4];
        zero_entry(e);
        if ((uint32_t)lane < nr) gather_entry<C>(list[lane], g, colors, e);
        uint32_t wnext = (64u + (uint32_t)lane < nr) ? list[64u + lane] : 0u;      // two-deep prefetch as in k_render_fwd
        for (uint32_t c0 = 0; c0 < nr; c0 += 64u) {
            const int cnt = (int)min(64u, nr - c0);
            __builtin_amdgcn_wave_barrier();
            {
                float *se = s_e + lane * 16;
                *reinterpret_cast<float4 *>(se) = e[0];                                              // x y A B
                *reinterpret_cast<float4 *>(se + 4) = make_float4(e[1].x, e[1].y, 0.f, 0.f);        // C opacity
                *reinterpret_cast<float4 *>(se + 8) = e[2];                                          // colours 0..3
                *reinterpret_cast<float4 *>(se + 12) = make_float4(e[3].x, e[3].y, e[1].z, 1.0f);   // colours 4 5, depth, 1
            }
            zero_entry(e);
            const uint32_t wcur = wnext;
            if (c0 + 128u + (uint32_t)lane < nr) wnext = list[c0 + 128u + lane];
            if (c0 + 64u + (uint32_t)lane < nr) gather_entry<C>(wcur, g, colors, e);   // prefetch
            __builtin_amdgcn_wave_barrier();
            if (__ballot(!done) == 0) break;
            for (int t = 0; t < cnt; t += 4) {
                // ---- the four rows: alpha of entry t + r at pixel p (padding entries are inert: opacity 0) ----
                const float4 ga = *reinterpret_cast<const float4 *>(s_e + (t + r) * 16);
                const float4 gb = *reinterpret_cast<const float4 *>(s_e + (t + r) * 16 + 4);
                const float dx = ga.x - pxf, dy = ga.y - pyf;
                const float power = -0.5f * ((ga.z * dx) * dx + (gb.x * dy) * dy) - (ga.w * dx) * dy;
                const float alpha_r = fminf(0.99f, gb.y * det_expf(power));
                float al[4];     // the four alphas of the pixel, in every row (lane swaps, no LDS round trip)
                rows_allgather(((power <= 0.0f) & (alpha_r >= 1.0f / 255.0f)) ? alpha_r : -1.0f, al);
                // ---- row 0: the sequential blend of the four entries ----
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const float *se = s_e + (t + h) * 16;
                    const f4v e0 = *reinterpret_cast<const f4v *>(se + 8), e1 = *reinterpret_cast<const f4v *>(se + 12);
                    const float alpha = al[h];
                    const float test_T = T_ * (1.0f - alpha);
                    const bool valid = (!done) & (alpha >= 0.0f);
                    const bool stop = valid & (test_T < 0.0001f);
                    const bool contrib = valid & (!stop);
                    const float w = contrib ? alpha * T_ : 0.f;
                    const f2v ww = (f2v)(w);
                    C01 = __builtin_elementwise_fma(e0.xy, ww, C01);
                    C23 = __builtin_elementwise_fma(e0.zw, ww, C23);
                    if (C > 3) C45 = __builtin_elementwise_fma(e1.xy, ww, C45);
                    DW = __builtin_elementwise_fma(e1.zw, ww, DW);
                    T_ = contrib ? test_T : T_;
                    lastj = contrib ? c0 + (uint32_t)(t + h) + 1u : lastj;
                    done = done | stop;
                }
                __builtin_amdgcn_wave_barrier();
                if (__ballot(!done) == 0) break;
            }
        }
        if (inside && r == 0) {
            const size_t P = (size_t)vp.H * vp.W;
            const size_t pid = (size_t)py * vp.W + px;
            im.final_T[pid] = T_;
            im.n_contrib[pid] = lastj;
            const float Cacc[6] = {C01.x, C01.y, C23.x, C23.y, C45.x, C45.y};
#pragma unroll
            for (int ch = 0; ch < C; ++ch) c.out_color[ch * P + pid] = __builtin_fmaf(T_, vp.bg[ch], Cacc[ch]);
            c.out_depth[pid] = DW.x;
            c.out_alpha[pid] = DW.y;
        }
        const uint32_t wj = row_max_u32(lastj);
        if (lane == 0) g.cdone[cellid] = wj;
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------- B1
// ENTRY-PARALLEL blend backward.  A lane owns one LIST ENTRY (a Gaussian of a 4x4-pixel cell's depth-sorted list) and walks
// the cell's 16 pixels serially, so the 16-pixel sums of an entry's gradient values are plain FMAs into the lane's own
// registers: no cross-lane reduction per value, no LDS, and the lane ends up holding the entry's complete
// (Gaussian, cell) record.  What IS sequential along the list -- the transmittance in front of an entry and the
// colour sum behind it -- becomes two scans ACROSS LANES per pixel (DPP row shifts: on the VALU, no LDS):
//     P_inc(l) = prod_{i behind or at l} (1 - alpha_i)      T_before(l) = T_behind_chunk / P_inc(l)
//     S_exc(l) = sum_{i behind l} V_i w_i                   dL/dalpha_l = T_before V_l - (S_exc + S_behind_chunk) / (1 - alpha_l)
// with the per-pixel carries (T, S behind the chunk) handed from chunk to chunk, back to front.  The per-pixel
// quantities an iteration needs (upstream gradients, n_contrib, the carries) live in the lane that owns that pixel
// (lane p of the row holds pixel p) and reach the other lanes as DPP `row_newbcast:p` operands of the VOP2
// instructions that consume them -- a broadcast costs no instruction.
//   regular blocks: wave = 8x8 quadrant, DPP row = cell, lane = entry of a 16-entry chunk of the row's list;
//   wide blocks:    wave = ONE cell with a long list (>= kWideBwd entries, K4's `longlist`), lane = entry of a
//                   64-entry chunk; the scans cross the rows with row_bcast:15 / :31.  A 1200-entry silhouette cell is
//                   19 chunks instead of 75, and the launch no longer ends with it.
// Lanes hold the chunk REVERSED (lane 0 = the entry farthest back), so "behind" = lower lanes and the scans are the
// plain prefix scans row_shr / row_bcast were made for.
// The record a lane writes is what B2 (k_gather_bwd) expects: moments sum q (dx, dy, dx^2, dx dy, dy^2), q = dL/dG G
// (dL/dmean2D and dL/dconic are linear maps of them, applied once per Gaussian in B2), opacity * dL/dopacity, dL/ddepth,
// dL/dcolour.
// Deterministic: fixed order everywhere, no atomics.
#define DM4D_RM " row_mask:0xf bank_mask:0xf\n\t"
// inclusive prefix scans over the lanes of a row (WIDE: of the wave) of two independent values at once: the two chains
// are interleaved so that a DPP read of a freshly written VGPR has its two wait states (inline asm is opaque to the
// compiler's hazard recogniser; the leading s_nop covers the producer of the inputs)
// a, b: in place; ea, eb (preset to the identity by the caller): the same scans EXCLUSIVE (the value of the lane below;
// the first lane of the row / wave keeps the identity)
#define DM4D_SCAN2(OP, SHIFT, BC)                                                                                                  \
    asm volatile("s_nop 1\n\t"                                                                                                     \
                 OP " %0, %0, %0 row_shr:1" DM4D_RM OP " %1, %1, %1 row_shr:1" DM4D_RM "s_nop 0\n\t"                               \
                 OP " %0, %0, %0 row_shr:2" DM4D_RM OP " %1, %1, %1 row_shr:2" DM4D_RM "s_nop 0\n\t"                               \
                 OP " %0, %0, %0 row_shr:4" DM4D_RM OP " %1, %1, %1 row_shr:4" DM4D_RM "s_nop 0\n\t"                               \
                 OP " %0, %0, %0 row_shr:8" DM4D_RM OP " %1, %1, %1 row_shr:8" DM4D_RM "s_nop 0\n\t"                               \
                 BC                                                                                                                \
                 "v_mov_b32_dpp %2, %0 " SHIFT DM4D_RM "v_mov_b32_dpp %3, %1 " SHIFT DM4D_RM "s_nop 1\n\t"                        \
                 : "+v"(a), "+v"(b), "+v"(ea), "+v"(eb))
#define DM4D_BC(OP)                                                                                                                \
    OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" OP " %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"     \
    "s_nop 0\n\t"                                                                                                                  \
    OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" OP " %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"     \
    "s_nop 0\n\t"
template <bool WIDE>
__device__ __forceinline__ void scan2_mul(float &a, float &b, float &ea, float &eb)
{
    if (WIDE) DM4D_SCAN2("v_mul_f32_dpp", "wave_shr:1", DM4D_BC("v_mul_f32_dpp"));
    else DM4D_SCAN2("v_mul_f32_dpp", "row_shr:1", "");
}
template <bool WIDE>
__device__ __forceinline__ void scan2_add(float &a, float &b, float &ea, float &eb)
{
    if (WIDE) DM4D_SCAN2("v_add_f32_dpp", "wave_shr:1", DM4D_BC("v_add_f32_dpp"));
    else DM4D_SCAN2("v_add_f32_dpp", "row_shr:1", "");
}
#undef DM4D_SCAN2
#undef DM4D_BC

// Per-pixel state of the backward: a table in LDS, one 48-byte line per pixel of the row's cell (WIDE: of the wave's cell),
//   C = 6: {g0 g1 g2 g3 | g4 g5 gD gA | T S last -}      C = 3: {g0 g1 g2 gD | gA T S last}
// T, S: the carries (transmittance behind the current chunk; sum V w behind it + T_final bg.g), rewritten by the lane that
// holds the chunk's FRONT entry.  Every lane of a row reads the SAME line per pixel iteration (an LDS broadcast read):
// the values arrive in plain VGPRs and feed full-rate VALU instructions -- as DPP row_newbcast operands (the values
// kept in "lane p of the row") every consumer ran at the DPP half rate (measured: tools/ubench/valu.hip, 4.2 against
// 2.4 cycles per wave instruction).
// LEAN == 3 (round 5): a 6-channel FORWARD whose channels 3..5 (the normal pass) receive no gradient -- the shipped dynamic
// configuration has every normal weight at 0 (C/configs/sugar_dynamic_dg.yaml:145-157), so in the reference autograd never enters
// the normal pass's backward: the table line is the 3-channel one, V carries three colour terms, a record five moments.
template <int C, int LEAN = 0> struct PixTab { static constexpr int kLine = (C > 3 && LEAN != 3) ? 12 : 8; };

// one list entry, held by the lane that owns it
template <int C>
struct EntryRegs {
    float dx[4], dy[4];          // x - (cell x0 + i), y - (cell y0 + j): bit-identical to the forward's xy - pixf
    float Adx2[4], Cdy2[4], Bdx[4];
    float o, dep;
    float c[C];
    uint32_t k;
};

// Pixels P and P + 1 of the cell for the lane's entry (two independent chains side by side: the dependent DPP steps of
// one hide behind the other).  acc: the lane's record.  tab: the row's (WIDE: wave's) pixel table.
// running sums of the moments of q = dL/dG G over the cell's pixels (i, j in 0..3): the current row's and the cell's
struct Moments { float r0, r1, r2, S0, Si, Sii, Sj, Sjj, Sij; };
template <int C, int LEAN, bool WIDE, int P>
__device__ __forceinline__ void pixel_pair(const EntryRegs<C> &e, float (&acc)[13], Moments &mo, const bool front_lane, float *tab)
{
    constexpr int LN = PixTab<C, LEAN>::kLine;
    constexpr bool kSix = C > 3 && LEAN != 3;         // gradients on six colour channels
    float pw[2], G[2], araw[2], a[2], am[2], om[2], Pinc[2], Pexc[2], V[2], Tb[2], inv_om[2], w[2], Sinc[2], Sexc[2], Stot[2];
    float g[2][6], gD[2], gA[2], T[2], S[2];
    uint32_t last[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float4 *ln = reinterpret_cast<const float4 *>(tab + (P + h) * LN);
#if defined(DM4D_BWD_PROBE) && (DM4D_BWD_PROBE & 2)      // TIMING PROBE: no table reads (values from registers)
        if constexpr (kSix) {
            const float f_ = e.o + (float)(P + h);
            g[h][0] = f_; g[h][1] = e.dep; g[h][2] = f_ + 1.f; g[h][3] = f_ * 0.5f; g[h][4] = e.dep + 2.f; g[h][5] = f_ - 1.f; gD[h] = 0.f; gA[h] = f_;
            T[h] = 0.5f; S[h] = e.dep; last[h] = 0xFFFFu;
            (void)ln;
            continue;
        }
#endif
        if constexpr (kSix) {
            const float4 q2 = ln[2], q0 = ln[0], q1 = ln[1];
            g[h][0] = q0.x; g[h][1] = q0.y; g[h][2] = q0.z; g[h][3] = q0.w; g[h][4] = q1.x; g[h][5] = q1.y; gD[h] = q1.z; gA[h] = q1.w;
            T[h] = q2.x; S[h] = q2.y; last[h] = __float_as_uint(q2.z);
        } else {
            const float4 q1 = ln[1], q0 = ln[0];
            g[h][0] = q0.x; g[h][1] = q0.y; g[h][2] = q0.z; gD[h] = q0.w; gA[h] = q1.x; T[h] = q1.y; S[h] = q1.z; last[h] = __float_as_uint(q1.w);
            g[h][3] = g[h][4] = g[h][5] = 0.f;
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = (P + h) & 3, j = (P + h) >> 2;
        const float u = e.Adx2[i] + e.Cdy2[j];
        const float wv = e.Bdx[i] * e.dy[j];
        pw[h] = __builtin_fmaf(-0.5f, u, -wv);      // == (-0.5f * u) - wv of the forward: the product by -0.5 is exact
    }
    // alpha: the hardware exp2 (1 ulp) instead of the contract's det_expf (12 instructions) -- but the DECISIONS of the
    // forward (power <= 0, alpha >= 1/255) must be reproduced exactly, so a pair within 5e-6 (relative) of the alpha
    // threshold or 1e-6 of power == 0 sends the wave through det_expf (rare: a wave-uniform branch)
#ifdef DM4D_BWD_MASKPROBE
    // TIMING PROBE ONLY (tools/build_variant.sh): what the backward would cost if the forward handed it a 16-bit contributor mask per
    // list entry -- one bit test per pixel instead of the two compares, the select chain and the exp guard (results are NOT the reference's)
    G[0] = __builtin_amdgcn_exp2f(pw[0] * 0x1.715476p+0f);
    G[1] = __builtin_amdgcn_exp2f(pw[1] * 0x1.715476p+0f);
    araw[0] = e.o * G[0]; araw[1] = e.o * G[1];
    am[0] = __uint_as_float(__float_as_uint(araw[0]) & (uint32_t)__builtin_amdgcn_sbfe((int)e.k, P, 1));
    am[1] = __uint_as_float(__float_as_uint(araw[1]) & (uint32_t)__builtin_amdgcn_sbfe((int)e.k, P + 1, 1));
    a[0] = __builtin_amdgcn_fmed3f(am[0], 0.0f, 0.99f); a[1] = __builtin_amdgcn_fmed3f(am[1], 0.0f, 0.99f);
    (void)last;
#else
    constexpr float kThr = 1.0f / 255.0f;
    G[0] = __builtin_amdgcn_exp2f(pw[0] * 0x1.715476p+0f);
    G[1] = __builtin_amdgcn_exp2f(pw[1] * 0x1.715476p+0f);
    araw[0] = e.o * G[0]; araw[1] = e.o * G[1];
    bool pwok0 = true, pwok1 = true;
    {
        const float d = fminf(fminf(fabsf(araw[0] - kThr), fabsf(araw[1] - kThr)), fminf(-pw[0], -pw[1]) * 0.02f);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(d < 2.0e-8f) != 0ull, 0)) {
            det_expf_n<2>(pw, G);
            araw[0] = e.o * G[0]; araw[1] = e.o * G[1];
            pwok0 = pw[0] <= 0.0f; pwok1 = pw[1] <= 0.0f;
        }
    }
    {
        const bool c0 = (e.k < last[0]) & pwok0 & (araw[0] >= kThr), c1 = (e.k < last[1]) & pwok1 & (araw[1] >= kThr);
        am[0] = c0 ? araw[0] : 0.f; am[1] = c1 ? araw[1] : 0.f;        // o G of the pairs that contribute (the straight-through factor of q)
        // min(0.99, am) for am >= 0 (never NaN: o G with finite o, G in [0, 1]) as ONE v_med3_f32: fminf costs a canonicalising v_max
        // in front of its v_min, and this kernel's duration is its VALU instruction count (profiles/r03_pmc_sq.md)
        a[0] = __builtin_amdgcn_fmed3f(am[0], 0.0f, 0.99f); a[1] = __builtin_amdgcn_fmed3f(am[1], 0.0f, 0.99f);
    }
#endif
    om[0] = 1.f - a[0]; om[1] = 1.f - a[1];
    Pinc[0] = om[0]; Pinc[1] = om[1];
    Pexc[0] = Pexc[1] = 1.0f;
#if defined(DM4D_BWD_PROBE) && (DM4D_BWD_PROBE & 4)      // TIMING PROBE: no scans
    Pexc[0] = Pinc[0] * 0.5f; Pexc[1] = Pinc[1] * 0.5f;
#else
    scan2_mul<WIDE>(Pinc[0], Pinc[1], Pexc[0], Pexc[1]);     // product over the entries behind, this one included / excluded
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float v = __builtin_fmaf(e.c[0], g[h][0], gA[h]);
        v = __builtin_fmaf(e.c[1], g[h][1], v);
        v = __builtin_fmaf(e.c[2], g[h][2], v);
        if (kSix) {
            v = __builtin_fmaf(e.c[C > 3 ? 3 : 0], g[h][3], v);
            v = __builtin_fmaf(e.c[C > 3 ? 4 : 0], g[h][4], v);
            v = __builtin_fmaf(e.c[C > 3 ? 5 : 0], g[h][5], v);
        }
        V[h] = LEAN >= 2 ? v : __builtin_fmaf(e.dep, gD[h], v);      // (LEAN == 2: no depth gradient -- the product would be an exact 0)
    }
#if defined(DM4D_BWD_PROBE) && (DM4D_BWD_PROBE & 8)      // TIMING PROBE: no rcp
    const float R0 = Pinc[0] + 1.f, R1 = Pinc[1] + 1.f;
#else
    const float R0 = __builtin_amdgcn_rcpf(Pinc[0]), R1 = __builtin_amdgcn_rcpf(Pinc[1]);
#endif
    Tb[0] = T[0] * R0; Tb[1] = T[1] * R1;                    // transmittance in front of the entry
    inv_om[0] = Pexc[0] * R0; inv_om[1] = Pexc[1] * R1;      // 1 / (1 - alpha)
    w[0] = a[0] * Tb[0]; w[1] = a[1] * Tb[1];
    Sinc[0] = V[0] * w[0]; Sinc[1] = V[1] * w[1];
    Sexc[0] = Sexc[1] = 0.0f;
#if defined(DM4D_BWD_PROBE) && (DM4D_BWD_PROBE & 4)
    Sexc[0] = Sinc[0] * 0.5f; Sexc[1] = Sinc[1] * 0.5f;
#else
    scan2_add<WIDE>(Sinc[0], Sinc[1], Sexc[0], Sexc[1]);
#endif
    Stot[0] = S[0] + Sexc[0]; Stot[1] = S[1] + Sexc[1];      // everything behind the entry
#if defined(DM4D_BWD_PROBE) && (DM4D_BWD_PROBE & 16)     // TIMING PROBE: no carry write
    if (false) {
#else
    if (front_lane) {      // the chunk in front starts from (T before, S from) this chunk's front entry
#endif
        float *t0 = tab + P * LN + (kSix ? 8 : 5), *t1 = t0 + LN;
        *reinterpret_cast<float2 *>(t0) = make_float2(Tb[0], S[0] + Sinc[0]);
        *reinterpret_cast<float2 *>(t1) = make_float2(Tb[1], S[1] + Sinc[1]);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = (P + h) & 3;
        const float dL_da = Tb[h] * V[h] - Stot[h] * inv_om[h];
        const float q = dL_da * am[h];                       // dL/dG G = (opacity dL/dalpha) G
        // Moments of q in the CELL's own pixel coordinates (i, j in 0..3), a row of four pixels at a time: r0 = sum q, r1 = sum i q,
        // r2 = sum i^2 q (the factors are literals: 2.5 instructions per pixel), folded into S0, Si, Sii, Sj, Sjj, Sij when the row is
        // complete (cell_pixels) and re-centred on the Gaussian once per chunk (finish_moments) -- 3.8 instructions per pixel where
        // sum q (dx, dy, dx^2, dx dy, dy^2) with the entry's own offsets took 7 (round 4; this kernel's time is its VALU instruction count)
        if (i == 0) mo.r0 = q;                               // (a row starts: no zero fill, no add)
        else mo.r0 += q;
        if (i == 1) { mo.r1 = q; mo.r2 = q; }
        if (i == 2) { mo.r1 = __builtin_fmaf(2.f, q, mo.r1); mo.r2 = __builtin_fmaf(4.f, q, mo.r2); }
        if (i == 3) { mo.r1 = __builtin_fmaf(3.f, q, mo.r1); mo.r2 = __builtin_fmaf(9.f, q, mo.r2); }
        if constexpr (LEAN == 3) {       // no depth gradient, no gradient on the normal channels: the five moments are the record
        } else if constexpr (LEAN == 2) {       // no depth gradient: 8 values
            acc[5] = __builtin_fmaf(w[h], g[h][3], acc[5]);
            acc[6] = __builtin_fmaf(w[h], g[h][4], acc[6]);
            acc[7] = __builtin_fmaf(w[h], g[h][5], acc[7]);
        } else if constexpr (LEAN == 1) {
            acc[5] = __builtin_fmaf(w[h], gD[h], acc[5]);
            acc[6] = __builtin_fmaf(w[h], g[h][3], acc[6]);
            acc[7] = __builtin_fmaf(w[h], g[h][4], acc[7]);
            acc[8] = __builtin_fmaf(w[h], g[h][5], acc[8]);
        } else {
            // (sum q = opacity * dL/dopacity, B2 divides by the opacity: it is S0 of the moments, finish_moments)
            acc[6] = __builtin_fmaf(w[h], gD[h], acc[6]);
#pragma unroll
            for (int ch = 0; ch < C; ++ch) acc[7 + ch] = __builtin_fmaf(w[h], g[h][ch], acc[7 + ch]);
        }
    }
    // pin the pair's accumulations HERE: volatile asm statements keep their order, so without this the compiler sinks the
    // tails of all eight pairs below the last scan (they only feed `acc`) and keeps ~8 values per pair alive until then
    // (147 VGPRs instead of ~80)
    asm volatile("" : "+v"(mo.r0), "+v"(mo.r1), "+v"(mo.r2), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]));      // (pins the pair's tail here, see above)
    if (!LEAN) asm volatile("" : "+v"(acc[9]), "+v"(acc[C > 3 ? 10 : 9]), "+v"(acc[C > 3 ? 11 : 9]), "+v"(acc[C > 3 ? 12 : 9]));
    if constexpr ((P & 3) == 2) {      // the row j = P >> 2 of the cell is complete
        constexpr int j = P >> 2;
        mo.S0 += mo.r0; mo.Si += mo.r1; mo.Sii += mo.r2;
        if (j == 1) { mo.Sj += mo.r0; mo.Sjj += mo.r0; mo.Sij += mo.r1; }
        if (j > 1) {
            mo.Sj = __builtin_fmaf((float)j, mo.r0, mo.Sj);
            mo.Sjj = __builtin_fmaf((float)(j * j), mo.r0, mo.Sjj);
            mo.Sij = __builtin_fmaf((float)j, mo.r1, mo.Sij);
        }
        asm volatile("" : "+v"(mo.S0), "+v"(mo.Si), "+v"(mo.Sii), "+v"(mo.Sj), "+v"(mo.Sjj), "+v"(mo.Sij));
    }
}
#undef DM4D_RM

// the 16 pixels of the cell for the lane's entry; acc[0..4] (and acc[5] of the full records) come out of the moments at the end:
// with dx_i = X - i, dy_j = Y - j (X = e.dx[0], Y = e.dy[0]: the offsets of a cell's pixels differ by exact integers, load_entry)
//   sum q dx = X S0 - Si,  sum q dy = Y S0 - Sj,  sum q dx^2 = X (X S0 - 2 Si) + Sii,  sum q dx dy = X (Y S0 - Sj) - Y Si + Sij,
//   sum q dy^2 = Y (Y S0 - 2 Sj) + Sjj
template <int C, int LEAN, bool WIDE>
__device__ __forceinline__ void cell_pixels(const EntryRegs<C> &e, float (&acc)[13], const bool front_lane, float *tab)
{
    Moments mo = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#if defined(DM4D_BWD_PROBE) && (DM4D_BWD_PROBE & 128)    // TIMING PROBE: no pixel work at all (gathers, zero fills and record stores only)
    acc[0] = e.o; acc[1] = e.dep; acc[2] = e.dx[0]; acc[3] = e.dy[1]; acc[4] = e.Bdx[2]; acc[5] = e.c[0]; acc[6] = e.c[C > 3 ? 3 : 1]; acc[7] = e.Adx2[3] + e.Cdy2[2];
    (void)front_lane; (void)tab;
    return;
#endif
    pixel_pair<C, LEAN, WIDE, 0>(e, acc, mo, front_lane, tab);
    pixel_pair<C, LEAN, WIDE, 2>(e, acc, mo, front_lane, tab);
    pixel_pair<C, LEAN, WIDE, 4>(e, acc, mo, front_lane, tab);
    pixel_pair<C, LEAN, WIDE, 6>(e, acc, mo, front_lane, tab);
    pixel_pair<C, LEAN, WIDE, 8>(e, acc, mo, front_lane, tab);
    pixel_pair<C, LEAN, WIDE, 10>(e, acc, mo, front_lane, tab);
    pixel_pair<C, LEAN, WIDE, 12>(e, acc, mo, front_lane, tab);
    pixel_pair<C, LEAN, WIDE, 14>(e, acc, mo, front_lane, tab);
    const float X = e.dx[0], Y = e.dy[0];
    acc[0] = __builtin_fmaf(X, mo.S0, -mo.Si);
    acc[1] = __builtin_fmaf(Y, mo.S0, -mo.Sj);
    acc[2] = __builtin_fmaf(X, __builtin_fmaf(X, mo.S0, -2.f * mo.Si), mo.Sii);
    acc[3] = __builtin_fmaf(X, __builtin_fmaf(Y, mo.S0, -mo.Sj), __builtin_fmaf(-Y, mo.Si, mo.Sij));
    acc[4] = __builtin_fmaf(Y, __builtin_fmaf(Y, mo.S0, -2.f * mo.Sj), mo.Sjj);
    if (!LEAN) acc[5] = mo.S0;
}

// backward record of a cell-list entry (raster.h, BinPtrs::clist): the Gaussian's first record + the rank K4 stored with the
// entry, or -- dense blocks too large for the rank field -- the cell's index in the block.  (gx, gy): the cell.
__device__ __forceinline__ uint32_t entry_slot(const uint32_t word, const GeomPtrs &g, const int gx, const int gy)
{
    const uint32_t gid = word & kGidMask;
    uint32_t rank = word >> kGidBits;
    if (rank == kRankBig) {
        const uint4 ci = g.cellinfo[gid];
        const int bx0 = (int)(ci.x & 0xFFFFu), by0 = (int)(ci.x >> 16), nbx = (int)(ci.y & 0xFFFFu);
        rank = (uint32_t)((gy - by0) * nbx + (gx - bx0));
    }
    return g.rec0[gid] + rank;
}

// gather the lane's entry (list position j of the cell list; `live` false: an inert entry) and derive what the pixel
// loop needs
template <int C, bool SLOT = true>
__device__ __forceinline__ void load_entry(EntryRegs<C> &e, uint32_t &slot, const bool live, const uint32_t *__restrict__ list,
                                           const uint32_t j, const GeomPtrs &g, const float *__restrict__ colors,
                                           const float cx0, const float cy0, const int gx, const int gy)
{
    // An inert entry (opacity 0: alpha = 0 whatever the rest) sits far off the cell with a unit conic, so that its power is
    // hugely negative: with an all-zero entry power == 0 lands exactly on the "within 1e-6 of power == 0" test of
    // pixel_pair and sends the WHOLE wave through the det_expf fallback for every pixel pair a padded chunk evaluates
    float x = -1.0e3f, y = 0.f, cA = 1.f, cB = 0.f, cC = 0.f;
    e.o = 0.f; e.dep = 0.f; e.k = 0xFFFFFFFFu; slot = 0xFFFFFFFFu;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) e.c[ch] = 0.f;
#if defined(DM4D_BWD_PROBE) && (DM4D_BWD_PROBE & 64)     // TIMING PROBE: no gathers of the entry's attributes
    if (live) {
        const uint32_t word = list[j];
        if (SLOT) slot = (word & kGidMask) * 3u + (word >> kGidBits);
        e.k = j;
        x = cx0 + (float)(word & 3u); y = cy0 + (float)((word >> 2) & 3u); cA = 0.3f; cB = 0.01f; cC = 0.2f; e.o = 0.8f; e.dep = 3.f;
        for (int ch = 0; ch < C; ++ch) e.c[ch] = 0.1f * (float)ch;
    } else
#endif
    if (live) {
        const uint32_t word = list[j];
        if (SLOT) slot = entry_slot(word, g, gx, gy);
        const uint32_t gid = word & kGidMask;
        e.k = j;                       // n_contrib counts cell-list positions
        const float2 xy = g.xy[gid];
        const float4 co = g.conic_opacity[gid];
        e.dep = g.depth[gid];
        const float *c = colors + (size_t)C * gid;
        if (C <= 3) { e.c[0] = c[0]; e.c[1] = c[1]; e.c[2] = c[2]; }
        else {
            const float2 c01 = *reinterpret_cast<const float2 *>(c), c23 = *reinterpret_cast<const float2 *>(c + 2),
                         c45 = *reinterpret_cast<const float2 *>(c + 4);
            e.c[0] = c01.x; e.c[1] = c01.y; e.c[2] = c23.x; e.c[C > 3 ? 3 : 0] = c23.y; e.c[C > 3 ? 4 : 0] = c45.x; e.c[C > 3 ? 5 : 0] = c45.y;
        }
        x = xy.x; y = xy.y; cA = co.x; cB = co.y; cC = co.z; e.o = co.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        e.dx[i] = x - (cx0 + (float)i);           // cx0 + i is an exact small integer: == xy.x - (float)px
        e.dy[i] = y - (cy0 + (float)i);
        e.Adx2[i] = (cA * e.dx[i]) * e.dx[i];
        e.Cdy2[i] = (cC * e.dy[i]) * e.dy[i];
        e.Bdx[i] = cB * e.dx[i];
    }
}

// table line of one pixel (written by the lane that "owns" it: lane p of the row); outside the image: nothing contributes
template <int C, int LEAN = 0>
__device__ __forceinline__ void load_pixel(float *line, const ViewCtx &c, const int pxi, const int pyi)
{
    constexpr int CG = LEAN == 3 ? 3 : C;           // channels that carry a gradient
    const ViewParams &vp = c.vp;
    const bool inside = pxi < vp.W && pyi < vp.H;
    const size_t P = (size_t)vp.H * vp.W, pid = (size_t)pyi * vp.W + pxi;
    float T_final = 0.f, gD = 0.f, gA = 0.f, g[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint32_t last = 0;
    if (inside) {
        T_final = c.im.final_T[pid];
        last = c.im.n_contrib[pid];
#pragma unroll
        for (int ch = 0; ch < CG; ++ch) g[ch] = c.dL_dcolor[ch * P + pid];
        if (LEAN < 2 && c.dL_ddepth) gD = c.dL_ddepth[pid];
        if (c.dL_dalpha) gA = c.dL_dalpha[pid];
    }
    float bgdot = 0.f;
#pragma unroll
    for (int ch = 0; ch < CG; ++ch) bgdot += vp.bg[ch] * g[ch];
    float4 *ln = reinterpret_cast<float4 *>(line);
    if constexpr (CG > 3) {
        ln[0] = make_float4(g[0], g[1], g[2], g[3]);
        ln[1] = make_float4(g[4], g[5], gD, gA);
        ln[2] = make_float4(T_final, T_final * bgdot, __uint_as_float(last), 0.f);
    } else {
        ln[0] = make_float4(g[0], g[1], g[2], gD);
        ln[1] = make_float4(gA, T_final, T_final * bgdot, __uint_as_float(last));
    }
}

// The 64 records of a chunk (one per lane) go out as whole records: staged in LDS, then lanes 4i .. 4i + 2 (.. 4i + 3 for
// the 64-byte records) write the consecutive 16-byte parts of record i, so that a record reaches the memory system as
// ONE contiguous request (two when a 48-byte record crosses a line) instead of three or four scattered 16-byte pieces from
// one lane -- measured on the bench scene, the blend backward cost ~60 us per (16-byte piece per record): 457 -> 391 us
// with two pieces instead of three, 290 us with the same bytes written in list order, 265 us without the stores.
template <int RSP>
struct RecStage { __attribute__((aligned(16))) float rec[64][RSP]; uint32_t slot[64]; };
template <int RSP>
__device__ __forceinline__ void store_records(float *__restrict__ rec, const uint32_t rec_cap, const uint32_t slot, const float (&acc)[13],
                                              RecStage<RSP> &st, const int lane)
{
    __builtin_amdgcn_wave_barrier();
    float4 *mine = reinterpret_cast<float4 *>(st.rec[lane]);
    mine[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    mine[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    if (RSP > 8) mine[2] = make_float4(acc[8], acc[9], acc[10], acc[11]);
    if (RSP > 12) mine[3] = make_float4(acc[12], 0.f, 0.f, 0.f);
    st.slot[lane] = slot;
    __builtin_amdgcn_wave_barrier();
    if constexpr (RSP == 8) {      // 32-byte records: two lanes per record, every lane of both store instructions active
#if !defined(DM4D_BWD_STORE4)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int entry = 32 * p + (lane >> 1), part = lane & 1;
            const uint32_t sl = st.slot[entry];
#if defined(DM4D_BWD_PROBE) && (DM4D_BWD_PROBE & 32)
            if (sl == 0xFFFFFFF0u)
#else
            if (sl < rec_cap)
#endif
            {
#if defined(DM4D_BWD_NT)
                typedef float f4v __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(*reinterpret_cast<const f4v *>(&st.rec[entry][4 * part]), reinterpret_cast<f4v *>(rec + (size_t)sl * RSP) + part);
#else
                reinterpret_cast<float4 *>(rec + (size_t)sl * RSP)[part] = reinterpret_cast<const float4 *>(st.rec[entry])[part];
#endif
            }
        }
        return;
#endif
    }
    const int part = lane & 3;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int entry = 16 * p + (lane >> 2);
        const uint32_t sl = st.slot[entry];
#if defined(DM4D_BWD_PROBE) && (DM4D_BWD_PROBE & 32)     // TIMING PROBE: no record stores
        if (part < RSP / 4 && sl == 0xFFFFFFF0u)
#else
        if (part < RSP / 4 && sl < rec_cap)
#endif
            reinterpret_cast<float4 *>(rec + (size_t)sl * RSP)[part] = reinterpret_cast<const float4 *>(st.rec[entry])[part];
    }
}

// LDS of one wave of the backward kernel: the pixel tables of its rows
template <int C, int RSP> struct BwdSmemV2 { __attribute__((aligned(16))) float tab[4][16 * PixTab<C>::kLine]; RecStage<RSP> stage; };      // (LEAN == 3 uses 8 of a line's 12 floats)

// regular blocks: wave = quadrant, row = cell
template <int C, int LEAN>
__device__ __forceinline__ void render_bwd_cells(const BatchDesc &d, const uint32_t bid, BwdSmemV2<C, (LEAN >= 2 ? 8 : (C <= 3 || LEAN) ? 12 : 16)> &sm)
{
    constexpr int RSP = (LEAN >= 2 ? 8 : (C <= 3 || LEAN) ? 12 : 16);   // == grad_stride(C, LEAN)
    const WaveTrace trace;
    int view, tile, q;
    if (!block_to_quadrant(d, bid, view, tile, q)) { trace.done(0); return; }
    const ViewCtx c = resolve(d, view);
    const ViewParams &vp = c.vp;
    const GeomPtrs &g = c.g;
    const BinPtrs &b = c.b;
    float *__restrict__ rec = c.dLq;
    const uint32_t rec_cap = c.rec_cap;
    const int lane = threadIdx.x;
    const int tx = tile % vp.gx, ty = tile / vp.gx;
    const LanePixel lp = lane_pixel(lane, tx, ty, q);
    const int row = lp.row, li = lp.li;

    const uint32_t s = g.tile_start[tile];
    uint32_t nr = (s < c.cap) ? g.ccount[tile * kCells + lp.cell] : 0u;
    uint32_t nd = (s < c.cap) ? min(g.cdone[tile * kCells + lp.cell], nr) : 0u;   // entries this row's forward consumed
    if (nr >= kWideBwd) nr = nd = 0u;                                               // a wide block's
    const uint32_t ndmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(nd));
    const uint32_t *__restrict__ list = b.clist + (size_t)lp.cell * b.cap + s;
    const int gx = lp.px >> 2, gy = lp.py >> 2;      // the row's cell, in cells from the image origin (as cellinfo counts)
    // entries the forward never reached get all-zero records, so that B2 can sum every Gaussian's contiguous record
    // block without looking anything up
#if defined(DM4D_BWD_PROBE) && (DM4D_BWD_PROBE & 256)    // TIMING PROBE: no zero records for the entries the forward never reached
    for (uint32_t j = nr; j < nr; j += 16u) {
#else
    for (uint32_t j = nd + (uint32_t)li; j < nr; j += 16u) {
#endif
        const uint32_t slot = entry_slot(list[j], g, gx, gy);
        if (slot < rec_cap) {
            float4 *dst = reinterpret_cast<float4 *>(rec + (size_t)slot * RSP);
#pragma unroll
            for (int i = 0; i < RSP / 4; ++i) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (ndmax == 0 || ndmax < g_min_work) { trace.done(0); return; }
    set_priority_by_length(ndmax);
    float *tab = sm.tab[row];
    load_pixel<C, LEAN>(tab + li * PixTab<C, LEAN>::kLine, c, lp.px, lp.py);
    __builtin_amdgcn_wave_barrier();
    const float cx0 = (float)(lp.px - (li & 3)), cy0 = (float)(lp.py - (li >> 2));
    const bool front_lane = li == 15;
    for (uint32_t c0 = ((ndmax - 1u) / 16u) * 16u;; c0 -= 16u) {
        const uint32_t j = c0 + 15u - (uint32_t)li;       // reversed: lane 0 holds the entry farthest back
        EntryRegs<C> e;
        uint32_t slot;
        load_entry<C>(e, slot, j < nd, list, j, g, c.colors, cx0, cy0, gx, gy);
        float acc[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) acc[i] = 0.f;
        cell_pixels<C, LEAN, false>(e, acc, front_lane, tab);
        store_records<RSP>(rec, rec_cap, j < nd ? slot : 0xFFFFFFFFu, acc, sm.stage, lane);
        if (c0 == 0u) break;
    }
    trace.done(ndmax);
}

// wide blocks: wave = one long cell
template <int C, int LEAN>
__device__ __forceinline__ void render_bwd_wide_cells(const BatchDesc &d, const uint32_t bid, const uint32_t nblocks, BwdSmemV2<C, (LEAN >= 2 ? 8 : (C <= 3 || LEAN) ? 12 : 16)> &sm)
{
    constexpr int RSP = (LEAN >= 2 ? 8 : (C <= 3 || LEAN) ? 12 : 16);
    const WaveTrace trace;
    uint32_t traced = 0;
    const int view = (int)(bid % (uint32_t)d.B);
    const uint32_t first = bid / (uint32_t)d.B, step = nblocks / (uint32_t)d.B;
    const ViewCtx c = resolve(d, view);
    const ViewParams &vp = c.vp;
    const GeomPtrs &g = c.g;
    const BinPtrs &b = c.b;
    float *__restrict__ rec = c.dLq;
    const uint32_t rec_cap = c.rec_cap;
    const int lane = threadIdx.x, p = lane & 15;
    const uint32_t n_long = min(g.counters[kCntLong], (uint32_t)(c.T * kCells));
    if (first < n_long) __builtin_amdgcn_s_setprio(3);   // the launch's critical path: issue before the regular blocks' waves
    float *tab = sm.tab[0];
    const bool front_lane = lane == 63;
    for (uint32_t it = first; it < n_long; it += step) {
        const uint32_t cellid = g.longlist[it];
        const int tile = (int)(cellid / kCells), cell = (int)(cellid % kCells);
        const int tx = tile % vp.gx, ty = tile / vp.gx, q = cell >> 2, rw = cell & 3;
        const int cxi = tx * kTile + (q & 1) * 8 + (rw & 1) * 4, cyi = ty * kTile + (q >> 1) * 8 + (rw >> 1) * 4;
        const uint32_t s = g.tile_start[tile];
        const uint32_t nr = g.ccount[cellid], nd = min(g.cdone[cellid], nr);
        const uint32_t *__restrict__ list = b.clist + (size_t)cell * b.cap + s;
        const int gx = cxi >> 2, gy = cyi >> 2;
#if defined(DM4D_BWD_PROBE) && (DM4D_BWD_PROBE & 256)
        for (uint32_t j = nr; j < nr; j += 64u) {
#else
        for (uint32_t j = nd + (uint32_t)lane; j < nr; j += 64u) {     // zero records for the unconsumed entries
#endif
            const uint32_t slot = entry_slot(list[j], g, gx, gy);
            if (slot < rec_cap) {
                float4 *dst = reinterpret_cast<float4 *>(rec + (size_t)slot * RSP);
#pragma unroll
                for (int i = 0; i < RSP / 4; ++i) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (nd == 0u) continue;
        traced += nd;
        __builtin_amdgcn_wave_barrier();
        if (lane < 16) load_pixel<C, LEAN>(tab + p * PixTab<C, LEAN>::kLine, c, cxi + (p & 3), cyi + (p >> 2));
        __builtin_amdgcn_wave_barrier();
        const float cx0 = (float)cxi, cy0 = (float)cyi;
        for (uint32_t c0 = ((nd - 1u) / 64u) * 64u;; c0 -= 64u) {
            const uint32_t j = c0 + 63u - (uint32_t)lane;
            EntryRegs<C> e;
            uint32_t slot;
            load_entry<C>(e, slot, j < nd, list, j, g, c.colors, cx0, cy0, gx, gy);
            float acc[13];
#pragma unroll
            for (int i = 0; i < 13; ++i) acc[i] = 0.f;
            cell_pixels<C, LEAN, true>(e, acc, front_lane, tab);
            store_records<RSP>(rec, rec_cap, j < nd ? slot : 0xFFFFFFFFu, acc, sm.stage, lane);
            if (c0 == 0u) break;
        }
        __builtin_amdgcn_wave_barrier();
    }
    trace.done(traced);
}


// ---------------------------------------------------------------------------------------- B1, tile-record mode
// ONE WORKGROUP PER TILE (4 waves).  The entry-parallel arithmetic above is unchanged -- a lane still ends a chunk holding
// the complete record of its (Gaussian, cell) entry -- but instead of going to HBM as its own record it is ADDED, with
// ds_add_f32, to the accumulator of the entry's (Gaussian, TILE) pair in the workgroup's LDS: slot = the entry's position in
// the tile list (cpos, written by K4 beside the cell lists).  When the sixteen cells are done the workgroup writes one record
// per tile-list entry: 3.4x fewer records than (Gaussian, cell) pairs on the bench scene (0.18 GB per 8-view step instead of
// 0.61 GB), and B2 reads that much less.  The order in which the cells' contributions meet in an accumulator depends on how
// the waves are scheduled, so the last bits of the gradients vary from run to run (upstream's float atomicAdd has the same
// property); the deterministic kernels above stay the default of the C ABI and of the parity tests.
// (Global float atomics instead of LDS ones are not an option: measured with tools/ubench/atomics.hip, the L2 retires
// ~125 G atomic dwords per second whatever their locality -- 0.9 ms for the 115 M of a step.)
//   LDS holds kTileWindow accumulators: longer tile lists are processed in WINDOWS of tile-list positions, back to front;
//   a row only takes the entries of its list whose position lies in the current window (the lists are ordered by
//   position, so these are the next entries from the back), the per-pixel carries simply stay in the row's pixel table
//   between windows.
//   Cells with >= kWideBwd entries are walked by a whole wave (64 entries per step, as in the wide blocks above), dealt
//   round-robin to the four waves before they turn to their quadrants.
#ifndef DM4D_TILE_WAVES
#define DM4D_TILE_WAVES 5
#endif
template <int C, int LEAN, int WN>
__global__ __launch_bounds__(256, DM4D_TILE_WAVES) void k_render_bwd_tile(BatchDesc d)
{
    constexpr int LN = PixTab<C>::kLine;
    constexpr int NV = LEAN ? 9 : (C > 3 ? 13 : 10);      // values of a record
    constexpr int RSP = (LEAN >= 2 ? 8 : (C <= 3 || LEAN) ? 12 : 16);       // floats of a record in memory (== grad_stride)
    __shared__ float s_acc[WN * NV];
    __shared__ __attribute__((aligned(16))) float s_tab[kCells][16 * LN];
    __shared__ uint32_t s_pos[kCells], s_cnt[kCells];
    __shared__ uint32_t s_slot[WN];     // record slot of every tile-list entry of the window (resolved while the tables load)
    __shared__ uint32_t s_done;
    const int T = ((d.W + kTile - 1) / kTile) * ((d.H + kTile - 1) / kTile);
    const uint32_t rank = blockIdx.x;
    const int view = (int)(rank % (uint32_t)d.B);
    const ViewCtx c = resolve(d, view);
    const ViewParams &vp = c.vp;
    const GeomPtrs &g = c.g;
    const BinPtrs &b = c.b;
    const int tile = (int)g.order[rank / (uint32_t)d.B];
    (void)T;
    const uint32_t s = g.tile_start[tile];
    uint32_t n = g.tile_count[tile];
    if (s >= c.cap) n = 0;
    else if (s + n > c.cap) n = c.cap - s;
    if (n == 0) return;
#ifdef DM4D_TILE_SKIP_ABOVE
    if (n > DM4D_TILE_SKIP_ABOVE) return;      // timing experiment only (wrong gradients)
#endif
    float *__restrict__ rec = c.dLq;
    const uint32_t rec_cap = c.rec_cap;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tx = tile % vp.gx, ty = tile / vp.gx;
    if (n >= 2048u) __builtin_amdgcn_s_setprio(3);
    else if (n >= 1024u) __builtin_amdgcn_s_setprio(2);
    else if (n >= 512u) __builtin_amdgcn_s_setprio(1);
    {   // pixel tables of the sixteen cells: thread = pixel
        const int cell = tid >> 4, p = tid & 15, q = cell >> 2, rw = cell & 3;
        load_pixel<C>(s_tab[cell] + p * LN, c, tx * kTile + (q & 1) * 8 + (rw & 1) * 4 + (p & 3), ty * kTile + (q >> 1) * 8 + (rw >> 1) * 4 + (p >> 2));
        if (tid == 0) s_done = 0u;
        if (tid < kCells) {
            const uint32_t nr = g.ccount[tile * kCells + tid];
            s_cnt[tid] = nr;
            s_pos[tid] = min(g.cdone[tile * kCells + tid], nr);      // entries the forward consumed: [0, pos) are left to do
        }
    }
    for (uint32_t hi = n; hi > 0u;) {
        const uint32_t lo = hi > (uint32_t)WN ? hi - (uint32_t)WN : 0u;
        for (uint32_t i = tid; i < (hi - lo) * (uint32_t)NV; i += 256u) s_acc[i] = 0.f;
        // Gaussian-major record slot of the window's tile-list entries = first record of the Gaussian + the rank of this tile
        // among the tiles its cell block spans (two dependent gathers: resolved here, off the critical path of the flush)
        for (uint32_t i = tid; i < hi - lo; i += 256u) {
            const uint32_t gid = b.point_list[s + lo + i];
            const uint4 ci = g.cellinfo[gid];
            const TileSpan ts = tile_span(ci.x, ci.y);
            const int ox = tx - ts.tx0, oy = ty - ts.ty0;
            const uint32_t slot = ci.z + (uint32_t)(oy * ts.tnx + ox);
            s_slot[i] = (ox >= 0 && ox < ts.tnx && oy >= 0 && oy < ts.tny && slot < rec_cap) ? slot : 0xFFFFFFFFu;
        }
        __syncthreads();
        // ---- the wide cells of the tile, dealt to the waves ----
        uint32_t widx = 0;
        for (int cell = 0; cell < kCells; ++cell) {
            if (s_cnt[cell] < kWideBwd) continue;
            if ((widx++ & 3u) != (uint32_t)wave) continue;
            const int q = cell >> 2, rw = cell & 3;
            const int cxi = tx * kTile + (q & 1) * 8 + (rw & 1) * 4, cyi = ty * kTile + (q >> 1) * 8 + (rw >> 1) * 4;
            const uint32_t *__restrict__ list = b.clist + (size_t)cell * b.cap + s;
            const uint16_t *__restrict__ kp = b.cpos + (size_t)cell * b.cap + s;
            float *tab = s_tab[cell];
            uint32_t p = s_pos[cell];
            for (;;) {
                const bool inb = (uint32_t)lane < p;
                const uint32_t j = p - 1u - (uint32_t)lane;
                const uint32_t kk = inb ? (uint32_t)kp[j] : 0u;
                const bool live = inb && kk >= lo;
                const uint64_t bl = __builtin_amdgcn_ballot_w64(live);
                if (bl == 0ull) break;
                EntryRegs<C> e;
                uint32_t unused;
                load_entry<C, false>(e, unused, live, list, j, g, c.colors, (float)cxi, (float)cyi, cxi >> 2, cyi >> 2);
                float acc[13];
#pragma unroll
                for (int i = 0; i < 13; ++i) acc[i] = 0.f;
                cell_pixels<C, LEAN, true>(e, acc, lane == 63, tab);
                if (live) {
                    float *a = s_acc + (kk - lo) * (uint32_t)NV;
#pragma unroll
                    for (int i = 0; i < NV; ++i) lds_fadd(a + i, acc[i]);
                }
                __builtin_amdgcn_wave_barrier();
                p -= (uint32_t)__builtin_popcountll(bl);
            }
            if (lane == 0) s_pos[cell] = p;
        }
        // ---- this wave's quadrant: row = cell ----
        {
            const int row = lane >> 4, li = lane & 15, cell = 4 * wave + row;
            const int cxi = tx * kTile + (wave & 1) * 8 + (row & 1) * 4, cyi = ty * kTile + (wave >> 1) * 8 + (row >> 1) * 4;
            const uint32_t *__restrict__ list = b.clist + (size_t)cell * b.cap + s;
            const uint16_t *__restrict__ kp = b.cpos + (size_t)cell * b.cap + s;
            float *tab = s_tab[cell];
            const bool is_wide = s_cnt[cell] >= kWideBwd;
            uint32_t p = is_wide ? 0u : s_pos[cell];
            for (;;) {
                const bool inb = (uint32_t)li < p;
                const uint32_t j = p - 1u - (uint32_t)li;
                const uint32_t kk = inb ? (uint32_t)kp[j] : 0u;
                const bool live = inb && kk >= lo;
                const uint64_t bl = __builtin_amdgcn_ballot_w64(live);
                if (bl == 0ull) break;
                EntryRegs<C> e;
                uint32_t unused;
                load_entry<C, false>(e, unused, live, list, j, g, c.colors, (float)cxi, (float)cyi, cxi >> 2, cyi >> 2);
                float acc[13];
#pragma unroll
                for (int i = 0; i < 13; ++i) acc[i] = 0.f;
                // (a wide cell's table belongs to the wave that walks it: this row is inert and must not write its carries back)
                cell_pixels<C, LEAN, false>(e, acc, li == 15 && !is_wide, tab);
                if (live) {
                    float *a = s_acc + (kk - lo) * (uint32_t)NV;
#pragma unroll
                    for (int i = 0; i < NV; ++i) lds_fadd(a + i, acc[i]);
                }
                __builtin_amdgcn_wave_barrier();
                p -= (uint32_t)__builtin_popcount((uint32_t)(bl >> (16 * row)) & 0xFFFFu);
            }
            if (li == 0 && !is_wide) s_pos[cell] = p;
        }
        // A tile of ONE window needs no barrier here: a wave that is done leaves (its wave slot and registers are free for
        // the next workgroup's waves at once), the last one out writes the records.
        const bool single = n <= (uint32_t)WN;
        uint32_t fw = (uint32_t)wave, fstep = 256u;
        if (single) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            uint32_t prev = 0;
            if (lane == 0) prev = __hip_atomic_fetch_add(&s_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            prev = (uint32_t)__builtin_amdgcn_readfirstlane((int)prev);
            if (prev != 3u) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            fw = 0u; fstep = 64u;
        } else __syncthreads();
        // ---- one record per tile-list entry of the window; lanes 4i .. 4i+2 write the 16-byte parts of record i ----
        for (uint32_t base = fw * 64u; base < hi - lo; base += fstep) {
            const int part = lane & 3;
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const uint32_t entry = base + 16u * pp + ((uint32_t)lane >> 2);
                const uint32_t sl = entry < hi - lo ? s_slot[entry] : 0xFFFFFFFFu;
                if (part < RSP / 4 && sl != 0xFFFFFFFFu) {
                    const float *a = s_acc + entry * (uint32_t)NV;
                    float4 v;
                    v.x = (4 * part + 0 < NV) ? a[4 * part + 0] : 0.f;
                    v.y = (4 * part + 1 < NV) ? a[4 * part + 1] : 0.f;
                    v.z = (4 * part + 2 < NV) ? a[4 * part + 2] : 0.f;
                    v.w = (4 * part + 3 < NV) ? a[4 * part + 3] : 0.f;
                    reinterpret_cast<float4 *>(rec + (size_t)sl * RSP)[part] = v;
                }
            }
        }
        hi = lo;
        if (hi > 0u) __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------- debug read-out of n_contrib
// The kernels count a pixel's contributors in its CELL list (positions in a subsequence of the tile list); upstream's
// n_contrib is the position of the last contributor in the TILE list + 1.  dm4d_raster_read_image_state translates: the
// pixel's last cell-list entry names a Gaussian, which occurs once in the tile's sorted list.
__global__ __launch_bounds__(256) void k_n_contrib_tile_positions(GeomPtrs g, BinPtrs b, ImgPtrs im, int H, int W, uint32_t cap,
                                                                   uint32_t *__restrict__ out)
{
    const int pid = blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= H * W) return;
    const int px = pid % W, py = pid / W, gx = (W + kTile - 1) / kTile;
    const int tile = (py / kTile) * gx + px / kTile;
    const int ix = px % kTile, iy = py % kTile;
    const int cell = 4 * ((ix >> 3) + 2 * (iy >> 3)) + ((ix >> 2) & 1) + 2 * ((iy >> 2) & 1);
    const uint32_t lastj = im.n_contrib[pid];
    uint32_t res = 0u;
    const uint32_t s = g.tile_start[tile], e = min(g.tile_start[tile + 1], cap);
    if (lastj > 0u && s + lastj - 1u < cap) {
        const uint32_t gid = b.clist[(size_t)cell * b.cap + s + lastj - 1u] & kGidMask;
        for (uint32_t k = s; k < e; ++k)
            if (b.point_list[k] == gid) { res = k - s + 1u; break; }
    }
    out[pid] = res;
}
int launch_n_contrib_tile_positions(void *geom, void *binning, void *image, int N, int H, int W, int64_t cap, uint32_t *out, hipStream_t st)
{
    const GeomLayout L = geom_layout(N, H, W);
    const GeomPtrs g = geom_ptrs(geom, L);
    const BinPtrs b = bin_ptrs(binning, cap);
    const ImgPtrs im = img_ptrs(image, H, W);
    const int P = H * W;
    if (P > 0) hipLaunchKernelGGL(k_n_contrib_tile_positions, dim3((P + 255) / 256), dim3(256), 0, st, g, b, im, H, W, (uint32_t)b.cap, out);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

// ---------------------------------------------------------------------------------------- launchers
// One launch for both kinds of block: the first `wide_blocks` workgroups take the wide cells (they are dispatched
// first and raise their issue priority: the launch's critical path), the rest the regular quadrants.
template <int C, int LEAN>
__global__ __launch_bounds__(64) void k_render_bwd(BatchDesc d, uint32_t wide_blocks)
{
    __shared__ BwdSmemV2<C, (LEAN >= 2 ? 8 : (C <= 3 || LEAN) ? 12 : 16)> sm;
    if (blockIdx.x < wide_blocks) render_bwd_wide_cells<C, LEAN>(d, blockIdx.x, wide_blocks, sm);
    else render_bwd_cells<C, LEAN>(d, blockIdx.x - wide_blocks, sm);
}

constexpr int kLongWaves = 128;    // waves per view of the forward's long-cell kernel (each loops over the long cells it owns)
constexpr int kWideWaves = 256;    // wide blocks per view of the backward (each loops over the wide cells it owns)
int launch_render_fwd(const BatchDesc &d, hipStream_t st)
{
    const int T = ((d.W + kTile - 1) / kTile) * ((d.H + kTile - 1) / kTile);
    if (T <= 0) return DM4D_OK;
    const int blocks = (int)((((int64_t)T * d.B + 7) / 8) * 8 * 4);
    ProfScope prof_(kKRenderFwd, st);
    AuxStream *a = aux_stream();
    if (!a) {                       // no helper streams: the long cells of the large tiles right here
        int rc = launch_render_fwd_long(d, st);
        if (rc) return rc;
    }
    if (d.C <= 3) hipLaunchKernelGGL(k_render_fwd<3>, dim3(blocks), dim3(64), 0, st, d);
    else hipLaunchKernelGGL(k_render_fwd<6>, dim3(blocks), dim3(64), 0, st, d);
    DM4D_HIP_CHECK(hipGetLastError());
    if (a && a->pending2) {         // k_render_fwd_long, started by launch_tile_sort right after the large variant
        DM4D_HIP_CHECK(hipStreamWaitEvent(st, a->join2, 0));
        a->pending2 = false;
    }
    return DM4D_OK;
}

int launch_render_fwd_long(const BatchDesc &d, hipStream_t st)
{
    const int T = ((d.W + kTile - 1) / kTile) * ((d.H + kTile - 1) / kTile);
    if (T <= 0) return DM4D_OK;
    static const int long_waves = getenv("DM4D_LONG_WAVES") ? atoi(getenv("DM4D_LONG_WAVES")) : kLongWaves;      // (A/B switch)
    const int long_blocks = (min(T * kCells, long_waves) * d.B + 3) / 4 * 4 / 4;   // 4 waves (cells) per workgroup, a multiple of B waves in all
    if (d.C <= 3) hipLaunchKernelGGL(k_render_fwd_long<3>, dim3(long_blocks), dim3(256), 0, st, d);
    else hipLaunchKernelGGL(k_render_fwd_long<6>, dim3(long_blocks), dim3(256), 0, st, d);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int launch_render_bwd(const BatchDesc &d, hipStream_t st)
{
    const int T = ((d.W + kTile - 1) / kTile) * ((d.H + kTile - 1) / kTile);
    if (T <= 0) return DM4D_OK;
    const int blocks = (int)((((int64_t)T * d.B + 7) / 8) * 8 * 4);
    ProfScope prof_(kKRenderBwd, st);
    if (d.lean && d.C != 6) { set_error("lean backward records need 6 channels"); return DM4D_ERR_INVALID; }
    if (d.tile_records) {      // one workgroup per tile, (Gaussian, tile) records summed in LDS
        const dim3 tgrid((unsigned)T * (unsigned)d.B);
        if (d.lean >= 2) { set_error("tile records: no 32-byte variant (lean must be 0 or 1)"); return DM4D_ERR_INVALID; }
        if (d.C <= 3) hipLaunchKernelGGL((k_render_bwd_tile<3, 0, kTileWindow>), tgrid, dim3(256), 0, st, d);
        else if (d.lean) hipLaunchKernelGGL((k_render_bwd_tile<6, 1, kTileWindow>), tgrid, dim3(256), 0, st, d);
        else hipLaunchKernelGGL((k_render_bwd_tile<6, 0, kTileWindow>), tgrid, dim3(256), 0, st, d);
        DM4D_HIP_CHECK(hipGetLastError());
        return DM4D_OK;
    }
    // the long cells' blocks first (multiple of 8 of them: the regular blocks keep their XCD), then the quadrants
    static const int wide_waves = getenv("DM4D_WIDE_WAVES") ? atoi(getenv("DM4D_WIDE_WAVES")) : kWideWaves;      // (A/B switch)
    const uint32_t long_blocks = (uint32_t)(min(T * kCells, wide_waves) * d.B);
    const dim3 grid(long_blocks + (uint32_t)blocks);
    if (d.C <= 3) hipLaunchKernelGGL((k_render_bwd<3, 0>), grid, dim3(64), 0, st, d, long_blocks);
    else if (d.lean == 3) hipLaunchKernelGGL((k_render_bwd<6, 3>), grid, dim3(64), 0, st, d, long_blocks);
    else if (d.lean == 2) hipLaunchKernelGGL((k_render_bwd<6, 2>), grid, dim3(64), 0, st, d, long_blocks);
    else if (d.lean) hipLaunchKernelGGL((k_render_bwd<6, 1>), grid, dim3(64), 0, st, d, long_blocks);
    else hipLaunchKernelGGL((k_render_bwd<6, 0>), grid, dim3(64), 0, st, d, long_blocks);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // namespace dm4d
