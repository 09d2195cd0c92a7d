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
#include "common.h"
#include "raster.h"

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

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int kChunk = 16;   // list entries a row stages per step (one per lane of the row)
constexpr int kFwdPairs = 2, kBwdPairs = 1;   // PAIRS of entries per inner-loop step

// LDS layout of a staged chunk: one 32-float block per PAIR of consecutive list entries (j even, j + 1),
// geometry interleaved across the two entries so that a ds_read_b128 delivers register pairs the packed
// FP32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes' worth of work per issue slot)
// consume directly, colours kept per entry as channel pairs for the packed accumulators:
//   [0..3]   x_j x_j1 y_j y_j1        [4..7]  A_j A_j1 B_j B_j1      [8..11] C_j C_j1 o_j o_j1   (conic A,B,C; opacity)
//   [12..19] entry j  : c0 c1 c2 c3 | c4 c5 depth 1.0                 [20..27] entry j + 1, same
//   [28..29] tile-list position k of j, j + 1 (bits)                  [30..35] padding
// Strides are chosen against LDS bank conflicts (measured: 64 % of the forward's LDS cycles were conflicts
// with 32-float pairs and 1 KB rows): 36 floats per pair spreads the 8 pairs a row stages at once over all
// banks, 292 floats per row puts the 4 rows' broadcast reads (4 distinct addresses per instruction) on
// disjoint bank groups.
constexpr int kPairFloats = 36;
constexpr int kRowFloats = (kChunk / 2) * kPairFloats + 4;
constexpr int kPair4 = kPairFloats / 4;   // float4 per pair block

// gather one list entry: r0 = (x, y, conic.x, conic.y)  r1 = (conic.z, opacity, depth, k bits)
//                        r2 = colours 0..3               r3 = colours 4..5
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
        r[3] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        const float2 c01 = *reinterpret_cast<const float2 *>(c);
        const float2 c23 = *reinterpret_cast<const float2 *>(c + 2);
        const float2 c45 = *reinterpret_cast<const float2 *>(c + 4);
        r[2] = make_float4(c01.x, c01.y, c23.x, c23.y);
        r[3] = make_float4(c45.x, c45.y, 0.f, 0.f);
    }
}
// An all-zero entry is inert: opacity 0 gives alpha 0 < 1/255, so rows whose list is shorter than the
// wave's longest one blend padding entries with weight exactly 0.
__device__ __forceinline__ void zero_entry(float4 (&r)[4])
{
#pragma unroll
    for (int v = 0; v < 4; ++v) r[v] = make_float4(0.f, 0.f, 0.f, 0.f);
}
// lane li of a row stores its gathered entry into the row's chunk (pair li >> 1, half li & 1)
__device__ __forceinline__ void stage_entry(float *row_base, int li, const float4 (&r)[4])
{
    float *pb = row_base + (li >> 1) * kPairFloats;
    const int h = li & 1;
    pb[0 + h] = r[0].x; pb[2 + h] = r[0].y; pb[4 + h] = r[0].z; pb[6 + h] = r[0].w;
    pb[8 + h] = r[1].x; pb[10 + h] = r[1].y; pb[28 + h] = r[1].w;
    *reinterpret_cast<float4 *>(pb + 12 + 8 * h) = r[2];
    *reinterpret_cast<float4 *>(pb + 16 + 8 * h) = make_float4(r[3].x, r[3].y, r[1].z, 1.0f);
}

// N pairs of alphas, written step-by-step across the pairs so that the instruction stream interleaves the
// independent dependency chains (a lone wave on a long silhouette list issues dependent VALU ops slowly).
// Per element bit-identical to:  power = -0.5f * ((A dx) dx + (C dy) dy) - (B dx) dy;  G = det_expf(power).
template <int N>
__device__ __forceinline__ void pair_gauss(const f4v (&g0)[N], const f4v (&g1)[N], const f4v (&g2)[N], f2v pxf, f2v pyf,
                                           f2v (&dx)[N], f2v (&dy)[N], f2v (&pw)[N], f2v (&G)[N])
{
    f2v u[N], v[N], w[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { dx[j] = g0[j].xy - pxf; dy[j] = g0[j].zw - pyf; }
#pragma unroll
    for (int j = 0; j < N; ++j) { u[j] = g1[j].xy * dx[j]; v[j] = g2[j].xy * dy[j]; w[j] = g1[j].zw * dx[j]; }
#pragma unroll
    for (int j = 0; j < N; ++j) { u[j] = u[j] * dx[j]; v[j] = v[j] * dy[j]; w[j] = w[j] * dy[j]; }
#pragma unroll
    for (int j = 0; j < N; ++j) u[j] = u[j] + v[j];
#pragma unroll
    for (int j = 0; j < N; ++j) pw[j] = (f2v)(-0.5f) * u[j] - w[j];
    // det_expf (common.h), two elements per instruction where the ISA has a packed form
    const float L2E_HI = 0x1.715476p+0f, L2E_LO = 0x1.4ae0c0p-26f;
    f2v x[N], n[N], f[N], p[N];
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] = f2v{fmaxf(pw[j].x, -87.0f), fmaxf(pw[j].y, -87.0f)};
#pragma unroll
    for (int j = 0; j < N; ++j) n[j] = __builtin_elementwise_rint(x[j] * (f2v)(L2E_HI));
#pragma unroll
    for (int j = 0; j < N; ++j) f[j] = __builtin_elementwise_fma(x[j], (f2v)(L2E_HI), -n[j]);
#pragma unroll
    for (int j = 0; j < N; ++j) f[j] = __builtin_elementwise_fma(x[j], (f2v)(L2E_LO), f[j]);
#pragma unroll
    for (int j = 0; j < N; ++j) p[j] = __builtin_elementwise_fma((f2v)(0x1.446c7ep-13f), f[j], (f2v)(0x1.5f48c8p-10f));
#pragma unroll
    for (int j = 0; j < N; ++j) p[j] = __builtin_elementwise_fma(p[j], f[j], (f2v)(0x1.3b29d8p-7f));
#pragma unroll
    for (int j = 0; j < N; ++j) p[j] = __builtin_elementwise_fma(p[j], f[j], (f2v)(0x1.c6aeccp-5f));
#pragma unroll
    for (int j = 0; j < N; ++j) p[j] = __builtin_elementwise_fma(p[j], f[j], (f2v)(0x1.ebfbe0p-3f));
#pragma unroll
    for (int j = 0; j < N; ++j) p[j] = __builtin_elementwise_fma(p[j], f[j], (f2v)(0x1.62e430p-1f));
#pragma unroll
    for (int j = 0; j < N; ++j) p[j] = __builtin_elementwise_fma(p[j], f[j], (f2v)(1.0f));
#pragma unroll
    for (int j = 0; j < N; ++j) G[j] = f2v{__builtin_ldexpf(p[j].x, (int)n[j].x), __builtin_ldexpf(p[j].y, (int)n[j].y)};
}

// block -> (view, tile, quadrant): rank in the launch order of K3 (the r-th longest tile of every view, views
// interleaved); the four quadrants of a tile are blocks b, b+8, b+16, b+24 (same XCD).  False past the end.
__device__ __forceinline__ bool block_to_quadrant(const BatchDesc &d, int b, int &view, int &tile, int &q)
{
    const int xcd = b & 7, r = b >> 3;
    q = r & 3;
    const uint32_t rank = (uint32_t)((r >> 2) * 8 + xcd);
    const int T = ((d.W + kTile - 1) / kTile) * ((d.H + kTile - 1) / kTile);
    if (rank >= (uint32_t)d.B * (uint32_t)T) return false;
    const GeomLayout L = geom_layout(d.N, d.H, d.W);
    view = (int)(rank % (uint32_t)d.B);
    tile = (int)reinterpret_cast<const uint32_t *>(d.geom + (size_t)view * d.geom_stride + L.order)[rank / (uint32_t)d.B];
    return true;
}
// Waves with long lists raise their issue priority: while the bulk of the (short) waves keeps the SIMD
// saturated a wave only gets a fair share of the issue slots, so the long waves -- started first by K4b --
// would still finish last.  With priority they run at lone-wave speed from the start.
__device__ __forceinline__ void set_priority_by_length(uint32_t n)
{
    if (n >= 384u) __builtin_amdgcn_s_setprio(3);
    else if (n >= 192u) __builtin_amdgcn_s_setprio(2);
    else if (n >= 128u) __builtin_amdgcn_s_setprio(1);
}
__device__ __forceinline__ uint32_t row_max_u32(uint32_t v) { return row_allmax_u32(v); }

// lane -> pixel: row (lane >> 4) = cell (row & 1, row >> 1) of the quadrant, lane & 15 = pixel of the cell
struct LanePixel { int px, py, row, li, cell; };
__device__ __forceinline__ LanePixel lane_pixel(int lane, int tx, int ty, int q)
{
    LanePixel L;
    L.row = lane >> 4;
    L.li = lane & 15;
    L.px = tx * kTile + (q & 1) * 8 + (L.row & 1) * 4 + (L.li & 3);
    L.py = ty * kTile + (q >> 1) * 8 + (L.row >> 1) * 4 + (L.li >> 2);
    L.cell = 4 * q + L.row;
    return L;
}

// ---------------------------------------------------------------------------------------- K5
template <int C>
__global__ __launch_bounds__(64) void k_render_fwd(BatchDesc d)
{
    // one wave-private staging buffer: the next chunk waits in registers (prefetched during the
    // blend loop) and is written after the loop -- same wave, program order, no hazard
    __shared__ __attribute__((aligned(16))) float s_p[4 * kRowFloats];
    const WaveTrace trace;
    int view, tile, q;
    if (!block_to_quadrant(d, blockIdx.x, view, tile, q)) { trace.done(0); return; }
    const ViewCtx c = resolve(d, view);
    const ViewParams &vp = c.vp;
    const float *__restrict__ colors = c.colors;
    const GeomPtrs &g = c.g;
    const BinPtrs &b = c.b;
    const uint32_t cap = c.cap;
    const ImgPtrs &im = c.im;
    float *__restrict__ out_color = c.out_color, *__restrict__ out_depth = c.out_depth,
                       *__restrict__ out_alpha = c.out_alpha;
    const int lane = threadIdx.x;
    const int tx = tile % vp.gx, ty = tile / vp.gx;
    const LanePixel lp = lane_pixel(lane, tx, ty, q);
    const int px = lp.px, py = lp.py, row = lp.row, li = lp.li;
    const bool inside = px < vp.W && py < vp.H;
    const f2v pxf = (f2v)((float)px), pyf = (f2v)((float)py);

    const uint32_t s = g.tile_start[tile];
    uint32_t nr = (s < cap) ? g.ccount[tile * kCells + lp.cell] : 0u;   // this row's list length
    const bool row_long = (s < cap) && g.cflag[tile * kCells + lp.cell] != 0u;   // blended by k_render_fwd_long
    if (row_long) nr = 0u;
    const uint32_t nmax = wave_max_u32(nr);
    set_priority_by_length(nmax);
    if (nmax < g_min_work) { trace.done(0); return; }
    const uint2 *__restrict__ list = b.clist + (size_t)lp.cell * b.cap + s;
    float *row_base = s_p + row * kRowFloats;

    float T_ = 1.0f;
    f2v C01 = (f2v)(0.f), C23 = (f2v)(0.f), C45 = (f2v)(0.f), DW = (f2v)(0.f);   // colours | (depth, alpha) sums
    uint32_t last = 0, lastj = 0;
    bool done = !inside | row_long;

    float4 r[4];
    zero_entry(r);
    if ((uint32_t)li < nr) gather_entry<C>(list[li], g, colors, r);
    for (uint32_t c0 = 0; c0 < nmax; c0 += kChunk) {
        const int cnt = (c0 < nr) ? (int)min((uint32_t)kChunk, nr - c0) : 0;
        __builtin_amdgcn_wave_barrier();
        stage_entry(row_base, li, r);
        zero_entry(r);
        if (c0 + kChunk + (uint32_t)li < nr) gather_entry<C>(list[c0 + kChunk + li], g, colors, r);   // prefetch
        __builtin_amdgcn_wave_barrier();
        if (__ballot((!done) & (cnt > 0)) == 0) break;   // every pixel with entries left is saturated
        int t = 0;
        do {
            // 2 * kFwdPairs entries per step.  Their alphas are independent and evaluated two per packed
            // instruction; the blend below is sequential and branch-free (selects, not exec-mask branches):
            // lanes that do not take an entry blend with weight 0, which leaves their accumulators
            // bit-identical; padding entries are inert.
            const f4v *P = reinterpret_cast<const f4v *>(row_base + (t >> 1) * kPairFloats);
            f4v g0[kFwdPairs], g1[kFwdPairs], g2[kFwdPairs];
#pragma unroll
            for (int j = 0; j < kFwdPairs; ++j) { g0[j] = P[kPair4 * j + 0]; g1[j] = P[kPair4 * j + 1]; g2[j] = P[kPair4 * j + 2]; }
            f2v dx[kFwdPairs], dy[kFwdPairs], pw[kFwdPairs], G[kFwdPairs], al[kFwdPairs];
            pair_gauss<kFwdPairs>(g0, g1, g2, pxf, pyf, dx, dy, pw, G);
#pragma unroll
            for (int j = 0; j < kFwdPairs; ++j) {
                const f2v oa = g2[j].zw * G[j];
                al[j] = f2v{fminf(0.99f, oa.x), fminf(0.99f, oa.y)};
            }
#pragma unroll
            for (int j = 0; j < kFwdPairs; ++j) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f4v e0 = P[kPair4 * j + 3 + 2 * h], e1 = P[kPair4 * j + 4 + 2 * h];
                    const float alpha = h ? al[j].y : al[j].x, power = h ? pw[j].y : pw[j].x;
                    const uint32_t kbits = __float_as_uint(h ? P[kPair4 * j + 7].y : P[kPair4 * j + 7].x);
                    const float test_T = T_ * (1.0f - alpha);
                    const bool valid = (!done) & (power <= 0.0f) & (alpha >= 1.0f / 255.0f);
                    const bool stop = valid & (test_T < 0.0001f);
                    const bool contrib = valid & (!stop);
                    const float w = contrib ? alpha * T_ : 0.f;
                    const f2v ww = (f2v)(w);
                    C01 = __builtin_elementwise_fma(e0.xy, ww, C01);
                    C23 = __builtin_elementwise_fma(e0.zw, ww, C23);
                    if (C > 3) C45 = __builtin_elementwise_fma(e1.xy, ww, C45);
                    DW = __builtin_elementwise_fma(e1.zw, ww, DW);      // depth * w | 1 * w
                    T_ = contrib ? test_T : T_;
                    last = contrib ? kbits + 1u : last;
                    lastj = contrib ? c0 + (uint32_t)(t + 2 * j + h) + 1u : lastj;
                    done = done | stop;
                }
            }
            t += 2 * kFwdPairs;
        } while (t < kChunk && __ballot((!done) & (t < cnt)) != 0);
    }
    if (inside && !row_long) {
        const size_t P = (size_t)vp.H * vp.W;
        const size_t pid = (size_t)py * vp.W + px;
        im.final_T[pid] = T_;
        im.n_contrib[pid] = last;
        const float Cacc[6] = {C01.x, C01.y, C23.x, C23.y, C45.x, C45.y};
#pragma unroll
        for (int ch = 0; ch < C; ++ch) out_color[ch * P + pid] = __builtin_fmaf(T_, vp.bg[ch], Cacc[ch]);
        out_depth[pid] = DW.x;
        out_alpha[pid] = DW.y;
    }
    const uint32_t wj = row_max_u32(lastj), wk = row_max_u32(last);
    if (li == 0 && !row_long) {
        g.cdone[tile * kCells + lp.cell] = wj;
        g.ckmax[tile * kCells + lp.cell] = wk;
    }
    trace.done(nmax);
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
// Staging: 64 entries per chunk, one per lane: [0..5] x y A B C opacity | [6] k bits | [8..13] colours | [14] depth | [15] 1
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
        const uint2 *__restrict__ list = b.clist + (size_t)cell * b.cap + s;

        float T_ = 1.0f;
        f2v C01 = (f2v)(0.f), C23 = (f2v)(0.f), C45 = (f2v)(0.f), DW = (f2v)(0.f);
        uint32_t last = 0, lastj = 0;
        bool done = !inside | (r != 0);      // the pixel state lives in row 0

        float4 e[4];
        zero_entry(e);
        if ((uint32_t)lane < nr) gather_entry<C>(list[lane], g, colors, e);
        for (uint32_t c0 = 0; c0 < nr; c0 += 64u) {
            const int cnt = (int)min(64u, nr - c0);
            __builtin_amdgcn_wave_barrier();
            {
                float *se = s_e + lane * 16;
                *reinterpret_cast<float4 *>(se) = e[0];                                              // x y A B
                *reinterpret_cast<float4 *>(se + 4) = make_float4(e[1].x, e[1].y, e[1].w, 0.f);     // C opacity k
                *reinterpret_cast<float4 *>(se + 8) = e[2];                                          // colours 0..3
                *reinterpret_cast<float4 *>(se + 12) = make_float4(e[3].x, e[3].y, e[1].z, 1.0f);   // colours 4 5, depth, 1
            }
            zero_entry(e);
            if (c0 + 64u + (uint32_t)lane < nr) gather_entry<C>(list[c0 + 64u + lane], g, colors, e);   // prefetch
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
                    const uint32_t kbits = __float_as_uint(se[6]);
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
                    last = contrib ? kbits + 1u : last;
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
            im.n_contrib[pid] = last;
            const float Cacc[6] = {C01.x, C01.y, C23.x, C23.y, C45.x, C45.y};
#pragma unroll
            for (int ch = 0; ch < C; ++ch) c.out_color[ch * P + pid] = __builtin_fmaf(T_, vp.bg[ch], Cacc[ch]);
            c.out_depth[pid] = DW.x;
            c.out_alpha[pid] = DW.y;
        }
        const uint32_t wj = row_max_u32(lastj), wk = row_max_u32(last);
        if (lane == 0) {
            g.cdone[cellid] = wj;
            g.ckmax[cellid] = wk;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------- B1
// r[i] (lanes 0..7 of every row) = v[i] + v[i] of lane l + 8; lanes 8..15 keep r[i].  One instruction per value
// (v_add_f32 with a bank-masked row rotation), which the compiler cannot form from update_dpp + add.  s_nop: a DPP read
// of a VGPR needs 2 wait states after the VALU write of it (inline asm is opaque to the hazard recogniser).
#define DM4D_DPPADD(O, I) "v_add_f32_dpp %" O ", %" I ", %" I " row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
template <int RS>
__device__ __forceinline__ void pair_sum_low_half(float (&r)[RS], const float (&v)[13])
{
    static_assert(RS == 9 || RS == 10 || RS == 13, "record sizes of grad_stride()");
    if constexpr (RS == 9) {
        asm volatile("s_nop 1\n\t" DM4D_DPPADD("0", "9") DM4D_DPPADD("1", "10") DM4D_DPPADD("2", "11") DM4D_DPPADD("3", "12") DM4D_DPPADD("4", "13") DM4D_DPPADD("5", "14") DM4D_DPPADD("6", "15") DM4D_DPPADD("7", "16") DM4D_DPPADD("8", "17")
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8])
                     : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]));
    } else if constexpr (RS == 10) {
        asm volatile("s_nop 1\n\t" DM4D_DPPADD("0", "10") DM4D_DPPADD("1", "11") DM4D_DPPADD("2", "12") DM4D_DPPADD("3", "13") DM4D_DPPADD("4", "14") DM4D_DPPADD("5", "15") DM4D_DPPADD("6", "16") DM4D_DPPADD("7", "17") DM4D_DPPADD("8", "18") DM4D_DPPADD("9", "19")
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9])
                     : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]));
    } else {
        asm volatile("s_nop 1\n\t" DM4D_DPPADD("0", "13") DM4D_DPPADD("1", "14") DM4D_DPPADD("2", "15") DM4D_DPPADD("3", "16") DM4D_DPPADD("4", "17") DM4D_DPPADD("5", "18") DM4D_DPPADD("6", "19") DM4D_DPPADD("7", "20") DM4D_DPPADD("8", "21") DM4D_DPPADD("9", "22") DM4D_DPPADD("10", "23") DM4D_DPPADD("11", "24") DM4D_DPPADD("12", "25")
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12])
                     : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]));
    }
}
#undef DM4D_DPPADD
constexpr int kRedStride = 68;   // floats per value row of the transposed reduction buffer (64 lanes + pad)
constexpr int kRedHalf = 36;     // the same with 8 lanes per row (32 + pad)

// LDS of one wave of the backward kernel (floats): the regular blocks' staging rows + record slots + reduction
// buffer, or the long-cell blocks' chunk + slots + reduction buffer -- the two kinds of block share one launch.
template <int C, bool LEAN>
struct BwdSmem {
    static constexpr int RS = LEAN ? 9 : 7 + C;
    static constexpr int U = 2 * kBwdPairs;
    static constexpr int regular = 4 * kRowFloats + 4 * kChunk + U * RS * kRedHalf;
    static constexpr int longc = 64 * 16 + 64 + RS * kRedStride;
    static constexpr int floats = regular > longc ? regular : longc;
};

template <int C, bool LEAN>
__device__ __forceinline__ void render_bwd_cells(const BatchDesc &d, const uint32_t bid, float *smem)
{
    constexpr int RS = LEAN ? 9 : 7 + C;              // values per record
    constexpr int RSP = (C <= 3 || LEAN) ? 12 : 16;   // == grad_stride(C, LEAN): floats per (padded) record
    constexpr int U = 2 * kBwdPairs;          // entries per inner-loop step
    float *s_p = smem;                                                                       // [4 * kRowFloats]
    uint32_t (*s_slot)[kChunk] = reinterpret_cast<uint32_t (*)[kChunk]>(smem + 4 * kRowFloats);   // [4][kChunk]
    // reduction buffer: lanes l and l + 8 of a row are added with one DPP row rotation first, so only 8 lanes per
    // row go through LDS (half the reduction's LDS bytes, and 2.4 KB less LDS per wave: 16 -> 20 waves per CU)
    float (*s_red)[RS][kRedHalf] = reinterpret_cast<float (*)[RS][kRedHalf]>(smem + 4 * kRowFloats + 4 * kChunk);   // [U]
    const WaveTrace trace;
    int view, tile, q;
    if (!block_to_quadrant(d, bid, view, tile, q)) { trace.done(0); return; }
    const ViewCtx c = resolve(d, view);
    const ViewParams &vp = c.vp;
    const float *__restrict__ colors = c.colors;
    const GeomPtrs &g = c.g;
    const BinPtrs &b = c.b;
    const uint32_t cap = c.cap;
    const ImgPtrs &im = c.im;
    const float *__restrict__ dL_dcolor = c.dL_dcolor, *__restrict__ dL_ddepth = c.dL_ddepth,
                             *__restrict__ dL_dalpha = c.dL_dalpha;
    float *__restrict__ rec = c.dLq;
    const uint32_t rec_cap = c.rec_cap;
    const int lane = threadIdx.x;
    const int tx = tile % vp.gx, ty = tile / vp.gx;
    const LanePixel lp = lane_pixel(lane, tx, ty, q);
    const int px = lp.px, py = lp.py, row = lp.row, li = lp.li;
    const bool inside = px < vp.W && py < vp.H;
    const f2v pxf = (f2v)((float)px), pyf = (f2v)((float)py);

    const uint32_t s = g.tile_start[tile];
    uint32_t nr = (s < cap) ? g.ccount[tile * kCells + lp.cell] : 0u;
    uint32_t nd = (s < cap) ? g.cdone[tile * kCells + lp.cell] : 0u;   // entries this row's forward consumed
    if (nr >= kLongCell) nr = nd = 0u;                                  // long cell: k_render_bwd_long's
    const uint32_t ndmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(nd));   // uniform: scalar loop control
    const uint2 *__restrict__ list = b.clist + (size_t)lp.cell * b.cap + s;
    const uint32_t *__restrict__ slots = b.cslot + (size_t)lp.cell * b.cap + s;
    {
        // entries the forward never reached get all-zero records, so that B2 can sum every Gaussian's
        // contiguous record block without looking anything up
        for (uint32_t j = nd + (uint32_t)li; j < nr; j += 16u) {
            const uint32_t slot = slots[j];
            if (slot < rec_cap) {
                float4 *dst = reinterpret_cast<float4 *>(rec + (size_t)slot * RSP);
#pragma unroll
                for (int i = 0; i < RSP / 4; ++i) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    if (ndmax == 0 || ndmax < g_min_work) { trace.done(0); return; }
    set_priority_by_length(ndmax);
    float *row_base = s_p + row * kRowFloats;

    const size_t P = (size_t)vp.H * vp.W;
    const size_t pid = (size_t)py * vp.W + px;
    float T_final = 0.f, gD = 0.f, gA = 0.f;
    float gCol[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
    float T_ = T_final, S = Tb;   // S: sum V w of the entries behind + T_final bg.g
    const f2v g01 = f2v{gCol[0], gCol[1]}, g23 = f2v{gCol[2], gCol[3]}, g45 = f2v{gCol[4], gCol[5]}, gDA = f2v{gD, gA};
    // reduction role of this lane: value li of its row (lanes with li >= RS idle in the sum)
    const int red_i = li < RS ? li : 0;
    const float4 *red_src = reinterpret_cast<const float4 *>(&s_red[0][red_i][row * 8]);
    constexpr int kRedBuf4 = RS * kRedHalf / 4;   // float4 per reduction buffer

    const uint32_t c_last = ((ndmax - 1) / kChunk) * kChunk;
    float4 r[4];
    uint32_t rslot = 0;
    zero_entry(r);
    if (c_last + (uint32_t)li < nd) {
        gather_entry<C>(list[c_last + li], g, colors, r);
        rslot = slots[c_last + li];
    }
    for (uint32_t c0 = c_last;; c0 -= kChunk) {
        const int cnt = (c0 < nd) ? (int)min((uint32_t)kChunk, nd - c0) : 0;
        const int tmax = (int)min((uint32_t)kChunk, ndmax - c0);
        __builtin_amdgcn_wave_barrier();
        stage_entry(row_base, li, r);
        s_slot[row][li] = rslot;
        zero_entry(r);
        if (c0 >= (uint32_t)kChunk && c0 - kChunk + (uint32_t)li < nd) {   // prefetch the chunk in front
            gather_entry<C>(list[c0 - kChunk + li], g, colors, r);
            rslot = slots[c0 - kChunk + li];
        }
        __builtin_amdgcn_wave_barrier();
        // U entries per step (aligned groups, highest first; slots past the row's count hold inert padding):
        // alphas two per packed instruction, one LDS round trip of the reduction for the whole group.
        for (int tg = ((tmax - 1) / U) * U; tg >= 0; tg -= U) {
            const f4v *P4 = reinterpret_cast<const f4v *>(row_base + (tg >> 1) * kPairFloats);
            f4v g0[kBwdPairs], g1[kBwdPairs], g2[kBwdPairs];
#pragma unroll
            for (int j = 0; j < kBwdPairs; ++j) { g0[j] = P4[kPair4 * j + 0]; g1[j] = P4[kPair4 * j + 1]; g2[j] = P4[kPair4 * j + 2]; }
            const uint2 slot2 = *reinterpret_cast<const uint2 *>(&s_slot[row][tg]);   // record slots of the pair (tg is even)
            f2v dx2[kBwdPairs], dy2[kBwdPairs], pw[kBwdPairs], Gr2[kBwdPairs];
            pair_gauss<kBwdPairs>(g0, g1, g2, pxf, pyf, dx2, dy2, pw, Gr2);
            __builtin_amdgcn_wave_barrier();
            float rsum[RS];
#pragma unroll
            for (int e = U - 1; e >= 0; --e) {          // back to front inside the group
                const int j = e >> 1, h = e & 1, t = tg + e;
                const f4v e0 = P4[kPair4 * j + 3 + 2 * h], e1 = P4[kPair4 * j + 4 + 2 * h];
                const uint32_t k = __float_as_uint(h ? P4[kPair4 * j + 7].y : P4[kPair4 * j + 7].x);
                const float dx = h ? dx2[j].y : dx2[j].x, dy = h ? dy2[j].y : dy2[j].x;
                const float power = h ? pw[j].y : pw[j].x, Gr = h ? Gr2[j].y : Gr2[j].x;
                const float op = h ? g2[j].w : g2[j].z;
                // Branch-free: lanes that do not take the entry contribute exact zeros.
                // (alpha and G are 0 there, so w and q are; dL/dalpha itself needs no mask: it is finite and only
                // ever multiplied by G.)  S carries sum V w + T_final bg.g.
                const float alpha_r = fminf(0.99f, op * Gr);
                const bool contrib = (k < last) & (power <= 0.0f) & (alpha_r >= 1.0f / 255.0f);
                const float alpha = contrib ? alpha_r : 0.f;
                const float G = contrib ? Gr : 0.f;
                const float inv_om = __builtin_amdgcn_rcpf(1.f - alpha);
                const float Tn = T_ * inv_om;
                T_ = contrib ? Tn : T_;
                const float w = alpha * Tn;
                // V = dL/d(blended value of this entry) = depth gD + gA + sum_ch colour_ch gCol_ch
                f2v va = e1.zw * gDA;
                va = __builtin_elementwise_fma(e0.xy, g01, va);
                va = __builtin_elementwise_fma(e0.zw, g23, va);
                if (C > 3) va = __builtin_elementwise_fma(e1.xy, g45, va);
                const float V = va.x + va.y;
                const float dL_da = Tn * V - S * inv_om;
                S = __builtin_fmaf(V, w, S);
                // dL/dmean2D and dL/dconic are linear in the five moments sum_pixels q (dx, dy, dx^2, dx dy, dy^2),
                // q = dL/dG G: the records carry the moments, B2 applies the (per-Gaussian) map once (k_gather_bwd)
                const float q = (op * dL_da) * G;
                const f2v dxy = f2v{dx, dy};
                const f2v v01 = dxy * (f2v)(q);                             // q dx, q dy
                const f2v v24 = v01 * dxy;                                  // q dx^2, q dy^2
                const float v3 = v01.x * dy;                                // q dx dy
                const f2v ww = (f2v)(w);
                const f2v c01 = ww * g01, c23 = ww * g23, c45 = ww * g45;
                float v[13];
                v[0] = v01.x; v[1] = v01.y; v[2] = v24.x; v[3] = v3; v[4] = v24.y;
                if (LEAN) {   // static appearance frozen: no dL/dopacity, no dL/dcolour for channels 0..2
                    v[5] = w * gD;
                    v[6] = c23.y; v[7] = c45.x; v[8] = c45.y;
                } else {
                    v[5] = G * dL_da;
                    v[6] = w * gD;
                    v[7] = c01.x; v[8] = c01.y; v[9] = c23.x; v[10] = c23.y; v[11] = c45.x; v[12] = c45.y;
                }
                // transposed reduction: [value][lane] in LDS, lane i of the row sums value i over the row.
                // v[i] of lane l + v[i] of lane l ^ 8 first (row_ror:8): the odd entry of the pair in all lanes, then
                // the even entry over it in lanes 0..7 of every row (bank_mask 0x3), so that ONE unmasked LDS write
                // per value stores both entries (lanes 0..7: entry e = 0, lanes 8..15: entry e = 1).
                static_assert(U == 2, "the pair reduction below");
                if (e & 1) {
#pragma unroll
                    for (int i = 0; i < RS; ++i)
                        rsum[i] = v[i] + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[i]), 0x128, 0xf, 0xf, true));
                } else {
                    pair_sum_low_half<RS>(rsum, v);
#pragma unroll
                    for (int i = 0; i < RS; ++i) s_red[li >> 3][i][row * 8 + (li & 7)] = rsum[i];
                }
                (void)t;
            }
            __builtin_amdgcn_wave_barrier();
            {
                // all LDS reads of the pair first (one wait), then the two predicated record stores
                float total[U];
#pragma unroll
                for (int e = 0; e < U; ++e) {
                    const float4 *src = red_src + e * kRedBuf4;
                    const float4 a0 = src[0], a1 = src[1];
                    // fixed summation tree (deterministic): pairs of packed adds
                    f2v p0 = f2v{a0.x, a0.y} + f2v{a0.z, a0.w};
                    f2v p1 = f2v{a1.x, a1.y} + f2v{a1.z, a1.w};
                    p0 = p0 + p1;
                    total[e] = p0.x + p0.y;
                }
#pragma unroll
                for (int e = U - 1; e >= 0; --e) {
                    const uint32_t slot = e ? slot2.y : slot2.x;
                    if (li < RS && tg + e < cnt && slot < rec_cap) rec[(size_t)slot * RSP + li] = total[e];   // the padding floats stay unwritten (never read as values)
                }
            }
        }
        if (c0 == 0) break;
    }
    trace.done(ndmax);
}

// ---------------------------------------------------------------------------------------- B1 (long cells)
// One wave per long cell (raster.h, kLongCell).  Lane = (entry slot r = lane >> 4, pixel p = lane & 15 of the cell):
// the four rows take four CONSECUTIVE list entries of the one cell for the same 16 pixels.  Per step: every row
// evaluates alpha and V of its entry; row 0 runs the sequential (T, S) chain over the four entries back to front
// and hands (w, dL/dalpha) back; every row then forms the gradient values of its entry and reduces them over its
// 16 lanes exactly as k_render_bwd does.  Same operations on the same values in the same order: the records are
// bit-identical, but a 1200-entry silhouette cell no longer costs the launch 1200 serial reduction round trips.
// Staging (64 entries per chunk, one per lane): [0..5] x y A B C opacity | [6] k bits | [8..13] colours | [14] depth | [15] 1
template <int C, bool LEAN>
__device__ __forceinline__ void render_bwd_long_cells(const BatchDesc &d, const uint32_t bid, const uint32_t nblocks, float *smem)
{
    constexpr int RS = LEAN ? 9 : 7 + C;
    constexpr int RSP = (C <= 3 || LEAN) ? 12 : 16;
    float *s_e = smem;                                                              // [64 * 16]
    uint32_t *s_slot = reinterpret_cast<uint32_t *>(smem + 64 * 16);               // [64]
    float (*s_red)[kRedStride] = reinterpret_cast<float (*)[kRedStride]>(smem + 64 * 16 + 64);   // [RS]
    const int view = (int)(bid % (uint32_t)d.B);
    const uint32_t first = bid / (uint32_t)d.B, step = nblocks / (uint32_t)d.B;
    const ViewCtx c = resolve(d, view);
    const ViewParams &vp = c.vp;
    const float *__restrict__ colors = c.colors;
    const GeomPtrs &g = c.g;
    const BinPtrs &b = c.b;
    const ImgPtrs &im = c.im;
    float *__restrict__ rec = c.dLq;
    const uint32_t rec_cap = c.rec_cap;
    const int lane = threadIdx.x, r = lane >> 4, p = lane & 15;
    const uint32_t n_long = min(g.counters[kCntLong], (uint32_t)(c.T * kCells));
    if (first < n_long) __builtin_amdgcn_s_setprio(3);   // the launch's critical path: issue before the regular kernel's waves
    const int red_i = p < RS ? p : 0;
    const float4 *red_src = reinterpret_cast<const float4 *>(&s_red[red_i][r * 16]);
    for (uint32_t it = first; it < n_long; it += step) {
        const uint32_t cellid = g.longlist[it];
        const int tile = (int)(cellid / kCells), cell = (int)(cellid % kCells);
        const int tx = tile % vp.gx, ty = tile / vp.gx, q = cell >> 2, rw = cell & 3;
        const int px = tx * kTile + (q & 1) * 8 + (rw & 1) * 4 + (p & 3);
        const int py = ty * kTile + (q >> 1) * 8 + (rw >> 1) * 4 + (p >> 2);
        const bool inside = px < vp.W && py < vp.H;
        const float pxf = (float)px, pyf = (float)py;
        const uint32_t s = g.tile_start[tile];
        const uint32_t nr = g.ccount[cellid], nd = min(g.cdone[cellid], nr);
        const uint2 *__restrict__ list = b.clist + (size_t)cell * b.cap + s;
        const uint32_t *__restrict__ slots = b.cslot + (size_t)cell * b.cap + s;
        for (uint32_t j = nd + (uint32_t)lane; j < nr; j += 64u) {     // zero records for the unconsumed entries
            const uint32_t slot = slots[j];
            if (slot < rec_cap) {
                float4 *dst = reinterpret_cast<float4 *>(rec + (size_t)slot * RSP);
#pragma unroll
                for (int i = 0; i < RSP / 4; ++i) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (nd == 0u) continue;
        // every row holds the pixel's upstream gradients (V and the contribution test need them)
        const size_t P = (size_t)vp.H * vp.W;
        const size_t pid = (size_t)py * vp.W + px;
        float T_final = 0.f, gD = 0.f, gA = 0.f;
        float gCol[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        uint32_t last = 0;
        if (inside) {
            T_final = im.final_T[pid];
            last = im.n_contrib[pid];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) gCol[ch] = c.dL_dcolor[ch * P + pid];
            if (c.dL_ddepth) gD = c.dL_ddepth[pid];
            if (c.dL_dalpha) gA = c.dL_dalpha[pid];
        }
        float bgdot = 0.f;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) bgdot += vp.bg[ch] * gCol[ch];
        const float Tb = T_final * bgdot;
        float T_ = T_final, S = Tb;   // S: sum V w of the entries behind + T_final bg.g
        const f2v g01 = f2v{gCol[0], gCol[1]}, g23 = f2v{gCol[2], gCol[3]}, g45 = f2v{gCol[4], gCol[5]}, gDA = f2v{gD, gA};

        const uint32_t c_last = ((nd - 1u) / 64u) * 64u;
        float4 e[4];
        uint32_t eslot = 0;
        zero_entry(e);
        if (c_last + (uint32_t)lane < nd) {
            gather_entry<C>(list[c_last + lane], g, colors, e);
            eslot = slots[c_last + lane];
        }
        for (uint32_t c0 = c_last;; c0 -= 64u) {
            const int cnt = (int)min(64u, nd - c0);
            __builtin_amdgcn_wave_barrier();
            {
                float *se = s_e + lane * 16;
                *reinterpret_cast<float4 *>(se) = e[0];                                              // x y A B
                *reinterpret_cast<float4 *>(se + 4) = make_float4(e[1].x, e[1].y, e[1].w, 0.f);     // C opacity k
                *reinterpret_cast<float4 *>(se + 8) = e[2];                                          // colours 0..3
                *reinterpret_cast<float4 *>(se + 12) = make_float4(e[3].x, e[3].y, e[1].z, 1.0f);   // colours 4 5, depth, 1
                s_slot[lane] = eslot;
            }
            zero_entry(e);
            if (c0 >= 64u) {                                               // prefetch the chunk in front (all < nd)
                gather_entry<C>(list[c0 - 64u + lane], g, colors, e);
                eslot = slots[c0 - 64u + lane];
            }
            __builtin_amdgcn_wave_barrier();
            for (int tg = ((cnt - 1) / 4) * 4; tg >= 0; tg -= 4) {
                // ---- every row: its entry t = tg + r (slots past the count hold inert padding: opacity 0) ----
                const float *se = s_e + (tg + r) * 16;
                const float4 ga = *reinterpret_cast<const float4 *>(se), gb = *reinterpret_cast<const float4 *>(se + 4);
                const f4v e0 = *reinterpret_cast<const f4v *>(se + 8), e1 = *reinterpret_cast<const f4v *>(se + 12);
                const float cA = ga.z, cB = ga.w, cC = gb.x, op = gb.y;
                const uint32_t k = __float_as_uint(gb.z);
                const float dx = ga.x - pxf, dy = ga.y - pyf;
                const float power = -0.5f * ((cA * dx) * dx + (cC * dy) * dy) - (cB * dx) * dy;
                const float Gr = det_expf(power);
                const float alpha_r = fminf(0.99f, op * Gr);
                const bool contrib_r = (k < last) & (power <= 0.0f) & (alpha_r >= 1.0f / 255.0f);
                f2v va = e1.zw * gDA;
                va = __builtin_elementwise_fma(e0.xy, g01, va);
                va = __builtin_elementwise_fma(e0.zw, g23, va);
                if (C > 3) va = __builtin_elementwise_fma(e1.xy, g45, va);
                // all four rows get the (alpha or -1, V) of all four entries: three gfx950 lane swaps per value (VALU, no
                // LDS round trip -- the long waves wait behind the regular kernel's LDS traffic otherwise)
                float al[4], Vv[4];
                rows_allgather(contrib_r ? alpha_r : -1.0f, al);
                rows_allgather(va.x + va.y, Vv);
                // ---- the sequential (T, S) chain over the four entries, back to front: every row runs it on the same
                // values (bit-identical T_ and S in all four), and keeps the (w, dL/dalpha) of its own entry ----
                float w = 0.f, dL_da = 0.f;
                {
#pragma unroll
                    for (int h = 3; h >= 0; --h) {
                        const bool contrib = al[h] >= 0.0f;
                        const float alpha = contrib ? al[h] : 0.0f;
                        const float V = Vv[h];
                        const float inv_om = __builtin_amdgcn_rcpf(1.f - alpha);
                        const float Tn = T_ * inv_om;
                        T_ = contrib ? Tn : T_;
                        const float w_h = alpha * Tn;
                        const float d_h = Tn * V - S * inv_om;   // unmasked: only ever multiplied by the masked G
                        S = __builtin_fmaf(V, w_h, S);
                        w = (r == h) ? w_h : w;
                        dL_da = (r == h) ? d_h : dL_da;
                    }
                }
                // ---- every row: the gradient values of its entry, reduced over its 16 lanes ----
                const float G = contrib_r ? Gr : 0.f;
                const float q = (op * dL_da) * G;                           // moments, as in k_render_bwd
                const f2v dxy = f2v{dx, dy};
                const f2v v01 = dxy * (f2v)(q);
                const f2v v24 = v01 * dxy;
                const float v3 = v01.x * dy;
                const f2v ww = (f2v)(w);
                const f2v c01 = ww * g01, c23 = ww * g23, c45 = ww * g45;
                float v[13];
                v[0] = v01.x; v[1] = v01.y; v[2] = v24.x; v[3] = v3; v[4] = v24.y;
                if (LEAN) {
                    v[5] = w * gD;
                    v[6] = c23.y; v[7] = c45.x; v[8] = c45.y;
                } else {
                    v[5] = G * dL_da;
                    v[6] = w * gD;
                    v[7] = c01.x; v[8] = c01.y; v[9] = c23.x; v[10] = c23.y; v[11] = c45.x; v[12] = c45.y;
                }
#pragma unroll
                for (int i = 0; i < RS; ++i) s_red[i][lane] = v[i];
                __builtin_amdgcn_wave_barrier();
                {
                    const float4 a0 = red_src[0], a1 = red_src[1], a2 = red_src[2], a3 = red_src[3];
                    f2v p0 = f2v{a0.x, a0.y} + f2v{a0.z, a0.w};
                    f2v p1 = f2v{a1.x, a1.y} + f2v{a1.z, a1.w};
                    f2v p2 = f2v{a2.x, a2.y} + f2v{a2.z, a2.w};
                    f2v p3 = f2v{a3.x, a3.y} + f2v{a3.z, a3.w};
                    p0 = p0 + p1;
                    p2 = p2 + p3;
                    p0 = p0 + p2;
                    const float total = p0.x + p0.y;
                    const int t = tg + r;
                    const uint32_t slot = s_slot[t < 64 ? t : 63];
                    if (p < RS && t < cnt && slot < rec_cap) rec[(size_t)slot * RSP + p] = total;
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (c0 == 0u) break;
        }
    }
}

// ---------------------------------------------------------------------------------------- launchers
// One launch for both kinds of block: the first `long_blocks` workgroups take the long cells (they are dispatched
// first and raise their issue priority: the launch's critical path), the rest the regular quadrants.  A separate
// launch on a helper stream cost a fork / join pair of stream events per step (~18 us).
template <int C, bool LEAN>
__global__ __launch_bounds__(64) void k_render_bwd(BatchDesc d, uint32_t long_blocks)
{
    __shared__ __attribute__((aligned(16))) float smem[BwdSmem<C, LEAN>::floats];
    if (blockIdx.x < long_blocks) render_bwd_long_cells<C, LEAN>(d, blockIdx.x, long_blocks, smem);
    else render_bwd_cells<C, LEAN>(d, blockIdx.x - long_blocks, smem);
}

constexpr int kLongWaves = 128;    // waves per view of the long-cell kernel (each loops over the long cells it owns)
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
    const int long_blocks = (min(T * kCells, kLongWaves) * d.B + 3) / 4 * 4 / 4;   // 4 waves (cells) per workgroup, a multiple of B waves in all
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
    // the long cells' blocks first (multiple of 8 of them: the regular blocks keep their XCD), then the quadrants
    const uint32_t long_blocks = (uint32_t)(min(T * kCells, kLongWaves) * d.B);
    const dim3 grid(long_blocks + (uint32_t)blocks);
    if (d.C <= 3) hipLaunchKernelGGL((k_render_bwd<3, false>), grid, dim3(64), 0, st, d, long_blocks);
    else if (d.lean) hipLaunchKernelGGL((k_render_bwd<6, true>), grid, dim3(64), 0, st, d, long_blocks);
    else hipLaunchKernelGGL((k_render_bwd<6, false>), grid, dim3(64), 0, st, d, long_blocks);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // namespace dm4d
