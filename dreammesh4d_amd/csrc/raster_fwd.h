// raster_fwd.h -- the forward blend of ONE WAVE (one 8x8-pixel quadrant of a tile: four 16-lane DPP rows, each walking the list of one
// 4x4-pixel cell), shared by raster_render.hip (k_render_fwd: a wave per workgroup) and raster_bin.hip (round 5: the tile sort's
// workgroup blends its own tile as soon as its cell lists are written).  Device code only; see raster_render.hip for the design notes.
#pragma once

#include "common.h"
#include "raster.h"

namespace dm4d {

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int kChunk = 16;   // list entries a row stages per step (one per lane of the row)
constexpr int kFwdPairs = 2;   // PAIRS of entries per inner-loop step of the forward

// LDS layout of a staged chunk (round 4): one 52-float block per PAIR of consecutive list entries (j even, j + 1).  The lane that
// stages an entry evaluates, ONCE for the row's 4 x 4-pixel cell, what every pixel lane computed for itself before: the offsets of the
// cell's four pixel columns / rows from the splat and the three terms of the quadratic form,
//     Adx2_i = (A dx_i) dx_i,  Bdx_i = B dx_i   (dx_i = x - (cell x0 + i), i = 0..3)      Cdy2_j = (C dy_j) dy_j,  dy_j   (j = 0..3)
// so that a pixel (i, j) of the cell gets its power from TWO 16-byte reads and three packed instructions for a pair of entries,
//     power = -0.5 (Adx2_i + Cdy2_j) - Bdx_i dy_j,
// bit-identical to -0.5f * ((A dx) dx + (C dy) dy) - (B dx) dy evaluated per pixel (the same products in the same association; the
// column / row offsets are the same subtractions of the same integers) -- ten packed instructions per pair before: the forward's time is
// its VALU instruction count (profiles/r03_pmc_sq.md: 98 % VALU-busy), and 16 pixel lanes no longer repeat what one staging lane can do.
//   [4 i .. 4 i + 3]        Adx2_i(e0) Adx2_i(e1) Bdx_i(e0) Bdx_i(e1)          i = 0..3
//   [16 + 4 j .. + 3]       Cdy2_j(e0) Cdy2_j(e1) dy_j(e0)  dy_j(e1)           j = 0..3
//   [32 33]                 opacity(e0) opacity(e1)       [34 35] padding
//   [36..43] entry e0: c0 c1 c2 c3 | c4 c5 depth 1.0      [44..51] entry e1, same
// 16 lanes of a row read 4 distinct 16-byte pieces per instruction (the colours: one); 420 floats per row keep the four rows' reads on
// different banks.
constexpr int kPairFloats = 52;
constexpr int kRowFloats = (kChunk / 2) * kPairFloats + 4;
constexpr int kPair4 = kPairFloats / 4;   // float4 per pair block

// gather one list entry: r0 = (x, y, conic.x, conic.y)  r1 = (conic.z, opacity, depth, -)
//                        r2 = colours 0..3               r3 = colours 4..5
template <int C>
__device__ __forceinline__ void gather_entry(const uint32_t word, const GeomPtrs &g, const float *__restrict__ colors,
                                             float4 (&r)[4])
{
    const uint32_t gid = word & kGidMask;
    const float2 xy = g.xy[gid];
    const float4 co = g.conic_opacity[gid];
    const float dep = g.depth[gid];
    const float *c = colors + (size_t)C * gid;
    r[0] = make_float4(xy.x, xy.y, co.x, co.y);
    r[1] = make_float4(co.z, co.w, dep, 0.f);
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
// lane li of a row stores its gathered entry into the row's chunk (pair li >> 1, half li & 1); (cx0, cy0): the pixel centre of
// the row's cell that is its column 0 / row 0
__device__ __forceinline__ void stage_entry(float *row_base, int li, const float4 (&r)[4], const float cx0, const float cy0)
{
    float *pb = row_base + (li >> 1) * kPairFloats;
    const int h = li & 1;
    const float x = r[0].x, y = r[0].y, A = r[0].z, B = r[0].w, C = r[1].x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float dx = x - (cx0 + (float)k), dy = y - (cy0 + (float)k);      // == xy - (float)pixel: the sums are exact small integers
        pb[4 * k + h] = (A * dx) * dx;
        pb[4 * k + 2 + h] = B * dx;
        pb[16 + 4 * k + h] = (C * dy) * dy;
        pb[16 + 4 * k + 2 + h] = dy;
    }
    pb[32 + h] = r[1].y;
    *reinterpret_cast<float4 *>(pb + 36 + 8 * h) = r[2];
    *reinterpret_cast<float4 *>(pb + 40 + 8 * h) = make_float4(r[3].x, r[3].y, r[1].z, 1.0f);
}

// N pairs of Gaussians G = det_expf(power), written step-by-step across the pairs so that the instruction stream interleaves the
// independent dependency chains (a lone wave on a long silhouette list issues dependent VALU ops slowly).  ga[j] / gc[j]: the pixel's
// column / row pieces of pair j (stage_entry).  Per element bit-identical to
//     power = -0.5f * ((A dx) dx + (C dy) dy) - (B dx) dy;  G = det_expf(power).
template <int N>
__device__ __forceinline__ void pair_gauss(const f4v (&ga)[N], const f4v (&gc)[N], f2v (&pw)[N], f2v (&G)[N])
{
    f2v u[N], w[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { u[j] = ga[j].xy + gc[j].xy; w[j] = ga[j].zw * gc[j].zw; }
#pragma unroll
    for (int j = 0; j < N; ++j) pw[j] = (f2v)(-0.5f) * u[j] - w[j];
#if defined(DM4D_FWD_PROBE) && (DM4D_FWD_PROBE & 1)      // TIMING PROBE (tools/build_variant.sh): the hardware exp2 instead of the contract's polynomial
#pragma unroll
    for (int j = 0; j < N; ++j) G[j] = f2v{__builtin_amdgcn_exp2f(pw[j].x * 0x1.715476p+0f), __builtin_amdgcn_exp2f(pw[j].y * 0x1.715476p+0f)};
    return;
#endif
    // det_expf (common.h), two elements per instruction where the ISA has a packed form
    const float L2E_HI = 0x1.715476p+0f, L2E_LO = 0x1.4ae0c0p-26f, MAGIC = 12582912.0f;
    f2v x[N], t[N], n[N], f[N], p[N];
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] = f2v{fmaxf(pw[j].x, -86.0f), fmaxf(pw[j].y, -86.0f)};
#pragma unroll
    for (int j = 0; j < N; ++j) t[j] = __builtin_elementwise_fma(x[j], (f2v)(L2E_HI), (f2v)(MAGIC));
#pragma unroll
    for (int j = 0; j < N; ++j) n[j] = t[j] - (f2v)(MAGIC);
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
    for (int j = 0; j < N; ++j)
        G[j] = f2v{__uint_as_float(__float_as_uint(p[j].x) + (__float_as_uint(t[j].x) << 23)),
                   __uint_as_float(__float_as_uint(p[j].y) + (__float_as_uint(t[j].y) << 23))};
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

// ---------------------------------------------------------------------------------------- K5: one wave = one quadrant
// (view, tile, q): the quadrant; lane: 0..63 of the wave; s_p: 4 * kRowFloats floats of LDS private to the wave (one staging buffer:
// the next chunk waits in registers -- prefetched during the blend loop -- and is written after the loop: same wave, program order, no
// hazard); Trace: WaveTrace of raster_render.hip or NoTrace; min_work: debug threshold (0 = off).  Called by k_render_fwd (one wave per
// workgroup) and, round 5, by the tile sort's workgroup right after it has written the tile's cell lists (raster_bin.hip).
struct NoTrace { __device__ __forceinline__ void done(uint32_t) const {} };
template <int C, typename Trace>
__device__ __forceinline__ void render_fwd_wave(const BatchDesc &d, const int view, const int tile, const int q, const int lane, float *s_p,
                                                const Trace &trace, const uint32_t min_work)
{
    const ViewCtx c = resolve(d, view);
    const ViewParams &vp = c.vp;
    const float *__restrict__ colors = c.colors;
    const GeomPtrs &g = c.g;
    const BinPtrs &b = c.b;
    const uint32_t cap = c.cap;
    const ImgPtrs &im = c.im;
    float *__restrict__ out_color = c.out_color, *__restrict__ out_depth = c.out_depth,
                       *__restrict__ out_alpha = c.out_alpha;
    const int tx = tile % vp.gx, ty = tile / vp.gx;
    const LanePixel lp = lane_pixel(lane, tx, ty, q);
    const int px = lp.px, py = lp.py, row = lp.row, li = lp.li;
    const bool inside = px < vp.W && py < vp.H;
    const float cx0 = (float)(px - (li & 3)), cy0 = (float)(py - (li >> 2));      // the row's cell: pixel centre of its column 0 / row 0
    const int pi4 = li & 3, pj4 = 4 + (li >> 2);                                   // this pixel's column / row piece of a pair block (float4 index)

    const uint32_t s = g.tile_start[tile];
    uint32_t nr = (s < cap) ? g.ccount[tile * kCells + lp.cell] : 0u;   // this row's list length
    const bool row_long = (s < cap) && g.cflag[tile * kCells + lp.cell] != 0u;   // blended by k_render_fwd_long
    if (row_long) nr = 0u;
    const uint32_t nmax = wave_max_u32(nr);
    set_priority_by_length(nmax);
    if (nmax < min_work) { trace.done(0); return; }
    const uint32_t *__restrict__ list = b.clist + (size_t)lp.cell * b.cap + s;
    float *row_base = s_p + row * kRowFloats;

    float T_ = 1.0f;
    f2v C01 = (f2v)(0.f), C23 = (f2v)(0.f), C45 = (f2v)(0.f), DW = (f2v)(0.f);   // colours | (depth, alpha) sums
    uint32_t lastj = 0;       // list position + 1 of the pixel's last contributor (n_contrib, counted in the CELL list)
    bool done = !inside | row_long;

    // two-deep prefetch: the LIST word of the chunk after next is loaded while the attributes of the next chunk are
    // gathered (list word -> attribute gather is a dependent pair of memory round trips; one chunk of blending is
    // shorter than the two of them)
    float4 r[4];
    zero_entry(r);
    if ((uint32_t)li < nr) gather_entry<C>(list[li], g, colors, r);
    uint32_t wnext = (kChunk + (uint32_t)li < nr) ? list[kChunk + li] : 0u;
    for (uint32_t c0 = 0; c0 < nmax; c0 += kChunk) {
        const int cnt = (c0 < nr) ? (int)min((uint32_t)kChunk, nr - c0) : 0;
        __builtin_amdgcn_wave_barrier();
        stage_entry(row_base, li, r, cx0, cy0);
        zero_entry(r);
        const uint32_t wcur = wnext;
        if (c0 + 2 * kChunk + (uint32_t)li < nr) wnext = list[c0 + 2 * kChunk + li];
        if (c0 + kChunk + (uint32_t)li < nr) gather_entry<C>(wcur, g, colors, r);   // prefetch
        __builtin_amdgcn_wave_barrier();
        if (__ballot((!done) & (cnt > 0)) == 0) break;   // every pixel with entries left is saturated
        int t = 0;
        do {
            // 2 * kFwdPairs entries per step.  Their alphas are independent and evaluated two per packed
            // instruction; the blend below is sequential and branch-free (selects, not exec-mask branches):
            // lanes that do not take an entry blend with weight 0, which leaves their accumulators
            // bit-identical; padding entries are inert.
            const f4v *P = reinterpret_cast<const f4v *>(row_base + (t >> 1) * kPairFloats);
            f4v ga[kFwdPairs], gc[kFwdPairs];
            f2v op[kFwdPairs];
#pragma unroll
            for (int j = 0; j < kFwdPairs; ++j) {
                ga[j] = P[kPair4 * j + pi4];
                gc[j] = P[kPair4 * j + pj4];
                op[j] = *reinterpret_cast<const f2v *>(reinterpret_cast<const float *>(P + kPair4 * j) + 32);
            }
            f2v pw[kFwdPairs], G[kFwdPairs], al[kFwdPairs];
            pair_gauss<kFwdPairs>(ga, gc, pw, G);
#pragma unroll
            for (int j = 0; j < kFwdPairs; ++j) {
                const f2v oa = op[j] * G[j];
                al[j] = f2v{fminf(0.99f, oa.x), fminf(0.99f, oa.y)};
            }
#pragma unroll
            for (int j = 0; j < kFwdPairs; ++j) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f4v e0 = P[kPair4 * j + 9 + 2 * h], e1 = P[kPair4 * j + 10 + 2 * h];
                    const float alpha = h ? al[j].y : al[j].x, power = h ? pw[j].y : pw[j].x;
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
        im.n_contrib[pid] = lastj;
        const float Cacc[6] = {C01.x, C01.y, C23.x, C23.y, C45.x, C45.y};
#pragma unroll
        for (int ch = 0; ch < C; ++ch) out_color[ch * P + pid] = __builtin_fmaf(T_, vp.bg[ch], Cacc[ch]);
        out_depth[pid] = DW.x;
        out_alpha[pid] = DW.y;
    }
    const uint32_t wj = row_max_u32(lastj);
    if (li == 0 && !row_long) g.cdone[tile * kCells + lp.cell] = wj;
    trace.done(nmax);
}


}  // namespace dm4d
