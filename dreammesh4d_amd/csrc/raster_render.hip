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

template <int C> struct StagedN { static constexpr int kVec = (C <= 3) ? 3 : 4; };
constexpr int kChunk = 16;   // list entries a row stages per step (one per lane of the row)
constexpr int kFwdUnroll = 4, kBwdUnroll = 2;   // entries per inner-loop step (divide kChunk)

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
// An all-zero entry is inert: opacity 0 gives alpha 0 < 1/255, so rows whose list is shorter than the
// wave's longest one blend padding entries with weight exactly 0.
__device__ __forceinline__ void zero_entry(float4 (&r)[4])
{
#pragma unroll
    for (int v = 0; v < 4; ++v) r[v] = make_float4(0.f, 0.f, 0.f, 0.f);
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
__device__ __forceinline__ uint32_t row_max_u32(uint32_t v)
{
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        const uint32_t w = __shfl_xor(v, o, 64);
        v = v > w ? v : w;
    }
    return v;
}

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
    constexpr int NV = StagedN<C>::kVec;
    // one wave-private staging buffer: the next chunk waits in registers (prefetched during the
    // blend loop) and is written after the loop -- same wave, program order, no hazard
    __shared__ float4 s_e[4][kChunk][NV];
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
    const float pxf = (float)px, pyf = (float)py;

    const uint32_t s = g.tile_start[tile];
    const uint32_t nr = (s < cap) ? g.ccount[tile * kCells + lp.cell] : 0u;   // this row's list length
    const uint32_t nmax = wave_max_u32(nr);
    set_priority_by_length(nmax);
    if (nmax < g_min_work) { trace.done(0); return; }
    const uint2 *__restrict__ list = b.clist + (size_t)lp.cell * b.cap + s;

    float T_ = 1.0f, D = 0.f, Wt = 0.f;
    float Cacc[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) Cacc[ch] = 0.f;
    uint32_t last = 0, lastj = 0;
    bool done = !inside;

    float4 r[4];
    zero_entry(r);
    if ((uint32_t)li < nr) gather_entry<C>(list[li], g, colors, r);
    for (uint32_t c0 = 0; c0 < nmax; c0 += kChunk) {
        const int cnt = (c0 < nr) ? (int)min((uint32_t)kChunk, nr - c0) : 0;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int v = 0; v < NV; ++v) s_e[row][li][v] = r[v];
        zero_entry(r);
        if (c0 + kChunk + (uint32_t)li < nr) gather_entry<C>(list[c0 + kChunk + li], g, colors, r);   // prefetch
        __builtin_amdgcn_wave_barrier();
        if (__ballot((!done) & (cnt > 0)) == 0) break;   // every pixel with entries left is saturated
        int t = 0;
        do {
            // kFwdUnroll entries per step: their alphas are independent (a lone wave on a long silhouette list is
            // bound by the ~25-deep dependent chain of one alpha, not by issue), the blend below is sequential.
            // Branch-free (selects, not exec-mask branches): lanes that do not take an entry blend with
            // weight 0, which leaves their accumulators bit-identical; padding entries are inert.
            float al[kFwdUnroll], pw[kFwdUnroll], ex[kFwdUnroll], op[kFwdUnroll];
            bool ok[kFwdUnroll];
            {
                float dx[kFwdUnroll], dy[kFwdUnroll], qa[kFwdUnroll], qb[kFwdUnroll], qc[kFwdUnroll];
#pragma unroll
                for (int j = 0; j < kFwdUnroll; ++j) {
                    const float4 ea = s_e[row][t + j][0], eb = s_e[row][t + j][1];
                    dx[j] = ea.x - pxf;
                    dy[j] = ea.y - pyf;
                    qa[j] = ea.z; qb[j] = ea.w; qc[j] = eb.x; op[j] = eb.y;
                }
                float u[kFwdUnroll], v[kFwdUnroll], w2[kFwdUnroll];
#pragma unroll
                for (int j = 0; j < kFwdUnroll; ++j) { u[j] = qa[j] * dx[j]; v[j] = qc[j] * dy[j]; w2[j] = qb[j] * dx[j]; }
#pragma unroll
                for (int j = 0; j < kFwdUnroll; ++j) { u[j] = u[j] * dx[j]; v[j] = v[j] * dy[j]; w2[j] = w2[j] * dy[j]; }
#pragma unroll
                for (int j = 0; j < kFwdUnroll; ++j) u[j] = u[j] + v[j];
#pragma unroll
                for (int j = 0; j < kFwdUnroll; ++j) pw[j] = -0.5f * u[j] - w2[j];
            }
            det_expf_n<kFwdUnroll>(pw, ex);
#pragma unroll
            for (int j = 0; j < kFwdUnroll; ++j) al[j] = fminf(0.99f, op[j] * ex[j]);
#pragma unroll
            for (int j = 0; j < kFwdUnroll; ++j) ok[j] = (pw[j] <= 0.0f) & (al[j] >= 1.0f / 255.0f);
#pragma unroll
            for (int j = 0; j < kFwdUnroll; ++j) {
                const float4 eb = s_e[row][t + j][1], ec = s_e[row][t + j][2];
                const float alpha = al[j];
                const float test_T = T_ * (1.0f - alpha);
                const bool valid = (!done) & ok[j];
                const bool stop = valid & (test_T < 0.0001f);
                const bool contrib = valid & (!stop);
                const float w = contrib ? alpha * T_ : 0.f;
                Cacc[0] = __builtin_fmaf(ec.x, w, Cacc[0]);
                Cacc[1] = __builtin_fmaf(ec.y, w, Cacc[1]);
                Cacc[2] = __builtin_fmaf(ec.z, w, Cacc[2]);
                if (C > 3) {
                    const float4 ed = s_e[row][t + j][NV - 1];
                    Cacc[3] = __builtin_fmaf(ec.w, w, Cacc[3]);
                    Cacc[C > 4 ? 4 : 0] = __builtin_fmaf(ed.x, w, Cacc[C > 4 ? 4 : 0]);
                    Cacc[C > 5 ? 5 : 0] = __builtin_fmaf(ed.y, w, Cacc[C > 5 ? 5 : 0]);
                }
                D = __builtin_fmaf(eb.z, w, D);
                Wt = Wt + w;
                T_ = contrib ? test_T : T_;
                last = contrib ? __float_as_uint(eb.w) + 1u : last;
                lastj = contrib ? c0 + (uint32_t)(t + j) + 1u : lastj;
                done = done | stop;
            }
            t += kFwdUnroll;
        } while (t < kChunk && __ballot((!done) & (t < cnt)) != 0);
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
    const uint32_t wj = row_max_u32(lastj), wk = row_max_u32(last);
    if (li == 0) {
        g.cdone[tile * kCells + lp.cell] = wj;
        g.ckmax[tile * kCells + lp.cell] = wk;
    }
    trace.done(nmax);
}

// ---------------------------------------------------------------------------------------- B1
typedef float f2v __attribute__((ext_vector_type(2)));
constexpr int kRedStride = 68;   // floats per value row of the transposed reduction buffer (64 lanes + pad)

template <int C>
__global__ __launch_bounds__(64) void k_render_bwd(BatchDesc d)
{
    constexpr int NV = StagedN<C>::kVec;
    constexpr int RS = 7 + C;             // values per record
    constexpr int RSP = (C <= 3) ? 12 : 16;   // == grad_stride(C): floats per (padded) record
    __shared__ float4 s_e[4][kChunk][NV];
    __shared__ uint32_t s_slot[4][kChunk];
    __shared__ __attribute__((aligned(16))) float s_red[kBwdUnroll][RS][kRedStride];
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
    const float *__restrict__ dL_dcolor = c.dL_dcolor, *__restrict__ dL_ddepth = c.dL_ddepth,
                             *__restrict__ dL_dalpha = c.dL_dalpha;
    float *__restrict__ rec = c.dLq;
    const uint32_t rec_cap = c.rec_cap;
    const int lane = threadIdx.x;
    const int tx = tile % vp.gx, ty = tile / vp.gx;
    const LanePixel lp = lane_pixel(lane, tx, ty, q);
    const int px = lp.px, py = lp.py, row = lp.row, li = lp.li;
    const bool inside = px < vp.W && py < vp.H;
    const float pxf = (float)px, pyf = (float)py;

    const uint32_t s = g.tile_start[tile];
    const uint32_t nd = (s < cap) ? g.cdone[tile * kCells + lp.cell] : 0u;   // entries this row's forward consumed
    const uint32_t ndmax = wave_max_u32(nd);
    const uint2 *__restrict__ list = b.clist + (size_t)lp.cell * b.cap + s;
    const uint32_t *__restrict__ slots = b.cslot + (size_t)lp.cell * b.cap + s;
    {
        // entries the forward never reached get all-zero records, so that B2 can sum every Gaussian's
        // contiguous record block without looking anything up
        const uint32_t nr = (s < cap) ? g.ccount[tile * kCells + lp.cell] : 0u;
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
    // reduction role of this lane: value li of its row (lanes with li >= RS idle in the sum)
    const int red_i = li < RS ? li : 0;
    const float4 *red_src = reinterpret_cast<const float4 *>(&s_red[0][red_i][row * 16]);
    constexpr int kRedBuf4 = RS * kRedStride / 4;   // float4 per reduction buffer

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
#pragma unroll
        for (int v = 0; v < NV; ++v) s_e[row][li][v] = r[v];
        s_slot[row][li] = rslot;
        zero_entry(r);
        if (c0 >= (uint32_t)kChunk && c0 - kChunk + (uint32_t)li < nd) {   // prefetch the chunk in front
            gather_entry<C>(list[c0 - kChunk + li], g, colors, r);
            rslot = slots[c0 - kChunk + li];
        }
        __builtin_amdgcn_wave_barrier();
        // kBwdUnroll entries per step (aligned groups, highest first; slots past the row's count hold inert
        // padding): independent alphas overlap, and one LDS round trip serves the whole group.
        for (int tg = ((tmax - 1) / kBwdUnroll) * kBwdUnroll; tg >= 0; tg -= kBwdUnroll) {
            float Gr_[kBwdUnroll], al_[kBwdUnroll], dx_[kBwdUnroll], dy_[kBwdUnroll];
            bool ok_[kBwdUnroll];
#pragma unroll
            for (int j = 0; j < kBwdUnroll; ++j) {
                const int t = tg + kBwdUnroll - 1 - j;
                const float4 ea = s_e[row][t][0], eb = s_e[row][t][1];
                dx_[j] = ea.x - pxf;
                dy_[j] = ea.y - pyf;
                const float power = -0.5f * ((ea.z * dx_[j]) * dx_[j] + (eb.x * dy_[j]) * dy_[j]) - (ea.w * dx_[j]) * dy_[j];
                Gr_[j] = det_expf(power);
                al_[j] = fminf(0.99f, eb.y * Gr_[j]);
                ok_[j] = (__float_as_uint(eb.w) < last) & (power <= 0.0f) & (al_[j] >= 1.0f / 255.0f);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < kBwdUnroll; ++j) {
                const int t = tg + kBwdUnroll - 1 - j;
                const float4 ea = s_e[row][t][0], eb = s_e[row][t][1], ec = s_e[row][t][2];
                // Branch-free: lanes that do not take the entry contribute exact zeros.
                const float dx = dx_[j], dy = dy_[j], alpha = al_[j];
                const bool contrib = ok_[j];
                float col[C];
                col[0] = ec.x; col[1] = ec.y; col[2] = ec.z;
                if (C > 3) {
                    const float4 ed = s_e[row][t][NV - 1];
                    col[3] = ec.w;
                    col[C > 4 ? 4 : 0] = ed.x;
                    col[C > 5 ? 5 : 0] = ed.y;
                }
                const float G = contrib ? Gr_[j] : 0.f;
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
                float v[RS];
                v[0] = dL_dG * (-gdx * ea.z - gdy * ea.w) * half_W;
                v[1] = dL_dG * (-gdy * eb.x - gdx * ea.w) * half_H;
                v[2] = -0.5f * gdx * dx * dL_dG;
                v[3] = -gdx * dy * dL_dG;
                v[4] = -0.5f * gdy * dy * dL_dG;
                v[5] = G * dL_da;
                v[6] = w * gD;
#pragma unroll
                for (int ch = 0; ch < C; ++ch) v[7 + ch] = w * gCol[ch];
                // transposed reduction: [value][lane] in LDS, lane i of the row sums value i over the row
#pragma unroll
                for (int i = 0; i < RS; ++i) s_red[j][i][lane] = v[i];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < kBwdUnroll; ++j) {
                const int t = tg + kBwdUnroll - 1 - j;
                const float4 *src = red_src + j * kRedBuf4;
                const float4 a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3];
                // fixed summation tree (deterministic): pairs of packed adds
                f2v p0 = f2v{a0.x, a0.y} + f2v{a0.z, a0.w};
                f2v p1 = f2v{a1.x, a1.y} + f2v{a1.z, a1.w};
                f2v p2 = f2v{a2.x, a2.y} + f2v{a2.z, a2.w};
                f2v p3 = f2v{a3.x, a3.y} + f2v{a3.z, a3.w};
                p0 = p0 + p1;
                p2 = p2 + p3;
                p0 = p0 + p2;
                const float total = li < RS ? p0.x + p0.y : 0.f;   // lanes RS..RSP-1 write the padding
                const uint32_t slot = s_slot[row][t];
                if (li < RSP && t < cnt && slot < rec_cap) rec[(size_t)slot * RSP + li] = total;
            }
        }
        if (c0 == 0) break;
    }
    trace.done(ndmax);
}

// ---------------------------------------------------------------------------------------- launchers
int launch_render_fwd(const BatchDesc &d, hipStream_t st)
{
    const int T = ((d.W + kTile - 1) / kTile) * ((d.H + kTile - 1) / kTile);
    if (T <= 0) return DM4D_OK;
    const int blocks = (int)((((int64_t)T * d.B + 7) / 8) * 8 * 4);
    ProfScope prof_(kKRenderFwd, st);
    if (d.C <= 3) hipLaunchKernelGGL(k_render_fwd<3>, dim3(blocks), dim3(64), 0, st, d);
    else hipLaunchKernelGGL(k_render_fwd<6>, dim3(blocks), dim3(64), 0, st, d);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int launch_render_bwd(const BatchDesc &d, hipStream_t st)
{
    const int T = ((d.W + kTile - 1) / kTile) * ((d.H + kTile - 1) / kTile);
    if (T <= 0) return DM4D_OK;
    const int blocks = (int)((((int64_t)T * d.B + 7) / 8) * 8 * 4);
    ProfScope prof_(kKRenderBwd, st);
    if (d.C <= 3) hipLaunchKernelGGL(k_render_bwd<3>, dim3(blocks), dim3(64), 0, st, d);
    else hipLaunchKernelGGL(k_render_bwd<6>, dim3(blocks), dim3(64), 0, st, d);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // namespace dm4d
