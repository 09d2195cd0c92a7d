// raster.h -- workspace layout and kernel launchers of the tile rasterizer.
//
// Pipeline (one view), all on the caller's stream, no host sync inside:
//   K1 preprocess      1024 Gaussians / WG   cull, cov3D, EWA cov2D, conic, radius, rect;
//                                            per-WG tile histogram in LDS -> dense hist[WG][tile]
//   K2 colscan         1 thread / tile       exclusive scan of hist over WGs, tile counts
//   K3 scatter         1024 Gaussians / WG   duplicates -> per-tile segments via LDS cursors
//                                            (no global atomics anywhere in binning)
//   K4 tile_sort       1 workgroup / tile    LDS depth-bucket sort by (depth bits, id); then the
//                                            tile's list is split into sixteen 4x4-pixel CELL lists
//   K5 render_fwd      1 WAVE / 8x8 quadrant one 16-lane DPP row per cell, each row walks its own
//                                            depth-sorted cell list; no barriers, early exit
//   B1 render_bwd      1 WAVE / 8x8 quadrant per-(Gaussian, cell) partial-gradient records, no atomics
//   B2 gather_bwd      1 thread / Gaussian   deterministic gather + preprocess backward
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/dm4d.h"

#ifdef __HIPCC__
#define DM4D_HD __host__ __device__
#else
#define DM4D_HD
#endif

namespace dm4d {

constexpr int kTile = DM4D_TILE;
constexpr int kPreThreads = 256; // threads per workgroup in K1/K3/B2
constexpr int kPreItems = 4;     // Gaussians per thread in K1/K3
constexpr int kPreBlock = kPreThreads * kPreItems;   // Gaussians per workgroup in K1/K3
constexpr int kMaxTiles = 36864; // tile histogram lives in LDS (4 B/tile of the 160 KB): <= 3072x3072 px
constexpr int kMaxChannels = 6;  // colour channels blended per pass: 3 (drop-in operator) or 6 (RGB + normal)
constexpr int kCells = 16;       // 4x4-pixel cells per 16x16 tile; cell id = 4 * quadrant + (cx & 1) + 2 * (cy & 1)
// floats per (Gaussian, cell) record of the backward scratch: 2 mean + 3 conic + opacity + depth + C colours,
// padded to whole float4s (C = 6: one aligned 64-byte line per record -- measured: 52-byte records cost MORE
// HBM write traffic than 64-byte ones, partial-line writes)
// floats per backward record.  lean = 1 (6-channel batched path with the static appearance frozen): 9 values
// (mean2D 2 | conic 3 | depth | colour channels 3..5) instead of 13 (+ opacity, colour channels 0..2); lean = 2: the same
// WITHOUT a depth gradient (dL_ddepth == NULL: the shipped dynamic configuration has no depth loss,
// configs/sugar_dynamic_dg.yaml:142-154) -- 8 values = 32 bytes, two 16-byte pieces per record instead of three.
DM4D_HD static inline int grad_stride(int C, int lean = 0) { return lean >= 2 ? 8 : (C <= 3 || lean) ? 12 : 16; }

// counters[]: duplicates, duplicate-capacity overflow, records (sum of the Gaussians' cells), record-capacity overflow
// kCntLong: number of LONG cells (cell lists of >= kLongCell entries, K4 appends them to `longlist`)
// kCntLongEarly: the long cells of the tiles the large sort variant handled (`earlylist`), see k_render_fwd_long
enum GeomCounter { kCntD = 0, kCntOverflow = 1, kCntR = 2, kCntRecOverflow = 3, kCntLong = 4, kCntLongEarly = 5 };
// A wave of the blend kernels walks four cell lists side by side, one entry per row and step, so the launch cannot
// end before its longest list (silhouette cells hold > 1000 entries against a mean of 75: measured, the longest
// wave alone took as long as the whole launch).  Cells with at least this many entries are therefore left out of
// the regular kernels and blended by k_render_{fwd,bwd}_long: one wave per cell, its four rows evaluating four
// CONSECUTIVE entries of the one list for the same 16 pixels, the sequential transmittance chain run by row 0.
constexpr uint32_t kLongCell = 384;
constexpr int kGidBits = 25;           // cell-list word: Gaussian id (N <= 2^25) | record rank << 25
constexpr uint32_t kRankBig = 127u;    // rank field: index of the cell's record among the Gaussian's records, or kRankBig: a dense block of
                                       // more cells than the field holds, the blend backward computes the index from cellinfo
constexpr uint32_t kGidMask = (1u << kGidBits) - 1u;
// The entry-parallel backward gives cells with at least this many entries a whole wave (64 entries per step) instead
// of a DPP row (16 per step): `longlist` holds them.
// (160: same-box A/B of 128 / 160 / 176 / 192 / 224 / 256 / 384 by bench.py's step, tools/ab_many.sh -- 64 and 96 double the kernel: a wide
// block's six-step scans and one-cell table cost more per entry than a row's)
#ifndef DM4D_WIDE_BWD
#define DM4D_WIDE_BWD 160
#endif
constexpr uint32_t kWideBwd = DM4D_WIDE_BWD;
// tile-record mode: (Gaussian, tile) accumulators a workgroup of k_render_bwd_tile holds in LDS at a time (a window of the
// tile list; longer lists take several windows, back to front)
#ifndef DM4D_TILE_WINDOW
#define DM4D_TILE_WINDOW 496
#endif
constexpr int kTileWindow = DM4D_TILE_WINDOW;
// tiles a Gaussian's cell block spans (tile-record mode: one record each, row-major from (tx0, ty0))
struct TileSpan { int tx0, ty0, tnx, tny; };
DM4D_HD static inline TileSpan tile_span(uint32_t cellinfo_x, uint32_t cellinfo_y)
{
    const int bx0 = (int)(cellinfo_x & 0xFFFFu), by0 = (int)(cellinfo_x >> 16);
    const int nbx = (int)(cellinfo_y & 0xFFFFu), nby = (int)(cellinfo_y >> 16);
    TileSpan t;
    t.tx0 = bx0 >> 2; t.ty0 = by0 >> 2;
    t.tnx = (nbx > 0 && nby > 0) ? ((bx0 + nbx - 1) >> 2) - t.tx0 + 1 : 0;
    t.tny = (nbx > 0 && nby > 0) ? ((by0 + nby - 1) >> 2) - t.ty0 + 1 : 0;
    return t;
}

DM4D_HD static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct GeomLayout {
    int N, T, nb;
    size_t counters, xy, depth, conic_opacity, rgb, tiles_touched, rec_touched, rec0, cellinfo, cellmask, clamped,
        block_sums, rec_block_sums, hist, tile_count, tile_start, ccount, cdone, order, longlist, earlylist, cflag, zero_begin, zero_bytes, total;
};

DM4D_HD static inline size_t take_(size_t &o, size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; }

DM4D_HD static inline GeomLayout geom_layout(int N, int H, int W)
{
    GeomLayout L;
    L.N = N;
    int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
    L.T = gx * gy;
    L.nb = (N + kPreBlock - 1) / kPreBlock;
    size_t o = 0;
    size_t n = (size_t)(N > 0 ? N : 1);
    // counters (offset 0) are cleared by one small memset node before K1
    L.counters = take_(o, 64 * 4);
    L.zero_begin = 0;
    L.zero_bytes = o;
    L.tile_count = take_(o, (size_t)L.T * 4);
    L.tile_start = take_(o, (size_t)(L.T + 1) * 4);
    L.ccount = take_(o, (size_t)L.T * kCells * 4);   // entries of each cell list                  [T][16]
    L.cdone = take_(o, (size_t)L.T * kCells * 4);    // entries the forward consumed               [T][16]
    L.order = take_(o, (size_t)L.T * 4);             // tiles by descending list length (blend launch order)
    L.longlist = take_(o, (size_t)L.T * kCells * 4); // tile * 16 + cell of the long cells, in no particular order
    L.earlylist = take_(o, (size_t)L.T * kCells * 4);   // the long cells of the tiles sorted by the large variant
    L.cflag = take_(o, (size_t)L.T * kCells * 4);       // 1: the cell is on earlylist (its forward is k_render_fwd_long's)
    L.xy = take_(o, n * 8);
    L.depth = take_(o, n * 4);
    L.conic_opacity = take_(o, n * 16);
    L.rgb = take_(o, n * 12);
    L.tiles_touched = take_(o, n * 4);
    L.rec_touched = take_(o, n * 4);
    L.rec0 = take_(o, n * 4);
    L.cellinfo = take_(o, n * 16);
    L.cellmask = take_(o, n * 8);
    L.clamped = take_(o, n * 3);
    L.block_sums = take_(o, (size_t)(L.nb + 1) * 4);
    L.rec_block_sums = take_(o, (size_t)(L.nb + 1) * 4);
    L.hist = take_(o, (size_t)(L.nb > 0 ? L.nb : 1) * L.T * 4);
    L.total = o;
    return L;
}

struct GeomPtrs {
    uint32_t *counters;
    float2 *xy;
    float *depth;
    float4 *conic_opacity;
    float *rgb;
    uint32_t *tiles_touched;
    uint32_t *rec_touched;   // cells (4x4 px) inside the Gaussian's alpha >= 1/255 bound and its tile rect
    uint32_t *rec0;          // first backward record of the Gaussian (== cellinfo.z; compact copy for the blend backward's gathers)
    uint4 *cellinfo;         // {bx0 | by0 << 16, nbx | nby << 16, first backward record (scan of rec_touched), dense}
    uint64_t *cellmask;      // cells of the block the splat really reaches (bit = by * nbx + bx), blocks of <= 64 cells
    uint8_t *clamped;
    uint32_t *block_sums;
    uint32_t *rec_block_sums;
    uint32_t *hist;        // [nb][T] per-WG tile histogram, exclusive-scanned over WGs in place by K2
    uint32_t *tile_count;
    uint32_t *tile_start;
    uint32_t *ccount;
    uint32_t *cdone;
    uint32_t *order;
    uint32_t *longlist;
    uint32_t *earlylist;
    uint32_t *cflag;
};

DM4D_HD static inline GeomPtrs geom_ptrs(void *base, const GeomLayout &L)
{
    char *b = (char *)base;
    GeomPtrs p;
    p.counters = (uint32_t *)(b + L.counters);
    p.xy = (float2 *)(b + L.xy);
    p.depth = (float *)(b + L.depth);
    p.conic_opacity = (float4 *)(b + L.conic_opacity);
    p.rgb = (float *)(b + L.rgb);
    p.tiles_touched = (uint32_t *)(b + L.tiles_touched);
    p.rec_touched = (uint32_t *)(b + L.rec_touched);
    p.rec0 = (uint32_t *)(b + L.rec0);
    p.cellinfo = (uint4 *)(b + L.cellinfo);
    p.cellmask = (uint64_t *)(b + L.cellmask);
    p.clamped = (uint8_t *)(b + L.clamped);
    p.block_sums = (uint32_t *)(b + L.block_sums);
    p.rec_block_sums = (uint32_t *)(b + L.rec_block_sums);
    p.hist = (uint32_t *)(b + L.hist);
    p.tile_count = (uint32_t *)(b + L.tile_count);
    p.tile_start = (uint32_t *)(b + L.tile_start);
    p.ccount = (uint32_t *)(b + L.ccount);
    p.cdone = (uint32_t *)(b + L.cdone);
    p.order = (uint32_t *)(b + L.order);
    p.longlist = (uint32_t *)(b + L.longlist);
    p.earlylist = (uint32_t *)(b + L.earlylist);
    p.cflag = (uint32_t *)(b + L.cflag);
    return p;
}

struct BinPtrs {
    uint32_t *u_depth;    // unsorted duplicates, tile-major segments
    uint32_t *u_idx;
    uint32_t *point_list; // sorted Gaussian ids  (== upstream point_list)
    uint32_t *clist;      // [16][cap] cell lists, cell c of tile t at clist[c*cap + tile_start[t] ...], in tile-list order.
                          // One word per entry: Gaussian id | rank << kGidBits, rank = which of the Gaussian's backward
                          // records (cellinfo.z + rank) belongs to this cell.  Positions in a cell list stand in for the
                          // tile-list positions upstream counts with (n_contrib): the list is a subsequence of the tile list.
    uint16_t *cpos;       // [16][cap], parallel to clist, written in tile-record mode only: the entry's position in the TILE list
                          // (the slot of its (Gaussian, tile) accumulator in the LDS of k_render_bwd_tile)
    size_t cap;
};
DM4D_HD static inline size_t binning_bytes(int64_t cap)
{
    size_t c = (size_t)(cap > 0 ? cap : 1);
    return 3 * align_up(c * 4, 256) + align_up(c * kCells * 4, 256) + align_up(c * kCells * 2, 256);
}
DM4D_HD static inline BinPtrs bin_ptrs(void *base, int64_t cap)
{
    size_t c = (size_t)(cap > 0 ? cap : 1);
    size_t stride = align_up(c * 4, 256);
    char *b = (char *)base;
    BinPtrs p;
    p.u_depth = (uint32_t *)(b);
    p.u_idx = (uint32_t *)(b + stride);
    p.point_list = (uint32_t *)(b + 2 * stride);
    p.clist = (uint32_t *)(b + 3 * stride);
    p.cpos = (uint16_t *)(b + 3 * stride + align_up(c * kCells * 4, 256));
    p.cap = c;
    return p;
}

struct ImgPtrs {
    float *final_T;
    uint32_t *n_contrib;
};
DM4D_HD static inline size_t image_bytes(int H, int W)
{
    size_t P = (size_t)H * W;
    return 2 * align_up((P > 0 ? P : 1) * 4, 256);
}
DM4D_HD static inline ImgPtrs img_ptrs(void *base, int H, int W)
{
    size_t P = (size_t)H * W;
    size_t stride = align_up((P > 0 ? P : 1) * 4, 256);
    ImgPtrs p;
    p.final_T = (float *)base;
    p.n_contrib = (uint32_t *)((char *)base + stride);
    return p;
}
DM4D_HD static inline size_t grad_bytes(int64_t n_records, int C)
{
    size_t c = (size_t)(n_records > 0 ? n_records : 1);
    return align_up(c * grad_stride(C) * 4, 256);
}

// Camera / image constants handed to kernels by value.
struct ViewParams {
    int W, H, gx, gy, C;   // C = colour channels (3 or 6)
    float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
    const float *bg, *view, *proj, *campos;
};

struct Rect { int x0, y0, x1, y1; };   // tile rect of a Gaussian, [x0, x1) x [y0, y1)

// The 4x4-pixel cells a Gaussian can contribute to form a dense nbx x nby block of the global cell
// grid (cell (bx, by) = pixels 4bx..4bx+3 x 4by..4by+3; tile = (bx >> 2, by >> 2)).
struct Bands { int bx0, by0, nbx, nby; };

#ifdef __HIPCC__
__device__ __forceinline__ Rect tile_rect(float px, float py, int r, int gx, int gy)
{
    float fr = (float)r;
    Rect q;
    q.x0 = min(gx, max(0, f2i_sat((px - fr) / (float)kTile)));
    q.y0 = min(gy, max(0, f2i_sat((py - fr) / (float)kTile)));
    q.x1 = min(gx, max(0, f2i_sat((px + fr + (float)(kTile - 1)) / (float)kTile)));
    q.y1 = min(gy, max(0, f2i_sat((py + fr + (float)(kTile - 1)) / (float)kTile)));
    return q;
}

// Cells of the tile rect `rc` that the alpha >= 1/255 support of a splat can reach: exact axis-aligned
// bound of the ellipse { d : 1/2 d^T A d <= ln(255 o) }, inflated by margins that cover the rounding
// of log/sqrt/div.  Conservative, so culled (splat, pixel) pairs are exactly ones the reference
// `continue`s on.  Evaluated once per Gaussian in K1 and stored (cellinfo).
__device__ __forceinline__ Bands cell_bands(float x, float y, float ca, float cb, float cc, float o, const Rect rc)
{
    Bands B;
    B.bx0 = 4 * rc.x0; B.by0 = 4 * rc.y0; B.nbx = 0; B.nby = 0;
    if (o < 1.0f / 255.0f) return B;   // alpha <= opacity < 1/255 for every pixel
    int bx1 = 4 * rc.x1, by1 = 4 * rc.y1;
    const float tau = __logf(255.0f * o) * 1.001f + 0.01f;
    const float det = ca * cc - cb * cb;
    // (hardware sqrt / rcp, 1 ulp each: the 1e-4 margins are there for exactly this)
    const float t2 = 2.0f * tau * __builtin_amdgcn_rcpf(det);
    const float hx = __builtin_amdgcn_sqrtf(t2 * cc) * 1.0001f + 0.02f;
    const float hy = __builtin_amdgcn_sqrtf(t2 * ca) * 1.0001f + 0.02f;
    if ((det > 0.f) && (hx == hx) && (hy == hy)) {
        // cell column b holds pixel centres 4b .. 4b+3: reached iff x + hx >= 4b and x - hx <= 4b + 3
        B.bx0 = max(B.bx0, f2i_sat(ceilf((x - hx - 3.0f) * 0.25f)));
        B.by0 = max(B.by0, f2i_sat(ceilf((y - hy - 3.0f) * 0.25f)));
        bx1 = min(bx1 - 1, f2i_sat(floorf((x + hx) * 0.25f))) + 1;
        by1 = min(by1 - 1, f2i_sat(floorf((y + hy) * 0.25f))) + 1;
    }
    const int nbx = bx1 - B.bx0, nby = by1 - B.by0;
    if (nbx > 0 && nby > 0) { B.nbx = nbx; B.nby = nby; }
    return B;
}
// Exact refinement of cell_bands, one ROW of cells at a time.  The ellipse E = { d : A dx^2 + 2 B dx dy + C dy^2 <= L }
// around (x, y) meets the cell [cx0, cx0 + 3] x [cy0, cy0 + 3] (pixel centres, +- 0.02) iff the cell's column
// range overlaps the x-extent of E intersected with the row's strip -- E and the strip are convex, so that
// intersection projects onto an interval [xlo, xhi].  On the strip the right boundary xr(dy) = (-B dy + sqrt(A L -
// det dy^2)) / A is concave and peaks at dy = -B sqrt(L / (det C)) (the rightmost point of E), the left boundary is
// its mirror image; clamping those two ordinates to the strip gives the extent with two square roots per ROW,
// where testing every cell against the four edges cost ~45 operations per CELL (36 of K1's 78 us).  Margins as in
// cell_bands (they dwarf the rounding of the blend kernels' own power / exp): a culled cell holds no pixel with
// alpha >= 1/255.
struct EllipseRows { float AL, det, ystar, yext, invA, B; };
__device__ __forceinline__ EllipseRows ellipse_rows(float A, float B, float C, float tau)
{
    // hardware sqrt / rcp (1 ulp): the 1e-4 relative margins of row_span / yext cover them
    EllipseRows e;
    const float L = 2.0f * tau * 1.0001f + 0.001f;
    e.det = A * C - B * B;
    const float L_det = L * __builtin_amdgcn_rcpf(e.det);
    e.ystar = B * __builtin_amdgcn_sqrtf(L_det * __builtin_amdgcn_rcpf(C));     // |ordinate| of the leftmost / rightmost point
    e.yext = __builtin_amdgcn_sqrtf(A * L_det) * 1.0002f;  // half extent in y
    e.invA = __builtin_amdgcn_rcpf(A);
    e.AL = A * L;
    e.B = B;
    return e;
}
// cell columns [b0, b1] (absolute cell indices, possibly empty: b0 > b1) of the row of cells whose pixel centres are
// cy0 .. cy0 + 3 that the ellipse around (x, y) reaches
__device__ __forceinline__ void row_span(const EllipseRows &e, float x, float y, float cy0, int &b0, int &b1)
{
    const float ya = fmaxf(cy0 - 0.02f - y, -e.yext), yb = fminf(cy0 + 3.02f - y, e.yext);
    b0 = 1; b1 = 0;
    if (!(ya <= yb)) return;
    const float yr = fminf(yb, fmaxf(ya, -e.ystar)), yl = fminf(yb, fmaxf(ya, e.ystar));
    const float xhi = (-e.B * yr + __builtin_amdgcn_sqrtf(fmaxf(0.f, e.AL - e.det * yr * yr))) * e.invA;
    const float xlo = (-e.B * yl - __builtin_amdgcn_sqrtf(fmaxf(0.f, e.AL - e.det * yl * yl))) * e.invA;
    const float m = 1.0e-4f * (fabsf(xhi) + fabsf(xlo)) + 0.02f;      // rounding of the two roots + the pixel margin
    b0 = f2i_sat(ceilf((x + xlo - m - 3.0f) * 0.25f));
    b1 = f2i_sat(floorf((x + xhi + m) * 0.25f));
}
// cell id inside its tile (cx, cy in 0..3): the four cells of an 8x8 quadrant are consecutive
__device__ __forceinline__ int cell_id(int cx, int cy) { return 4 * ((cx >> 1) + 2 * (cy >> 1)) + (cx & 1) + 2 * (cy & 1); }
#endif

struct BwdOutputs {
    float *dL_dmeans2D, *dL_dmeans3D, *dL_dopacity, *dL_dcolors, *dL_dsh, *dL_dscales, *dL_drotations, *dL_dcov3D;
};

// A batch of B views that share N, the image size, the intrinsics and (optionally, stride 0) some
// inputs.  Every per-view quantity is `base + b * stride`; kernels use blockIdx.y as the view index,
// so ONE launch covers the whole batch (>> 256 workgroups keep all 8 XCDs busy and average out the
// per-tile load imbalance of a single small image).  The single-view C API is the B = 1 case.
struct BatchDesc {
    int B, N, C, W, H, sh_coeffs;
    const int32_t *frame_index;   // optional [B] (device): view -> frame whose means / rotations / colours it renders
    int scales_by_frame;          // scales are per FRAME (indexed like means3D) instead of per view / shared
    float tanfovx, tanfovy, scale_modifier;
    const float *bg;
    const float *view, *proj, *campos; size_t cam_stride, campos_stride;
    const float *means3D; size_t means_stride;
    const float *rotations; size_t rot_stride;
    const float *scales; size_t scale_stride;
    const float *opacities; size_t opac_stride;
    const float *colors; size_t color_stride;
    const float *shs; size_t sh_stride;
    const float *cov3D; size_t cov_stride;
    int32_t *radii; size_t radii_stride;
    char *geom; size_t geom_stride;
    char *binning; size_t bin_stride; uint32_t cap;
    char *image; size_t img_stride;
    float *out_color, *out_depth, *out_alpha;
    const float *dL_dcolor, *dL_ddepth, *dL_dalpha;
    float *dLq; size_t dlq_stride; uint32_t rec_cap;   // backward records: capacity (records) per view
    int lean;              // backward: 0 full | 1 lean | 2 lean without depth gradient | 3 = 2 and no gradient on colour channels 3..5 (see grad_stride); lean requires C == 6
    int tile_records;      // 1: ONE backward record per (Gaussian, tile) instead of per (Gaussian, cell): K1 counts tiles, K4
                           // also writes cpos, the blend backward is k_render_bwd_tile (a workgroup per tile sums its
                           // sixteen cells' records in LDS with ds_add_f32: the order of the additions, and so the last
                           // bits of the gradients, vary from run to run).  Must be the same for forward and backward.
    BwdOutputs o;          // per-view stride of each = N * width
};

struct ViewCtx {
    ViewParams vp;
    dm4d_raster_inputs in;
    int32_t *radii;
    GeomPtrs g;
    BinPtrs b;
    ImgPtrs im;
    float *out_color, *out_depth, *out_alpha;
    const float *dL_dcolor, *dL_ddepth, *dL_dalpha;
    float *dLq;
    uint32_t rec_cap;
    BwdOutputs o;
    const float *colors;   // colours actually blended ([N,C]): colors_precomp or the SH-evaluated rgb
    int T;
    uint32_t cap;
    int trec;              // BatchDesc::tile_records
};

DM4D_HD static inline ViewCtx resolve(const BatchDesc &d, int b)
{
    ViewCtx c;
    const size_t sb = (size_t)b, N = (size_t)d.N, P = (size_t)d.H * d.W;
    const size_t sf = d.frame_index ? (size_t)d.frame_index[b] : sb;   // device-only when frame_index is set
    c.vp.W = d.W; c.vp.H = d.H; c.vp.C = d.C;
    c.vp.gx = (d.W + kTile - 1) / kTile;
    c.vp.gy = (d.H + kTile - 1) / kTile;
    c.vp.tanfovx = d.tanfovx; c.vp.tanfovy = d.tanfovy;
    c.vp.focal_y = (float)d.H / (2.0f * d.tanfovy);
    c.vp.focal_x = (float)d.W / (2.0f * d.tanfovx);
    c.vp.scale_modifier = d.scale_modifier;
    c.vp.bg = d.bg;
    c.vp.view = d.view + sb * d.cam_stride;
    c.vp.proj = d.proj + sb * d.cam_stride;
    c.vp.campos = d.campos ? d.campos + sb * d.campos_stride : nullptr;
    c.T = c.vp.gx * c.vp.gy;
    c.in.N = d.N; c.in.sh_coeffs = d.sh_coeffs; c.in.n_channels = d.C;
    c.in.means3D = d.means3D ? d.means3D + sf * d.means_stride : nullptr;
    c.in.rotations = d.rotations ? d.rotations + sf * d.rot_stride : nullptr;
    c.in.scales = d.scales ? d.scales + (d.scales_by_frame ? sf : sb) * d.scale_stride : nullptr;
    c.in.opacities = d.opacities ? d.opacities + sb * d.opac_stride : nullptr;
    c.in.colors_precomp = d.colors ? d.colors + sf * d.color_stride : nullptr;
    c.in.shs = d.shs ? d.shs + sb * d.sh_stride : nullptr;
    c.in.cov3D_precomp = d.cov3D ? d.cov3D + sb * d.cov_stride : nullptr;
    c.radii = d.radii ? d.radii + sb * d.radii_stride : nullptr;
    const GeomLayout L = geom_layout(d.N, d.H, d.W);
    c.g = geom_ptrs(d.geom + sb * d.geom_stride, L);
    c.b = bin_ptrs(d.binning ? d.binning + sb * d.bin_stride : nullptr, d.cap);
    c.im = img_ptrs(d.image ? d.image + sb * d.img_stride : nullptr, d.H, d.W);
    c.cap = d.cap;
    c.trec = d.tile_records;
    c.colors = c.in.colors_precomp ? c.in.colors_precomp : c.g.rgb;
    c.out_color = d.out_color ? d.out_color + sb * d.C * P : nullptr;
    c.out_depth = d.out_depth ? d.out_depth + sb * P : nullptr;
    c.out_alpha = d.out_alpha ? d.out_alpha + sb * P : nullptr;
    c.dL_dcolor = d.dL_dcolor ? d.dL_dcolor + sb * d.C * P : nullptr;
    c.dL_ddepth = d.dL_ddepth ? d.dL_ddepth + sb * P : nullptr;
    c.dL_dalpha = d.dL_dalpha ? d.dL_dalpha + sb * P : nullptr;
    c.dLq = d.dLq ? d.dLq + sb * d.dlq_stride : nullptr;
    c.rec_cap = d.rec_cap;
    c.o.dL_dmeans2D = d.o.dL_dmeans2D ? d.o.dL_dmeans2D + sb * N * 3 : nullptr;
    c.o.dL_dmeans3D = d.o.dL_dmeans3D ? d.o.dL_dmeans3D + sb * N * 3 : nullptr;
    c.o.dL_dopacity = d.o.dL_dopacity ? d.o.dL_dopacity + sb * N : nullptr;
    c.o.dL_dcolors = d.o.dL_dcolors ? d.o.dL_dcolors + sb * N * d.C : nullptr;
    c.o.dL_dsh = d.o.dL_dsh ? d.o.dL_dsh + sb * N * d.sh_coeffs * 3 : nullptr;
    c.o.dL_dscales = d.o.dL_dscales ? d.o.dL_dscales + sb * N * 3 : nullptr;
    c.o.dL_drotations = d.o.dL_drotations ? d.o.dL_drotations + sb * N * 4 : nullptr;
    c.o.dL_dcov3D = d.o.dL_dcov3D ? d.o.dL_dcov3D + sb * N * 6 : nullptr;
    return c;
}

// division of labour of the tile sort's two variants (raster_bin.hip) -- the forward blend needs it too (raster_render.hip): the
// 256-thread variant sorts every tile of <= kSortSmallCap entries in LDS (and, fused, blends it), the 1024-thread variant the larger
// ones among the first kLargeRanks ranks of a view's longest-first tile order
constexpr int kSortSmallCap = 2048;
constexpr int kLargeRanks = 64;

// ---- launchers (defined in the .hip files); every launch covers the d.B views of the batch ----
int launch_preprocess(const BatchDesc &d, hipStream_t st);
int launch_colscan(const BatchDesc &d, hipStream_t st);
int launch_scatter(const BatchDesc &d, hipStream_t st);
int launch_tile_sort(const BatchDesc &d, hipStream_t st);
// per-device helper stream for launches that run beside the caller's stream (fork / join events); nullptr if it
// cannot be created (callers then launch on the caller's stream)
struct AuxStream { hipStream_t st, st2; hipEvent_t fork, join, fork2, join2; bool ok; bool pending2; };
AuxStream *aux_stream();
int launch_render_fwd(const BatchDesc &d, hipStream_t st);
int launch_sort_and_forward(const BatchDesc &d, hipStream_t st);      // K4 + K5 of a forward (every caller runs them back to back)
int launch_render_fwd_long(const BatchDesc &d, hipStream_t st);   // the long cells of the large tiles (earlylist)
int launch_render_bwd(const BatchDesc &d, hipStream_t st);
int launch_gather_bwd(const BatchDesc &d, hipStream_t st);
int launch_zero_counters(const BatchDesc &d, hipStream_t st);
int launch_n_contrib_tile_positions(void *geom, void *binning, void *image, int N, int H, int W, int64_t cap, uint32_t *out, hipStream_t st);
int set_trace_buffer(void *dev_ptr, uint32_t min_work);
int set_sort_trace_buffer(void *dev_ptr);
int launch_mark_visible(int N, const float *means3D, const float *view, uint8_t *present, hipStream_t st);

}  // namespace dm4d
