// raster_bin.hip -- binning kernels of the tile rasterizer (gfx950).
//
//   K2 k_colscan    : per-tile exclusive scan of the dense per-workgroup tile histogram written
//                     by K1 (-> per-(workgroup, tile) slot bases, tile counts); the tile segment
//                     starts (== upstream's `ranges`) follow from a scan K3 does in LDS.
//   K4 k_tile_sort  : one workgroup per tile; sorts the tile's duplicates by the 64-bit key
//                     (depth bits << 32 | Gaussian id) with a depth-bucket counting sort in LDS, then
//                     splits the sorted list into the sixteen 4x4-pixel cell lists.
//
// Why not upstream's global radix sort (cub::DeviceRadixSort over tile<<32|depth, 6 passes of
// 24 B/duplicate): duplicates are already partitioned by tile after K3, a tile's list is a few
// hundred entries and MI355X has 160 KB of LDS per CU, so each duplicate is read once (12 B) and
// written once (8 B).  Because (tile, depth bits, id) is a total order, the result is exactly
// the order a stable radix sort of tile<<32|depth produces from the Gaussian-major duplicate
// list -- the parity tests compare the (key, value) list bit-for-bit.
#include <stdlib.h>

#include "common.h"
#include "raster.h"

namespace dm4d {

// ---------------------------------------------------------------------------------------- K2
// hist[w][t] (duplicates of workgroup w in tile t) -> exclusive scan over w in place, and
// tile_count[t] = column total.  A column is only ~200 rows, but walked by one thread it is ~25 dependent memory
// round trips (the kernel was 13 us of pure latency).  So kColSegs threads share a column: thread (segment, tile)
// loads its rows in ONE round trip (consecutive threads = consecutive tiles of a row: coalesced), the segment
// totals meet in LDS, and every thread writes its rows back with its offset.
constexpr int kColTiles = 64, kColSegs = 16, kColRows = 16;   // tiles per workgroup, threads per column, rows held in registers
constexpr int kColThreads = kColTiles * kColSegs;
__global__ __launch_bounds__(kColThreads) void k_colscan(BatchDesc d)
{
    __shared__ uint32_t s_part[kColSegs][kColTiles];
    const ViewCtx c = resolve(d, blockIdx.y);
    const GeomPtrs &g = c.g;
    const int T = c.T, nb = (c.in.N + kPreBlock - 1) / kPreBlock;
    const int tl = threadIdx.x % kColTiles, seg = threadIdx.x / kColTiles;
    const int t = blockIdx.x * kColTiles + tl;
    const bool live = t < T;
    const int rps = (nb + kColSegs - 1) / kColSegs;            // rows per segment
    const int w0 = min(seg * rps, nb), w1 = min(w0 + rps, nb);
    uint32_t v[kColRows];
    uint32_t sum = 0;
    if (rps <= kColRows) {
#pragma unroll
        for (int k = 0; k < kColRows; ++k) {
            v[k] = (live && w0 + k < w1) ? g.hist[(size_t)(w0 + k) * T + t] : 0u;
            sum += v[k];
        }
    } else if (live) {
        for (int w = w0; w < w1; ++w) sum += g.hist[(size_t)w * T + t];
    }
    s_part[seg][tl] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kColSegs; ++k) {
        const uint32_t pk = s_part[k][tl];
        run += (k < seg) ? pk : 0u;
        total += pk;
    }
    if (!live) return;
    if (rps <= kColRows) {
#pragma unroll
        for (int k = 0; k < kColRows; ++k) {
            if (w0 + k < w1) g.hist[(size_t)(w0 + k) * T + t] = run;
            run += v[k];
        }
    } else {
        for (int w = w0; w < w1; ++w) {
            const uint32_t x = g.hist[(size_t)w * T + t];
            g.hist[(size_t)w * T + t] = run;
            run += x;
        }
    }
    if (seg == 0) g.tile_count[t] = total;
}

// ---------------------------------------------------------------------------------------- K4
// Per-tile sort by the 64-bit key (depth bits << 32 | Gaussian id) == the order a stable radix sort
// of tile<<32|depth produces from the Gaussian-major duplicate list.
//
// Fast path (n <= kSortLdsCap, everything in LDS): ADAPTIVE DEPTH-BUCKET COUNTING SORT.
//   1. min / max of the tile's depths                      (block reduction)
//   2. bucket = monotone quantisation of the depth into kBins buckets, LDS histogram, scan
//   3. scatter keys bucket-major (slot order inside a bucket is arbitrary)
//   4. rank inside the bucket by the full 64-bit key (buckets hold a handful of entries; in the
//      degenerate all-equal-depth case this degrades to an O(n^2/256) rank sort, still exact)
// The result is fully determined by the keys (a total order), so it is deterministic although the
// scatter uses LDS atomics.  O(n) work and 6 barriers instead of the O(n log^2 n) / 55-barrier
// bitonic network this replaced.
// Large tiles (n > kSortLdsCap): the same algorithm with the keys resident in HBM/L2 (the bucket-major
// copy borrows the tile's cell-0 list segment, which is only written afterwards).
// debug (dm4d_debug_trace): per-tile phase timestamps {start, loaded+binned, sorted, end} in 100 MHz ticks
__device__ uint64_t *g_sort_trace = nullptr;
__device__ int g_sort_trace_variant = 0;       // whose tiles are recorded: 0 the large (1024-thread) variant's, 1 the small one's
int set_sort_trace_buffer(void *dev_ptr)
{
    uint64_t *p = (uint64_t *)dev_ptr;
    const int variant = getenv("DM4D_SORT_TRACE_VARIANT") ? atoi(getenv("DM4D_SORT_TRACE_VARIANT")) : 0;
    DM4D_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_sort_trace), &p, sizeof(p)));
    DM4D_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_sort_trace_variant), &variant, sizeof(variant)));
    return DM4D_OK;
}
constexpr int kSortPerThread = 8;        // keys a thread holds in registers: LDS capacity = 8 x threads
constexpr int kSortSmall = 256, kSortLarge = 1024;   // threads of the two variants (tiles <= 2048 / <= 8192 entries)
static_assert(kSortPerThread * kSortSmall == kSortSmallCap, "raster.h: the small variant's LDS capacity");
// (kLargeRanks, raster.h: tiles per view -- the first ranks of K3's longest-first order -- the large variant covers)
constexpr int kBins = 1024;

// Split the sorted tile list into the sixteen cell lists (stable compaction by the Gaussian's cell block,
// cellinfo / cellmask of K1/K3).  A cell-list entry is ONE 32-bit word: the Gaussian id and, in the top bits, which of
// the Gaussian's backward records (they start at rec0[i], one per reached cell of its block) belongs to this cell
// (raster.h, BinPtrs::clist).  Round 1 wrote 12 bytes per entry -- id, tile-list position, record index: 230 MB per
// step on the bench scene; the position is not needed (the blend kernels count contributors in cell-list positions,
// the list being a subsequence of the tile list) and the record index is rec0 + a small rank.
//
// Chunks of 2048 entries, 8 CONSECUTIVE entries per thread: all gathers of a chunk are issued together
// (one memory round trip), the sixteen per-cell counts of a thread travel as four 64-bit words of
// 16-bit fields through ONE block scan, and each thread then walks its entries in order.  3 barriers per
// chunk (the first version compacted 256 entries at a time with 16 ballots: 3 barriers and two
// dependent gathers per 256 entries, 70 % of the kernel).
// ---- round 4: the split of a sorted tile list into its sixteen cell lists by BALLOT COMPACTION (finish_tile_lds) ----
// The first version (finish_tile below, still used by the tiles whose lists do not fit the LDS) gives a thread kFinE consecutive
// entries and lets it loop over the set bits of each entry's 16-cell mask: ~35 instructions per (entry, cell) pair in a loop
// whose trip count is the LARGEST popcount of the wave's 64 entries (2.3 cells per entry on average, ~7 at the maximum) -- the
// phase trace (tools/sort_trace.py) has it at 11 of the 17.6 us a tile's workgroup lives (62 % of the kernel).
// Now: phase A, a thread per entry: gather cellinfo / cellmask, compute the entry's 16-bit cell mask and, per row of the tile's
// 4 x 4 cell window, the record rank of the row's first reached cell (the ranks of the others follow from the mask: the cells a
// row reaches are consecutive) -- 8 bytes per entry into LDS.  Phase B, a wave per four cells: 64 entries at a time, `ballot` of
// "entry reaches cell k", position = cell base + the number of set bits below the lane: consecutive lanes write consecutive
// words of the cell list, no loop over bits, no divergence, no block scan.
constexpr uint32_t kCoarse = 256;              // level-1 depth buckets of the LDS sort (below)
// the sixteen-cell mask of a tile-list entry (bit k = the Gaussian's alpha >= 1/255 support reaches cell k of tile (tx, ty); cell
// ids as raster.h: 4 * quadrant + (cx & 1) + 2 * (cy & 1)) and, in `rowrank`, for each window row cy the record rank of its first
// reached cell (8 bits each, saturated at kRankBig: ranks beyond are recomputed from cellinfo by the blend backward)
__device__ __forceinline__ uint32_t tile_cell_mask(const uint4 ci, const uint64_t cm, const int tx, const int ty, uint32_t &rowrank)
{
    const int bx0 = (int)(ci.x & 0xFFFFu), by0 = (int)(ci.x >> 16);
    const int nbx = (int)(ci.y & 0xFFFFu), nby = (int)(ci.y >> 16);
    const int ox = 4 * tx - bx0, oy = 4 * ty - by0;
    const bool dense = ci.w != 0u;
    const int cx0 = max(bx0, 4 * tx) - 4 * tx, cx1 = min(bx0 + nbx, 4 * tx + 4) - 4 * tx;
    const int cy0 = max(by0, 4 * ty) - 4 * ty, cy1 = min(by0 + nby, 4 * ty + 4) - 4 * ty;
    const uint32_t xmask = cx1 > cx0 ? ((1u << (cx1 - cx0)) - 1u) << cx0 : 0u;   // (the block may miss the tile)
    uint32_t mm = 0u;
    rowrank = 0u;
#pragma unroll
    for (int cy = 0; cy < 4; ++cy) {
        const int sh = (oy + cy) * nbx + ox;         // bit of the block mask that is cell (cx = 0, cy) of the window
        uint32_t nib = dense ? xmask : (uint32_t)(sh >= 0 ? (cm >> (sh & 63)) : (cm << ((-sh) & 63))) & xmask;
        nib = (cy >= cy0 && cy < cy1) ? nib : 0u;
        mm |= ((nib & 3u) | ((nib & 12u) << 2)) << (8 * (cy >> 1) + 2 * (cy & 1));
        // rank of the row's first reached cell: its index in a dense block, else the number of reached cells before it
        const int first = sh + (nib ? __builtin_ctz(nib) : 0);
        const uint32_t rk = dense ? (uint32_t)max(first, 0) : (uint32_t)__popcll(cm & ((1ull << (first & 63)) - 1ull));
        rowrank |= min(rk, kRankBig) << (8 * cy);
    }
    return mm;
}
// ids[0 .. n): the tile's sorted Gaussian ids (LDS); rec[0 .. n): 8-byte scratch per entry (LDS); kSortThreads / 64 waves
template <int kSortThreads>
__device__ __forceinline__ void finish_tile_lds(const ViewCtx &c, const int tile, const uint32_t s, const uint32_t n,
                                                const uint32_t *ids, uint2 *rec)
{
    constexpr int kWaves = kSortThreads / 64;
    static_assert(kCells % kWaves == 0 || kWaves > kCells, "cells are dealt to the waves");
    const GeomPtrs &g = c.g;
    const BinPtrs &b = c.b;
    const uint32_t cap = c.cap;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tx = tile % c.vp.gx, ty = tile / c.vp.gx;
    // ---- phase A: a thread per entry ----
    for (uint32_t e = tid; e < n; e += kSortThreads) {
        const uint32_t gid = ids[e];
        const uint4 ci = g.cellinfo[gid];
        const uint64_t cm = g.cellmask[gid];      // only meaningful when the block is not dense
        uint32_t rr;
        const uint32_t mm = tile_cell_mask(ci, cm, tx, ty, rr);
        rec[e] = make_uint2(mm, rr);
        b.point_list[s + e] = gid;
    }
    __syncthreads();
    // ---- phase B: wave wv takes the cells k = wv, wv + kWaves, ...; 64 entries at a time, every one of the wave's cells per step
    //      (one LDS read of the entry's record serves them all) ----
    constexpr int kPer = (kCells + kWaves - 1) / kWaves;      // cells per wave: 4 (256 threads), 1 (1024 threads)
    uint32_t base[kPer], below[kPer];
    int ksh[kPer], cysh[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const int k = min(wv + kWaves * i, kCells - 1);
        const int qd = k >> 2, r4 = k & 3;
        const int cx = 2 * (qd & 1) + (r4 & 1), cy = 2 * (qd >> 1) + (r4 >> 1);      // the cell's column / row in the tile's window
        // bits of the mask that are the cells of row cy LEFT of column cx (the mask is in cell-id order: columns 0..3 of row cy sit at
        // bit 8 (cy >> 1) + 2 (cy & 1) + {0, 1, 4, 5})
        const uint32_t colbits = cx == 0 ? 0u : cx == 1 ? 1u : cx == 2 ? 3u : 0x13u;
        below[i] = colbits << (8 * (cy >> 1) + 2 * (cy & 1));
        base[i] = 0u;
        ksh[i] = k;
        cysh[i] = 8 * cy;
    }
    for (uint32_t e0 = 0; e0 < n; e0 += 64u) {
        const uint32_t e = e0 + (uint32_t)lane;
        const uint2 r = e < n ? rec[e] : make_uint2(0u, 0u);
        const uint32_t id = e < n ? ids[e] : 0u;
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            if (wv + kWaves * i >= kCells) continue;
            const bool has = (r.x >> ksh[i]) & 1u;
            const uint64_t bal = __builtin_amdgcn_ballot_w64(has);
            if (has) {
                const uint32_t pos = base[i] + mbcnt(bal);
                if (s + pos < cap) {
                    const uint32_t rank = min(((r.y >> cysh[i]) & 0xFFu) + (uint32_t)__builtin_popcount(r.x & below[i]), kRankBig);
                    b.clist[(size_t)ksh[i] * b.cap + s + pos] = id | (rank << kGidBits);
                    if (c.trec) b.cpos[(size_t)ksh[i] * b.cap + s + pos] = (uint16_t)min(e, 0xFFFFu);
                }
            }
            base[i] += (uint32_t)__builtin_popcountll(bal);
        }
    }
    if (lane == 0) {
        // long cells are blended by their own kernels (raster.h, kLongCell); which slot a cell gets does not matter.
        // The backward takes all of them (longlist); the forward only those of the tiles the LARGE variant sorts: it
        // finishes long before the small variant, so their forward starts that much earlier.
        // ONE returning atomic per wave and list for the wave's cells together (round 4, second half: one per cell was up to four
        // DEPENDENT round trips to one L2 address at the end of every workgroup's life -- the small variant 91 -> 75 us when fewer cells
        // took them, tools/prof_two_libs.sh)
        uint32_t nw = 0, ne = 0;
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int k = wv + kWaves * i;
            if (k >= kCells) continue;
            nw += base[i] >= kWideBwd ? 1u : 0u;
            ne += (base[i] >= kLongCell && kSortThreads == 1024) ? 1u : 0u;
        }
        uint32_t sw = nw ? atomicAdd(&g.counters[kCntLong], nw) : 0u;
        uint32_t se = ne ? atomicAdd(&g.counters[kCntLongEarly], ne) : 0u;
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int k = wv + kWaves * i;
            if (k >= kCells) continue;
            g.ccount[tile * kCells + k] = base[i];
            const bool early = base[i] >= kLongCell && kSortThreads == 1024;
            if (base[i] >= kWideBwd) g.longlist[sw++] = (uint32_t)(tile * kCells + k);
            if (early) g.earlylist[se++] = (uint32_t)(tile * kCells + k);
            g.cflag[tile * kCells + k] = early ? 1u : 0u;
        }
    }
    if (c.trec && n > 0x10000u && tid == 0) g.counters[kCntRecOverflow] = 1u;   // cpos holds 16-bit tile-list positions
}
__device__ __forceinline__ uint32_t depth_bin(const uint32_t dz, const float zmin, const float scale, const int bins)
{
    return min((uint32_t)(bins - 1), (uint32_t)max(0, f2i_sat((__uint_as_float(dz) - zmin) * scale)));
}

constexpr int kFinE = 2;      // (round 4: only the HBM path -- tiles beyond the LDS capacity -- still takes this routine; 8 entries per thread cost the kernel 60 VGPRs)
__device__ __forceinline__ uint64_t spread4(uint32_t x)   // bit i of x (< 16) -> 16-bit field i
{
    // four copies of x at bit offsets 0, 15, 30, 45: bit i of copy i sits at 16 i
    return ((uint64_t)x * 0x0000200040008001ull) & 0x0001000100010001ull;
}
// four 16-bit fields that never overflow into each other: the two halves are scanned as independent 32-bit words
__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v, int lane)
{
    const uint32_t lo = wave_incl_scan_u32((uint32_t)v, lane), hi = wave_incl_scan_u32((uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}
template <int kSortThreads, typename GidAt>
__device__ __forceinline__ void finish_tile(const ViewCtx &c, int tile, uint32_t s, uint32_t n, GidAt &&gid_at)
{
    __shared__ uint32_t s_cbase[kCells];
    __shared__ uint64_t s_ws[kSortThreads / 64][4];
    const GeomPtrs &g = c.g;
    const BinPtrs &b = c.b;
    const uint32_t cap = c.cap;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tx = tile % c.vp.gx, ty = tile / c.vp.gx;
    if (tid < kCells) s_cbase[tid] = 0u;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < n; c0 += kSortThreads * kFinE) {
        // consecutive entries per thread in this chunk: as few as cover it, so a short list still uses every thread
        const uint32_t E = min((uint32_t)kFinE, (n - c0 + kSortThreads - 1) / kSortThreads);
        const uint32_t e0 = c0 + (uint32_t)tid * E;
        const uint32_t e1 = min(n, e0 + E);      // this thread's entries: [e0, e1)
        uint32_t gid[kFinE], m[kFinE];
        uint4 ci[kFinE];
        uint64_t cm[kFinE];
#pragma unroll
        for (int j = 0; j < kFinE; ++j) {
            m[j] = 0u;
            if (e0 + j < e1) {
                gid[j] = gid_at(e0 + j);
                ci[j] = g.cellinfo[gid[j]];
                cm[j] = g.cellmask[gid[j]];   // only meaningful when the block is not dense
            }
        }
        uint64_t cnt[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
        for (int j = 0; j < kFinE; ++j) {
            if (e0 + j < e1) {
                b.point_list[s + e0 + j] = gid[j];
                const int bx0 = (int)(ci[j].x & 0xFFFFu), by0 = (int)(ci[j].x >> 16);
                const int nbx = (int)(ci[j].y & 0xFFFFu), nby = (int)(ci[j].y >> 16);
                const int ox = 4 * tx - bx0, oy = 4 * ty - by0;
                const bool dense = ci[j].w != 0u;
                const int cx0 = max(bx0, 4 * tx) - 4 * tx, cx1 = min(bx0 + nbx, 4 * tx + 4) - 4 * tx;
                const int cy0 = max(by0, 4 * ty) - 4 * ty, cy1 = min(by0 + nby, 4 * ty + 4) - 4 * ty;
                // the tile's 4 x 4 window of the Gaussian's cell block, a row (nibble) at a time: bit cx of row cy is
                // bit (oy + cy) * nbx + ox + cx of the block mask; cells cx = 0..3 of row cy have the ids
                // 8 (cy >> 1) + 2 (cy & 1) + {0, 1, 4, 5}
                const uint32_t xmask = cx1 > cx0 ? ((1u << (cx1 - cx0)) - 1u) << cx0 : 0u;   // (the block may miss the tile)
                uint32_t mm = 0u;
#pragma unroll
                for (int cy = 0; cy < 4; ++cy) {
                    const int sh = (oy + cy) * nbx + ox;
                    uint32_t nib = dense ? xmask : (uint32_t)(sh >= 0 ? (cm[j] >> (sh & 63)) : (cm[j] << ((-sh) & 63))) & xmask;
                    nib = (cy >= cy0 && cy < cy1) ? nib : 0u;
                    mm |= ((nib & 3u) | ((nib & 12u) << 2)) << (8 * (cy >> 1) + 2 * (cy & 1));
                }
                m[j] = mm;
#pragma unroll
                for (int w = 0; w < 4; ++w) cnt[w] += spread4((mm >> (4 * w)) & 15u);
            }
        }
        uint64_t run[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint64_t incl = wave_incl_scan_u64(cnt[w], lane);
            if (lane == 63) s_ws[wv][w] = incl;
            run[w] = incl - cnt[w];
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; ++w)
            for (int ww = 0; ww < wv; ++ww) run[w] += s_ws[ww][w];
#pragma unroll
        for (int j = 0; j < kFinE; ++j) {
            uint32_t mm = m[j];
            if (mm) {
                const int bx0 = (int)(ci[j].x & 0xFFFFu), by0 = (int)(ci[j].x >> 16);
                const int nbx = (int)(ci[j].y & 0xFFFFu);
                const int ox = 4 * tx - bx0, oy = 4 * ty - by0;
                const bool dense = ci[j].w != 0u;
                while (mm) {
                    const int k = __builtin_ctz(mm);
                    mm &= mm - 1u;
                    const uint64_t rw = (k < 8) ? ((k < 4) ? run[0] : run[1]) : ((k < 12) ? run[2] : run[3]);
                    const uint32_t pos = s_cbase[k] + (uint32_t)((rw >> (16 * (k & 3))) & 0xFFFFull);
                    if (s + pos < cap) {
                        const int qd = k >> 2, rr = k & 3;
                        const int cx = 2 * (qd & 1) + (rr & 1), cy = 2 * (qd >> 1) + (rr >> 1);
                        const int bit = (oy + cy) * nbx + ox + cx;
                        // the entry's backward record is rec0[gid] + rank: the cell's index in a dense block, or its rank
                        // among the cells really reached (< 64); kRankBig: a dense block too large for the field
                        const uint32_t rank = dense ? min((uint32_t)bit, kRankBig) : (uint32_t)__popcll(cm[j] & ((1ull << bit) - 1ull));
                        b.clist[(size_t)k * b.cap + s + pos] = gid[j] | (rank << kGidBits);
                        if (c.trec) b.cpos[(size_t)k * b.cap + s + pos] = (uint16_t)min(e0 + (uint32_t)j, 0xFFFFu);
                    }
                }
#pragma unroll
                for (int w = 0; w < 4; ++w) run[w] += spread4((m[j] >> (4 * w)) & 15u);
            }
        }
        __syncthreads();
        if (tid < kCells) {
            uint32_t add = 0;
#pragma unroll
            for (int w = 0; w < kSortThreads / 64; ++w) add += (uint32_t)((s_ws[w][tid >> 2] >> (16 * (tid & 3))) & 0xFFFFull);
            s_cbase[tid] += add;
        }
        __syncthreads();
    }
    if (c.trec && n > 0x10000u && tid == 0) g.counters[kCntRecOverflow] = 1u;   // cpos holds 16-bit tile-list positions
    if (tid < kCells) {
        g.ccount[tile * kCells + tid] = s_cbase[tid];
        // long cells are blended by their own kernels (raster.h, kLongCell); which slot a cell gets does not matter.
        // The backward takes all of them (longlist); the forward only those of the tiles THIS, the large, variant
        // sorts: it finishes long before the small variant, so their forward starts that much earlier.
        const bool is_long = s_cbase[tid] >= kLongCell, early = is_long && kSortThreads == kSortLarge;
        const bool wide = s_cbase[tid] >= kWideBwd;
        // (one returning atomic per list for the tile's sixteen cells: the lanes of this branch are lanes 0..15 of wave 0)
        const uint64_t bw = __builtin_amdgcn_ballot_w64(wide), be = __builtin_amdgcn_ballot_w64(early);
        uint32_t sw = 0, se = 0;
        if (tid == 0 && bw) sw = atomicAdd(&g.counters[kCntLong], (uint32_t)__builtin_popcountll(bw));
        if (tid == 0 && be) se = atomicAdd(&g.counters[kCntLongEarly], (uint32_t)__builtin_popcountll(be));
        sw = (uint32_t)__builtin_amdgcn_readfirstlane((int)sw);
        se = (uint32_t)__builtin_amdgcn_readfirstlane((int)se);
        if (wide) g.longlist[sw + mbcnt(bw)] = (uint32_t)(tile * kCells + tid);
        if (early) g.earlylist[se + mbcnt(be)] = (uint32_t)(tile * kCells + tid);
        g.cflag[tile * kCells + tid] = early ? 1u : 0u;
    }
}

// (Round 5, measured and removed: the 256-thread variant going straight on to the FORWARD BLEND of its tile -- wave w = quadrant w, the
// code of k_render_fwd (raster_fwd.h), in the LDS the sort no longer needs.  Bit-identical images, one launch and one cross-stream
// hand-over less -- and the fused kernel takes exactly as long as sort + hand-over + forward (255 us against 71 + 12 + 170,
// profiles/r05_timeline_fused_sort_forward.txt): both halves are VALU-bound (74 % / 96 % busy), so there is nothing for them to hide in
// each other, a four-wave workgroup holds its slots until its slowest quadrant is done, and the large tiles' quadrants, launched on
// their own behind the large variant, delay the long cells' kernel: the backward starts 28 us LATER.)
template <int kSortThreads>
__device__ __forceinline__ void tile_sort_block(const BatchDesc &d, const uint32_t blk)
{
    constexpr int kSortLdsCap = kSortPerThread * kSortThreads;
    constexpr int kWaves = kSortThreads / 64;
    constexpr bool kIsLarge = kSortThreads == kSortLarge;
    __shared__ uint64_t s_b[kSortLdsCap];          // bucket-major keys; afterwards the 8-byte per-entry records of finish_tile_lds
    __shared__ uint32_t s_fine[kSortLdsCap + 1];   // fine histogram -> fine bucket starts / cursors -> ends; afterwards the sorted ids
    __shared__ uint32_t s_cstart[kCoarse + 1];     // coarse histogram -> coarse bucket ends
    __shared__ uint32_t s_red[2 * kWaves];
    uint32_t *s_bin = s_fine, *s_cur = s_fine + kBins + 1;      // HBM path (tiles beyond the LDS capacity): ONE level of linear buckets, as rounds 1-3
    static_assert(kSortLdsCap + 1 >= 2 * kBins + 1, "the HBM path's histogram and cursors borrow s_fine");
    // block -> (view, tile) in the launch order of K3: the r-th longest tile of every view, views interleaved
    const int view = (int)(blk % (uint32_t)d.B);
    const uint32_t rank = blk / (uint32_t)d.B;
    const ViewCtx c = resolve(d, view);
    const GeomPtrs &g = c.g;
    const BinPtrs &b = c.b;
    const uint32_t cap = c.cap;
    if (rank >= (uint32_t)c.T) return;
    const int t = (int)g.order[rank];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint64_t *tr = (g_sort_trace && (kIsLarge ? g_sort_trace_variant == 0 : g_sort_trace_variant == 1)) ? g_sort_trace + 5 * (size_t)blk : nullptr;
    if (tr && tid == 0) tr[0] = wall_clock64();
    const uint32_t s = g.tile_start[t];
    uint32_t n = g.tile_count[t];
    if (s >= cap) n = 0;
    else if (s + n > cap) n = cap - s;   // overflow: memory-safe, result flagged invalid by K3
    // Division of labour: the 256-thread variant takes every tile of <= 2048 entries; the 1024-thread variant
    // (launched over the first kLargeRanks ranks of every view, on a second stream) takes the larger ones.
    if (kIsLarge ? (n <= (uint32_t)(kSortPerThread * kSortSmall)) : (n > (uint32_t)kSortLdsCap && rank < (uint32_t)kLargeRanks)) return;
    if (n == 0) {
        if (tid < kCells) { g.ccount[t * kCells + tid] = 0u; g.cflag[t * kCells + tid] = 0u; }
        return;
    }
    if (n <= (uint32_t)kSortLdsCap) {
        // TWO-LEVEL, histogram-equalised depth buckets (round 4).  A tile of a closed surface sees a front and a back layer, each a few
        // percent of the tile's depth range: with ONE linear map depth -> 1024 buckets over [zmin, zmax] (rounds 1-3; still the HBM path
        // below) most entries share a few dozen buckets, and the rank step is linear in the bucket size per entry (measured: the 695
        // tiles of 1025 .. 2048 entries of the bench scene took 84 us by themselves, 46 us now; the large variant 45 -> 34 us).
        // Level 1: kCoarse linear buckets, histogram + scan.  Level 2: coarse bucket c is split LINEARLY into as many fine buckets as it
        // holds entries, so the fine buckets live in the index space of the output positions (fine = start[c] + floor(frac * count[c])),
        // a second histogram + scan over n counters gives their starts, and a fine bucket holds ~1 entry wherever the depths are locally
        // smooth.  Both maps are monotone in the depth: bucket order = depth order, the rank step resolves the rest by the full key.
        // ---- load, min / max depth ----
        uint64_t key[kSortPerThread];
        uint32_t dmin = 0xFFFFFFFFu, dmax = 0u;
#pragma unroll
        for (int r = 0; r < kSortPerThread; ++r) {
            const uint32_t e = r * kSortThreads + tid;
            key[r] = ~0ull;
            if (e < n) {
                const uint32_t dz = b.u_depth[s + e];
                key[r] = ((uint64_t)dz << 32) | b.u_idx[s + e];
                dmin = min(dmin, dz);
                dmax = max(dmax, dz);
            }
        }
        dmin = wave_min_u32(dmin);
        dmax = wave_max_u32(dmax);
        if (lane == 0) { s_red[wv] = dmin; s_red[kWaves + wv] = dmax; }
        for (uint32_t i = tid; i <= kCoarse; i += kSortThreads) s_cstart[i] = 0u;
        for (uint32_t i = tid; i <= n; i += kSortThreads) s_fine[i] = 0u;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { dmin = min(dmin, s_red[w]); dmax = max(dmax, s_red[kWaves + w]); }
        // depths are positive floats (> 0.2), so the bit patterns order like the values
        const float zmin = __uint_as_float(dmin), zspan = __uint_as_float(dmax) - zmin;
        const float scale = (zspan > 0.f && zspan < 3.0e38f) ? (float)kCoarse / zspan : 0.f;
        // ---- level 1: coarse histogram ----
#pragma unroll
        for (int r = 0; r < kSortPerThread; ++r)
            if ((uint32_t)(r * kSortThreads + tid) < n) atomicAdd(&s_cstart[depth_bin((uint32_t)(key[r] >> 32), zmin, scale, kCoarse) + 1], 1u);
        __syncthreads();
        if (tr && tid == 0) tr[1] = wall_clock64();
        // inclusive scan by the first wave: cstart[1 + c] = end of coarse bucket c = start of c + 1, cstart[0] = 0
        if (tid < 64) {
            constexpr int per = kCoarse / 64;
            uint32_t loc[per], sum = 0;
#pragma unroll
            for (int i = 0; i < per; ++i) { loc[i] = s_cstart[1 + tid * per + i]; sum += loc[i]; }
            uint32_t run = wave_incl_scan_u32(sum, lane) - sum;
#pragma unroll
            for (int i = 0; i < per; ++i) {
                run += loc[i];
                s_cstart[1 + tid * per + i] = run;
            }
        }
        __syncthreads();
        // ---- level 2: fine bucket = start of the coarse bucket + linear position inside it, in units of (range / count) ----
        uint32_t fb[kSortPerThread];
#pragma unroll
        for (int r = 0; r < kSortPerThread; ++r) {
            fb[r] = 0u;
            if ((uint32_t)(r * kSortThreads + tid) < n) {
                const float q = fmaxf(0.f, (__uint_as_float((uint32_t)(key[r] >> 32)) - zmin) * scale);      // depth in coarse-bucket units
                const uint32_t cb = min((uint32_t)(kCoarse - 1), (uint32_t)max(0, f2i_sat(q)));              // == depth_bin(...)
                const uint32_t lo = s_cstart[cb], cnt = s_cstart[cb + 1] - lo;
                const uint32_t sub = min(cnt - 1u, (uint32_t)max(0, f2i_sat((q - (float)cb) * (float)cnt)));
                fb[r] = lo + sub;
                atomicAdd(&s_fine[fb[r] + 1], 1u);
            }
        }
        __syncthreads();
        // exclusive scan of the n fine counters in place (<= kSortPerThread consecutive counters per thread): fine[1 + f] = start of f
        {
            const uint32_t per = (n + kSortThreads - 1) / kSortThreads;
            uint32_t loc[kSortPerThread], sum = 0;
#pragma unroll
            for (int i = 0; i < kSortPerThread; ++i) {
                const uint32_t f = (uint32_t)tid * per + (uint32_t)i;
                loc[i] = ((uint32_t)i < per && f < n) ? s_fine[1 + f] : 0u;
                sum += loc[i];
            }
            const uint32_t incl = wave_incl_scan_u32(sum, lane);
            if (lane == 63) s_red[wv] = incl;
            __syncthreads();
            uint32_t run = incl - sum;
            for (int w = 0; w < wv; ++w) run += s_red[w];
#pragma unroll
            for (int i = 0; i < kSortPerThread; ++i) {
                const uint32_t f = (uint32_t)tid * per + (uint32_t)i;
                if ((uint32_t)i < per && f < n) s_fine[1 + f] = run;
                run += loc[i];
            }
        }
        __syncthreads();
        // ---- bucket-major scatter: cursor of fine bucket f = fine[1 + f] (its END afterwards; its start is fine[f], fine[0] = 0) ----
#pragma unroll
        for (int r = 0; r < kSortPerThread; ++r)
            if ((uint32_t)(r * kSortThreads + tid) < n) s_b[atomicAdd(&s_fine[1 + fb[r]], 1u)] = key[r];
        __syncthreads();
        // ---- rank inside the fine bucket by the full key ----
        uint32_t pos[kSortPerThread];
#pragma unroll
        for (int r = 0; r < kSortPerThread; ++r) {
            pos[r] = 0xFFFFFFFFu;
            if ((uint32_t)(r * kSortThreads + tid) < n) {
                const uint32_t lo = s_fine[fb[r]], hi = s_fine[fb[r] + 1];
                uint32_t rank = 0;
                for (uint32_t j = lo; j < hi; ++j) rank += (s_b[j] < key[r]) ? 1u : 0u;
                pos[r] = lo + rank;
            }
        }
        __syncthreads();      // every read of the cursors and of the bucket-major keys is done: their memory is reused
        uint32_t *s_ids = s_fine;                                   // the tile's sorted Gaussian ids (all the later stages need)
#pragma unroll
        for (int r = 0; r < kSortPerThread; ++r)
            if (pos[r] != 0xFFFFFFFFu) s_ids[pos[r]] = (uint32_t)key[r];
        __syncthreads();
        if (tr && tid == 0) tr[2] = wall_clock64();
        finish_tile_lds<kSortThreads>(c, t, s, n, s_ids, reinterpret_cast<uint2 *>(s_b));
        if (tr && tid == 0) { tr[3] = wall_clock64(); tr[4] = n; }
    } else {
        // ---- large tile: the same bucket sort with the keys resident in HBM (L2) ----
        // temp (bucket-major keys, as two 32-bit halves) lives in the tile's own cell-0 and cell-1 list segments, which
        // finish_tile() only writes afterwards; the sorted keys go back in place.
        uint32_t *kd = b.u_depth + s, *ki_ = b.u_idx + s;
        uint32_t *tmp_hi = b.clist + s, *tmp_lo = b.clist + b.cap + s;      // (cells 0 and 1: n words each)
        uint32_t dmin = 0xFFFFFFFFu, dmax = 0u;
        for (uint32_t e = tid; e < n; e += kSortThreads) {
            const uint32_t dz = kd[e];
            dmin = min(dmin, dz);
            dmax = max(dmax, dz);
        }
        dmin = wave_min_u32(dmin);
        dmax = wave_max_u32(dmax);
        if (lane == 0) { s_red[wv] = dmin; s_red[kWaves + wv] = dmax; }
        for (int i = tid; i <= kBins; i += kSortThreads) s_bin[i] = 0u;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { dmin = min(dmin, s_red[w]); dmax = max(dmax, s_red[kWaves + w]); }
        const float zmin = __uint_as_float(dmin), zspan = __uint_as_float(dmax) - zmin;
        const float scale = (zspan > 0.f && zspan < 3.0e38f) ? (float)kBins / zspan : 0.f;
        auto bin_of = [&](uint32_t dz) {
            return min((uint32_t)(kBins - 1), (uint32_t)max(0, f2i_sat((__uint_as_float(dz) - zmin) * scale)));
        };
        for (uint32_t e = tid; e < n; e += kSortThreads) atomicAdd(&s_bin[bin_of(kd[e]) + 1], 1u);
        __syncthreads();
        {
            constexpr int per = kBins / kSortThreads;
            uint32_t loc[per], sum = 0;
#pragma unroll
            for (int i = 0; i < per; ++i) { loc[i] = s_bin[1 + tid * per + i]; sum += loc[i]; }
            const uint32_t incl = wave_incl_scan_u32(sum, lane);
            __syncthreads();
            if (lane == 63) s_red[wv] = incl;
            __syncthreads();
            uint32_t run = incl - sum;
            for (int w = 0; w < wv; ++w) run += s_red[w];
#pragma unroll
            for (int i = 0; i < per; ++i) {
                s_cur[tid * per + i] = run;
                run += loc[i];
                s_bin[1 + tid * per + i] = run;
            }
        }
        __syncthreads();
        for (uint32_t e = tid; e < n; e += kSortThreads) {
            const uint32_t dz = kd[e];
            const uint32_t at = atomicAdd(&s_cur[bin_of(dz)], 1u);
            tmp_hi[at] = dz;
            tmp_lo[at] = ki_[e];
        }
        __syncthreads();   // workgroup-scope: tmp writes visible to the whole workgroup
        for (uint32_t e = tid; e < n; e += kSortThreads) {
            const uint64_t k = ((uint64_t)tmp_hi[e] << 32) | tmp_lo[e];
            const uint32_t bn = bin_of((uint32_t)(k >> 32));
            const uint32_t lo = s_bin[bn], hi = s_bin[bn + 1];
            uint32_t rank = 0;
            for (uint32_t j = lo; j < hi; ++j) rank += ((((uint64_t)tmp_hi[j] << 32) | tmp_lo[j]) < k) ? 1u : 0u;
            kd[lo + rank] = (uint32_t)(k >> 32);
            ki_[lo + rank] = (uint32_t)k;
        }
        __syncthreads();
        if (tr && tid == 0) { tr[1] = tr[0]; tr[2] = wall_clock64(); }
        finish_tile<kSortThreads>(c, t, s, n, [&](uint32_t e) { return ki_[e]; });
        if (tr && tid == 0) { tr[3] = wall_clock64(); tr[4] = n; }
    }
}

// The kernel: workgroup -> blocks `blockIdx.x + k gridDim.x` of the (rank, view) order.  The small variant is launched with one
// workgroup per block; the LARGE variant with at most one workgroup per CU (round 6): its workgroups need a whole CU each (1024
// threads, 104 KB of LDS), and every one beyond the first round used to wait for a CU to drain of the small variant's workgroups that
// run beside it -- with 20 views per step (64 ranks x 20 = 1280 workgroups, ~220 of them with a tile to sort) the launch took 200 us
// against 33 us at 8 views, and the long cells' forward behind it ended after the regular forward.  The blocks of the first round are
// ranks 0 .. 256 / B - 1 of every view: the longest tiles, i.e. the ones this variant exists for; a block without such a tile costs its
// workgroup two dependent loads.
template <int kSortThreads>
__global__ __launch_bounds__(kSortThreads, kSortThreads == 256 ? 6 : 4) void k_tile_sort(BatchDesc d, const uint32_t n_blocks)
{
    for (uint32_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        tile_sort_block<kSortThreads>(d, blk);
        if (blk + gridDim.x < n_blocks) __syncthreads();      // the next block reuses the LDS
    }
}

int launch_colscan(const BatchDesc &d, hipStream_t st)
{
    const int T = ((d.W + kTile - 1) / kTile) * ((d.H + kTile - 1) / kTile);
    if (T <= 0) return DM4D_OK;
    ProfScope prof_(kKColscan, st);
    hipLaunchKernelGGL(k_colscan, dim3((T + kColTiles - 1) / kColTiles, d.B), dim3(kColThreads), 0, st, d);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

// The few tiles of > 2048 entries (silhouettes) would bound K4 from below on the 256-thread variant (their
// keys do not fit its LDS): they run on the 1024-thread / 104 KB variant, concurrently on a helper stream.
static AuxStream g_aux[64] = {};
AuxStream *aux_stream()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    AuxStream &a = g_aux[dev];
    if (!a.ok) {
        if (hipStreamCreateWithFlags(&a.st, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipStreamCreateWithFlags(&a.st2, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&a.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&a.join, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&a.fork2, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&a.join2, hipEventDisableTiming) != hipSuccess) return nullptr;
        a.pending2 = false;
        a.ok = true;
    }
    return &a;
}

// workgroups of the large variant's launch: one per CU (DM4D_SORT_LARGE_GRID overrides: the A/B switch; 0 = one per block, as rounds 1-5)
static unsigned large_grid_cap()
{
    static const unsigned cap = [] {
        if (const char *e = getenv("DM4D_SORT_LARGE_GRID")) { const long v = atol(e); return v <= 0 ? 0xFFFFFFFFu : (unsigned)v; }
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return (unsigned)max(cus, 1);
    }();
    return cap;
}

int launch_tile_sort(const BatchDesc &d, hipStream_t st)
{
    const int T = ((d.W + kTile - 1) / kTile) * ((d.H + kTile - 1) / kTile);
    if (T <= 0) return DM4D_OK;
    AuxStream *a = aux_stream();
    const unsigned large_blocks = (unsigned)min(T, kLargeRanks) * (unsigned)d.B;
    ProfScope prof_(kKTileSort, st);
    // The large variant goes FIRST on the caller's stream (it needs whole CUs: 1024 threads + 104 KB of LDS per
    // workgroup; started after the small variant has filled the machine it waits for CUs to drain -- measured: its
    // workgroups then started 90 us late); the small variant runs beside it on the helper stream.
    hipStream_t sst = a ? a->st : st;
    if (a) {
        DM4D_HIP_CHECK(hipEventRecord(a->fork, st));
        DM4D_HIP_CHECK(hipStreamWaitEvent(a->st, a->fork, 0));
    }
    hipLaunchKernelGGL(k_tile_sort<kSortLarge>, dim3(min(large_blocks, large_grid_cap())), dim3(kSortLarge), 0, st, d, large_blocks);
    DM4D_HIP_CHECK(hipGetLastError());
    if (a) {
        // the forward of the large tiles' long cells starts as soon as THEIR sort is done, on a second helper stream,
        // beside the small variant and later the regular forward kernel (launch_render_fwd joins it)
        DM4D_HIP_CHECK(hipEventRecord(a->fork2, st));
        DM4D_HIP_CHECK(hipStreamWaitEvent(a->st2, a->fork2, 0));
        int rc2 = launch_render_fwd_long(d, a->st2);
        if (rc2) return rc2;
        DM4D_HIP_CHECK(hipEventRecord(a->join2, a->st2));
        a->pending2 = true;
    }
    hipLaunchKernelGGL(k_tile_sort<kSortSmall>, dim3((unsigned)T * (unsigned)d.B), dim3(kSortSmall), 0, sst, d, (unsigned)T * (unsigned)d.B);
    DM4D_HIP_CHECK(hipGetLastError());
    if (a) {
        DM4D_HIP_CHECK(hipEventRecord(a->join, a->st));
        DM4D_HIP_CHECK(hipStreamWaitEvent(st, a->join, 0));
    }
    return DM4D_OK;
}

// K4 + K5 of a forward (every caller runs them back to back)
int launch_sort_and_forward(const BatchDesc &d, hipStream_t st)
{
    int rc = launch_tile_sort(d, st);
    if (rc) return rc;
    return launch_render_fwd(d, st);
}

}  // namespace dm4d
