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
#include "common.h"
#include "raster.h"

namespace dm4d {

// ---------------------------------------------------------------------------------------- K2
// hist[w][t] (duplicates of workgroup w in tile t) -> exclusive scan over w in place, and
// tile_count[t] = column total.  One thread per tile; consecutive threads read consecutive
// tiles of one histogram row, so every step is a coalesced row access; 8 rows in flight.
constexpr int kColThreads = 64;
__global__ __launch_bounds__(kColThreads) void k_colscan(BatchDesc d)
{
    const ViewCtx c = resolve(d, blockIdx.y);
    const GeomPtrs &g = c.g;
    const int T = c.T, nb = (c.in.N + kPreBlock - 1) / kPreBlock;
    const int t = blockIdx.x * kColThreads + threadIdx.x;
    if (t >= T) return;
    uint32_t run = 0;
    int w = 0;
    for (; w + 8 <= nb; w += 8) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = g.hist[(size_t)(w + k) * T + t];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            g.hist[(size_t)(w + k) * T + t] = run;
            run += v[k];
        }
    }
    for (; w < nb; ++w) {
        const uint32_t v = g.hist[(size_t)w * T + t];
        g.hist[(size_t)w * T + t] = run;
        run += v;
    }
    g.tile_count[t] = run;
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
constexpr int kSortThreads = 256;
constexpr int kSortLdsCap = 2048;
constexpr int kSortPerThread = kSortLdsCap / kSortThreads;
constexpr int kBins = 1024;

// Split the sorted tile list into the sixteen cell lists (stable compaction by the cell block of
// cell_bands) and record where every duplicate landed (sorted_pos).  Every cell-list entry also gets
// the index of its backward record: the records of Gaussian i are the dense nby x nbx block of its
// cells, starting at rec_offsets[i] (K3).
template <typename GidAt>
__device__ __forceinline__ void finish_tile(const ViewCtx &c, int tile, uint32_t s, uint32_t n, GidAt &&gid_at)
{
    __shared__ uint32_t s_cbase[kCells];
    __shared__ uint32_t s_wc[kSortThreads / 64][kCells];
    const GeomPtrs &g = c.g;
    const BinPtrs &b = c.b;
    const uint32_t cap = c.cap;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tx = tile % c.vp.gx, ty = tile / c.vp.gx;
    if (tid < kCells) s_cbase[tid] = 0u;
    __syncthreads();
    for (uint32_t e0 = 0; e0 < n; e0 += kSortThreads) {
        const uint32_t e = e0 + tid;
        uint32_t m = 0u, gid = 0u, rec0 = 0u;
        int nbx = 0, ox = 0, oy = 0;   // record of cell (cx, cy) of this tile: rec0 + (oy + cy) * nbx + ox + cx
        if (e < n) {
            gid = gid_at(e);
            b.point_list[s + e] = gid;
            const float2 xy = g.xy[gid];
            const float4 co = g.conic_opacity[gid];
            const Rect rc = tile_rect(xy.x, xy.y, c.radii[gid], c.vp.gx, c.vp.gy);
            // Gaussian-major duplicate index of (gid, tile): offsets[gid] + position of the tile in gid's rect
            const uint32_t p = g.offsets[gid] + (uint32_t)((ty - rc.y0) * (rc.x1 - rc.x0) + (tx - rc.x0));
            if (p < cap) b.sorted_pos[p] = s + e;
            const Bands bd = cell_bands(xy.x, xy.y, co.x, co.y, co.z, co.w, rc);
            const int cx0 = max(bd.bx0, 4 * tx) - 4 * tx, cx1 = min(bd.bx0 + bd.nbx, 4 * tx + 4) - 4 * tx;
            const int cy0 = max(bd.by0, 4 * ty) - 4 * ty, cy1 = min(bd.by0 + bd.nby, 4 * ty + 4) - 4 * ty;
            for (int cy = cy0; cy < cy1; ++cy)
                for (int cx = cx0; cx < cx1; ++cx) m |= 1u << cell_id(cx, cy);
            rec0 = g.rec_offsets[gid];
            nbx = bd.nbx;
            ox = 4 * tx - bd.bx0;
            oy = 4 * ty - bd.by0;
        }
        uint64_t bal[kCells];
#pragma unroll
        for (int k = 0; k < kCells; ++k) bal[k] = __ballot((m >> k) & 1u);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < kCells; ++k) s_wc[wv][k] = (uint32_t)__popcll(bal[k]);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kCells; ++k) {
            if ((m >> k) & 1u) {
                uint32_t pos = s_cbase[k] + mbcnt(bal[k]);
                for (int w = 0; w < wv; ++w) pos += s_wc[w][k];
                if (s + pos < cap) {
                    const int qd = k >> 2, rr = k & 3;
                    const int cx = 2 * (qd & 1) + (rr & 1), cy = 2 * (qd >> 1) + (rr >> 1);
                    b.clist[(size_t)k * b.cap + s + pos] = make_uint2(gid, e);
                    b.cslot[(size_t)k * b.cap + s + pos] = rec0 + (uint32_t)((oy + cy) * nbx + ox + cx);
                }
            }
        }
        __syncthreads();
        if (tid < kCells) {
            uint32_t add = 0;
#pragma unroll
            for (int w = 0; w < kSortThreads / 64; ++w) add += s_wc[w][tid];
            s_cbase[tid] += add;
        }
        __syncthreads();
    }
    if (tid < kCells) g.ccount[tile * kCells + tid] = s_cbase[tid];
}

__global__ __launch_bounds__(kSortThreads) void k_tile_sort(BatchDesc d)
{
    __shared__ uint64_t s_a[kSortLdsCap];      // sorted keys
    __shared__ uint64_t s_b[kSortLdsCap];      // bucket-major keys
    __shared__ uint32_t s_bin[kBins + 1];      // histogram -> bucket ends
    __shared__ uint32_t s_cur[kBins];          // bucket starts / scatter cursors
    __shared__ uint32_t s_red[2 * (kSortThreads / 64)];
    const ViewCtx c = resolve(d, blockIdx.y);
    const GeomPtrs &g = c.g;
    const BinPtrs &b = c.b;
    const uint32_t cap = c.cap;
    const int t = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t s = g.tile_start[t];
    uint32_t n = g.tile_count[t];
    if (s >= cap) n = 0;
    else if (s + n > cap) n = cap - s;   // overflow: memory-safe, result flagged invalid by K3
    if (n == 0) {
        if (tid < kCells) g.ccount[t * kCells + tid] = 0u;
        return;
    }
    if (n <= (uint32_t)kSortLdsCap) {
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
        for (int o = 32; o > 0; o >>= 1) {
            dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, o, 64));
            dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, o, 64));
        }
        if (lane == 0) { s_red[wv] = dmin; s_red[4 + wv] = dmax; }
        for (int i = tid; i <= kBins; i += kSortThreads) s_bin[i] = 0u;
        __syncthreads();
        dmin = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
        dmax = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
        // depths are positive floats (> 0.2), so the bit patterns order like the values
        const float zmin = __uint_as_float(dmin), zspan = __uint_as_float(dmax) - zmin;
        const float scale = (zspan > 0.f && zspan < 3.0e38f) ? (float)kBins / zspan : 0.f;
        uint32_t bin[kSortPerThread];
#pragma unroll
        for (int r = 0; r < kSortPerThread; ++r) {
            const uint32_t e = r * kSortThreads + tid;
            bin[r] = 0u;
            if (e < n) {
                const float z = __uint_as_float((uint32_t)(key[r] >> 32));
                bin[r] = min((uint32_t)(kBins - 1), (uint32_t)max(0, f2i_sat((z - zmin) * scale)));
                atomicAdd(&s_bin[bin[r] + 1], 1u);
            }
        }
        __syncthreads();
        // ---- exclusive scan of the kBins counts (kBins / 256 per thread) ----
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
                s_cur[tid * per + i] = run;      // start of bucket tid*per+i
                run += loc[i];
                s_bin[1 + tid * per + i] = run;  // end of that bucket == start of the next
            }
        }
        __syncthreads();
        // ---- bucket-major scatter ----
#pragma unroll
        for (int r = 0; r < kSortPerThread; ++r) {
            const uint32_t e = r * kSortThreads + tid;
            if (e < n) s_b[atomicAdd(&s_cur[bin[r]], 1u)] = key[r];
        }
        __syncthreads();
        // ---- rank inside the bucket by the full key, write the sorted keys ----
#pragma unroll
        for (int r = 0; r < kSortPerThread; ++r) {
            const uint32_t e = r * kSortThreads + tid;
            if (e < n) {
                const uint32_t lo = s_bin[bin[r]], hi = s_bin[bin[r] + 1];
                uint32_t rank = 0;
                for (uint32_t j = lo; j < hi; ++j) rank += (s_b[j] < key[r]) ? 1u : 0u;
                s_a[lo + rank] = key[r];
            }
        }
        __syncthreads();
        finish_tile(c, t, s, n, [&](uint32_t e) { return (uint32_t)s_a[e]; });
    } else {
        // ---- large tile: the same bucket sort with the keys resident in HBM (L2) ----
        // temp (bucket-major keys) lives in the tile's own cell-0 list segment, which
        // finish_tile() only writes afterwards; the sorted keys go back in place.
        uint32_t *kd = b.u_depth + s, *ki_ = b.u_idx + s;
        uint64_t *tmp = reinterpret_cast<uint64_t *>(b.clist + s);
        uint32_t dmin = 0xFFFFFFFFu, dmax = 0u;
        for (uint32_t e = tid; e < n; e += kSortThreads) {
            const uint32_t dz = kd[e];
            dmin = min(dmin, dz);
            dmax = max(dmax, dz);
        }
        for (int o = 32; o > 0; o >>= 1) {
            dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, o, 64));
            dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, o, 64));
        }
        if (lane == 0) { s_red[wv] = dmin; s_red[4 + wv] = dmax; }
        for (int i = tid; i <= kBins; i += kSortThreads) s_bin[i] = 0u;
        __syncthreads();
        dmin = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
        dmax = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
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
            tmp[atomicAdd(&s_cur[bin_of(dz)], 1u)] = ((uint64_t)dz << 32) | ki_[e];
        }
        __syncthreads();   // workgroup-scope: tmp writes visible to the whole workgroup
        for (uint32_t e = tid; e < n; e += kSortThreads) {
            const uint64_t k = tmp[e];
            const uint32_t bn = bin_of((uint32_t)(k >> 32));
            const uint32_t lo = s_bin[bn], hi = s_bin[bn + 1];
            uint32_t rank = 0;
            for (uint32_t j = lo; j < hi; ++j) rank += (tmp[j] < k) ? 1u : 0u;
            kd[lo + rank] = (uint32_t)(k >> 32);
            ki_[lo + rank] = (uint32_t)k;
        }
        __syncthreads();
        finish_tile(c, t, s, n, [&](uint32_t e) { return ki_[e]; });
    }
}

// ---------------------------------------------------------------------------------------- K4b
// Launch order of the blend kernels: the (view, tile) pairs of the WHOLE batch by DESCENDING cell-list
// length (longest-processing-time first).  A wave walks its lists serially, so the longest list
// (silhouette cells: > 1000 entries against a mean of 85) bounds the kernel from below; started
// last it runs alone at the end, started first it overlaps with everything else.  One workgroup,
// 256-bucket counting sort; rank i is stored in the geom workspace of view i / T, slot i % T, as
// view << 16 | tile.
__device__ __forceinline__ uint32_t *order_slot(const BatchDesc &d, const GeomLayout &L, uint32_t rank)
{
    return reinterpret_cast<uint32_t *>(d.geom + (size_t)(rank / (uint32_t)L.T) * d.geom_stride + L.order) +
           rank % (uint32_t)L.T;
}
constexpr int kOrderThreads = 1024;
__global__ __launch_bounds__(kOrderThreads) void k_tile_order(BatchDesc d)
{
    __shared__ uint32_t s_hist[256], s_cur[256], s_max;
    const GeomLayout L = geom_layout(d.N, d.H, d.W);
    const int T = L.T, tid = threadIdx.x;
    const uint32_t n = (uint32_t)d.B * (uint32_t)T;
    auto weight = [&](uint32_t i) {
        const uint32_t v = i / (uint32_t)T, t = i % (uint32_t)T;
        const uint4 *p = reinterpret_cast<const uint4 *>(d.geom + (size_t)v * d.geom_stride + L.ccount) + (size_t)t * (kCells / 4);
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < kCells / 4; ++k) {
            const uint4 x = p[k];
            w = max(max(w, x.x), max(max(x.y, x.z), x.w));
        }
        return w;
    };
    if (tid == 0) s_max = 1u;
    if (tid < 256) s_hist[tid] = 0u;
    __syncthreads();
    uint32_t wmax = 0;
    for (uint32_t i = tid; i < n; i += kOrderThreads) wmax = max(wmax, weight(i));
    wmax = wave_max_u32(wmax);
    if ((tid & 63) == 0) atomicMax(&s_max, wmax);
    __syncthreads();
    const float scale = 255.0f / (float)s_max;
    auto bucket = [&](uint32_t w) { return 255 - min(255, (int)((float)w * scale)); };
    for (uint32_t i = tid; i < n; i += kOrderThreads) atomicAdd(&s_hist[bucket(weight(i))], 1u);
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int b = 0; b < 256; ++b) { s_cur[b] = run; run += s_hist[b]; }
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kOrderThreads) {
        const uint32_t rank = atomicAdd(&s_cur[bucket(weight(i))], 1u);
        *order_slot(d, L, rank) = ((i / (uint32_t)T) << 16) | (i % (uint32_t)T);
    }
}

int launch_colscan(const BatchDesc &d, hipStream_t st)
{
    const int T = ((d.W + kTile - 1) / kTile) * ((d.H + kTile - 1) / kTile);
    if (T <= 0) return DM4D_OK;
    ProfScope prof_(kKColscan, st);
    hipLaunchKernelGGL(k_colscan, dim3((T + kColThreads - 1) / kColThreads, d.B), dim3(kColThreads), 0, st, d);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int launch_tile_sort(const BatchDesc &d, hipStream_t st)
{
    const int T = ((d.W + kTile - 1) / kTile) * ((d.H + kTile - 1) / kTile);
    if (T <= 0) return DM4D_OK;
    ProfScope prof_(kKTileSort, st);
    hipLaunchKernelGGL(k_tile_sort, dim3(T, d.B), dim3(kSortThreads), 0, st, d);
    DM4D_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(kOrderThreads), 0, st, d);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // namespace dm4d
