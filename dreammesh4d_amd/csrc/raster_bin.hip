// raster_bin.hip -- binning kernels of the tile rasterizer (gfx950).
//
//   K2 k_colscan    : per-tile exclusive scan of the dense per-workgroup tile histogram written
//                     by K1 (-> per-(workgroup, tile) slot bases, tile counts); the tile segment
//                     starts (== upstream's `ranges`) follow from a scan K3 does in LDS.
//   K4 k_tile_sort  : one workgroup per tile; sorts the tile's duplicates by the 64-bit key
//                     (depth bits << 32 | Gaussian id) in LDS with a bitonic network.
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
constexpr int kSortThreads = 256;
constexpr int kSortLdsCap = 4096;   // duplicates per tile sorted in LDS (48 KB); more -> global path

__device__ __forceinline__ uint32_t next_pow2(uint32_t v)
{
    v--;
    v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    return v + 1;
}

// Ascending bitonic network for arbitrary n ("flip" formulation): positions >= n behave as
// +inf and never move, so compare-exchanges touching them are skipped.
template <typename Swap>
__device__ __forceinline__ void bitonic_network(uint32_t n, Swap &&cmpswap)
{
    const uint32_t np2 = next_pow2(n);
    const uint32_t half_pairs = np2 >> 1;
    for (uint32_t k = 2; k <= np2; k <<= 1) {
        // flip step
        {
            const uint32_t h = k >> 1;
            for (uint32_t t = threadIdx.x; t < half_pairs; t += kSortThreads) {
                const uint32_t i = ((t / h) * k) + (t % h);
                const uint32_t l = i ^ (k - 1);
                if (l < n) cmpswap(i, l);
            }
            __syncthreads();
        }
        for (uint32_t j = k >> 2; j >= 1; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < half_pairs; t += kSortThreads) {
                const uint32_t i = ((t / j) * (j << 1)) + (t % j);
                const uint32_t l = i + j;
                if (l < n) cmpswap(i, l);
            }
            __syncthreads();
        }
    }
}

// Split the sorted tile list into the four quadrant lists (stable compaction by quadrant_mask).
template <typename GidAt>
__device__ __forceinline__ void build_quadrant_lists(const ViewParams &vp, int tile, uint32_t s, uint32_t n,
                                                     const GeomPtrs &g, const BinPtrs &b, uint32_t cap, GidAt &&gid_at)
{
    __shared__ uint32_t s_qbase[4];
    __shared__ uint32_t s_wq[kSortThreads / 64][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float ox = (float)((tile % vp.gx) * kTile), oy = (float)((tile / vp.gx) * kTile);
    if (tid < 4) s_qbase[tid] = 0u;
    __syncthreads();
    for (uint32_t e0 = 0; e0 < n; e0 += kSortThreads) {
        const uint32_t e = e0 + tid;
        uint32_t m = 0u, gid = 0u;
        if (e < n) {
            gid = gid_at(e);
            const float2 xy = g.xy[gid];
            const float4 co = g.conic_opacity[gid];
            m = quadrant_mask(xy.x, xy.y, co.x, co.y, co.z, co.w, ox, oy);
        }
        uint64_t bal[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bal[q] = __ballot((m >> q) & 1u);
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s_wq[wv][q] = (uint32_t)__popcll(bal[q]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if ((m >> q) & 1u) {
                uint32_t pos = s_qbase[q] + mbcnt(bal[q]);
                for (int w = 0; w < wv; ++w) pos += s_wq[w][q];
                if (s + pos < cap) b.qlist[(size_t)q * b.cap + s + pos] = make_uint2(gid, e);
            }
        }
        __syncthreads();
        if (tid < 4) {
            uint32_t add = 0;
#pragma unroll
            for (int w = 0; w < kSortThreads / 64; ++w) add += s_wq[w][tid];
            s_qbase[tid] += add;
        }
        __syncthreads();
    }
    if (tid < 4) g.qcount[tile * 4 + tid] = s_qbase[tid];
}

__global__ __launch_bounds__(kSortThreads) void k_tile_sort(BatchDesc d)
{
    const ViewCtx c = resolve(d, blockIdx.y);
    const ViewParams &vp = c.vp;
    const GeomPtrs &g = c.g;
    const BinPtrs &b = c.b;
    const uint32_t cap = c.cap;
    __shared__ uint64_t s_key[kSortLdsCap];
    __shared__ uint32_t s_p[kSortLdsCap];
    const int t = blockIdx.x;
    const int tid = threadIdx.x;
    const uint32_t s = g.tile_start[t];
    uint32_t n = g.tile_count[t];
    if (s >= cap) n = 0;
    else if (s + n > cap) n = cap - s;   // overflow: memory-safe, result flagged invalid by K3
    if (n == 0) {
        if (tid < 4) g.qcount[t * 4 + tid] = 0u;
        return;
    }
    if (n <= (uint32_t)kSortLdsCap) {
        for (uint32_t e = tid; e < n; e += kSortThreads) {
            s_key[e] = ((uint64_t)b.u_depth[s + e] << 32) | b.u_idx[s + e];
            s_p[e] = b.u_p[s + e];
        }
        __syncthreads();
        if (n > 1) {
            bitonic_network(n, [&](uint32_t i, uint32_t l) {
                const uint64_t ki = s_key[i], kl = s_key[l];
                if (kl < ki) {
                    s_key[i] = kl; s_key[l] = ki;
                    const uint32_t pi = s_p[i]; s_p[i] = s_p[l]; s_p[l] = pi;
                }
            });
        }
        for (uint32_t e = tid; e < n; e += kSortThreads) {
            b.point_list[s + e] = (uint32_t)s_key[e];
            const uint32_t p = s_p[e];
            if (p < cap) b.sorted_pos[p] = s + e;
        }
        build_quadrant_lists(vp, t, s, n, g, b, cap, [&](uint32_t e) { return (uint32_t)s_key[e]; });
    } else {
        // Oversized tile: same network on the HBM-resident segment (one workgroup; rare).
        uint32_t *kd = b.u_depth + s, *ki_ = b.u_idx + s, *kp = b.u_p + s;
        bitonic_network(n, [&](uint32_t i, uint32_t l) {
            const uint64_t a = ((uint64_t)kd[i] << 32) | ki_[i], c = ((uint64_t)kd[l] << 32) | ki_[l];
            if (c < a) {
                uint32_t x;
                x = kd[i]; kd[i] = kd[l]; kd[l] = x;
                x = ki_[i]; ki_[i] = ki_[l]; ki_[l] = x;
                x = kp[i]; kp[i] = kp[l]; kp[l] = x;
            }
        });
        for (uint32_t e = tid; e < n; e += kSortThreads) {
            b.point_list[s + e] = ki_[e];
            const uint32_t p = kp[e];
            if (p < cap) b.sorted_pos[p] = s + e;
        }
        build_quadrant_lists(vp, t, s, n, g, b, cap, [&](uint32_t e) { return ki_[e]; });
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
    return DM4D_OK;
}

}  // namespace dm4d
