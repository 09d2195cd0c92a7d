// heat.hip -- the device side of the HEAT-METHOD geodesic distances of the deformation-graph construction (gfx950).
//
// Reference: DynamicSuGaRModel.build_deformation_graph(mode="geodisc")
// (custom/threestudio-dreammesh4d/geometry/dynamic_sugar.py:794-861): for every one of the V vertices i one
// potpourri3d.MeshHeatMethodDistanceSolver.compute_distance(i) (:802,834) -- two sparse solves on the CPU -- read at the M
// nodes' nearest vertices (:806-812,836), then the K + 1 nearest nodes (:838).  V sequential solves: minutes at 16k vertices.
//
// The heat method (Crane, Weischedel, Wardetzky 2013; restated in oracle/graph.py::heat_method_distances):
//     (A + t L) u_i = delta_i        X_i = -grad u_i / |grad u_i| per face        L phi_i = -div X_i
// What the graph needs is only the RANKING of phi_i(t_m) over the nodes m (the weights are Euclidean, :842-855), and L is
// symmetric: phi_i(t_m) = e_{t_m}^T L^+ (-div X_i) = -sum_faces X_i[f] . W_m[f] with W_m[f] = sum_k g_m[f_k] D[f, k] and
// g_m = L^+ e_{t_m}.  So the ill-conditioned Poisson system is solved M times (once per NODE), not V times; the V heat
// systems are well conditioned (t = h^2: ~30 CG iterations); the V x M table is one float64 GEMM X^T W (rocBLAS through
// torch.matmul -- a plain library GEMM) over chunks of sources.  This file holds the three hand-written pieces:
//   * dm4d_cg_batched_f64       Jacobi-preconditioned conjugate gradients for MANY right-hand sides of one sparse SPD (or
//                               PSD with consistent right-hand sides) matrix: unknown-major [V][S] layout, so every kernel
//                               reads and writes rows of S contiguous doubles; one SpMM per iteration; all reductions in two
//                               deterministic stages (no atomics: every rank must build the same graph)
//   * dm4d_heat_face_directions X_i per face from u_i, written transposed ([3F][S]) for the GEMM
//   * dm4d_graph_select_knn     the K + 1 smallest of M scores per source + the reference's Euclidean weights
// float64 throughout: the Poisson system's condition number is ~1e4 on the bench mesh and the ranking must not depend on
// solver noise.
#include "common.h"
#include "../../include/dm4d.h"

namespace dm4d {

constexpr int kCgCols = 64;        // right-hand sides per workgroup (one per lane of a wave: contiguous doubles)
constexpr int kCgRowLanes = 4;     // waves per workgroup: wave w takes rows w, w + 4, ... of the workgroup's row block
constexpr int kCgRows = 32;        // rows per workgroup

struct CgDesc {
    int V, S;
    const int32_t *off, *col;
    const double *val, *dinv;
    double *x, *r, *z, *p, *Ap;
    double *part;                   // [2][n_row_blocks][S] partial dot products
    double *pAp, *rz, *rz_new, *rr, *bb;   // [S]
};

__device__ __forceinline__ void block_rows_reduce(double a, double b, double *pa, double *pb, const int col, const int S, const bool live)
{
    __shared__ double s_a[kCgRowLanes][kCgCols], s_b[kCgRowLanes][kCgCols];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    s_a[w][lane] = a;
    s_b[w][lane] = b;
    __syncthreads();
    if (w == 0 && live) {
        pa[col] = ((s_a[0][lane] + s_a[1][lane]) + s_a[2][lane]) + s_a[3][lane];
        if (pb) pb[col] = ((s_b[0][lane] + s_b[1][lane]) + s_b[2][lane]) + s_b[3][lane];
    }
}

// r = b - A x (x given), z = Dinv r, p = z; partials of r.z and b.b
__global__ __launch_bounds__(256) void k_cg_init(CgDesc d, const double *__restrict__ b)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int s = blockIdx.x * kCgCols + lane;
    const bool live = s < d.S;
    double rz = 0.0, bb = 0.0;
    if (live)
        for (int v = blockIdx.y * kCgRows + w; v < min(d.V, (int)(blockIdx.y + 1) * kCgRows); v += kCgRowLanes) {
            double ax = 0.0;
            for (int e = d.off[v]; e < d.off[v + 1]; ++e) ax += d.val[e] * d.x[(size_t)d.col[e] * d.S + s];
            const size_t o = (size_t)v * d.S + s;
            const double bv = b[o], rv = bv - ax, zv = rv * d.dinv[v];
            d.r[o] = rv; d.z[o] = zv; d.p[o] = zv;
            rz += rv * zv;
            bb += bv * bv;
        }
    const size_t nb = gridDim.y;
    block_rows_reduce(rz, bb, d.part + (size_t)blockIdx.y * d.S, d.part + (nb + blockIdx.y) * d.S, s, d.S, live);
}

// Ap = A p; partial p.Ap
__global__ __launch_bounds__(256) void k_cg_spmm(CgDesc d)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int s = blockIdx.x * kCgCols + lane;
    const bool live = s < d.S;
    double acc = 0.0;
    if (live)
        for (int v = blockIdx.y * kCgRows + w; v < min(d.V, (int)(blockIdx.y + 1) * kCgRows); v += kCgRowLanes) {
            double ap = 0.0;
            for (int e = d.off[v]; e < d.off[v + 1]; ++e) ap += d.val[e] * d.p[(size_t)d.col[e] * d.S + s];
            const size_t o = (size_t)v * d.S + s;
            d.Ap[o] = ap;
            acc += d.p[o] * ap;
        }
    block_rows_reduce(acc, 0.0, d.part + (size_t)blockIdx.y * d.S, nullptr, s, d.S, live);
}

// out0[s] = sum over row blocks of part[0][.][s] (and out1 of part[1][.][s]) in block order
__global__ __launch_bounds__(256) void k_cg_reduce(int S, int nb, const double *__restrict__ part, double *__restrict__ out0,
                                                   double *__restrict__ out1)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    double a = 0.0, b = 0.0;
    for (int k = 0; k < nb; ++k) {
        a += part[(size_t)k * S + s];
        if (out1) b += part[((size_t)nb + k) * S + s];
    }
    out0[s] = a;
    if (out1) out1[s] = b;
}

// x += alpha p, r -= alpha Ap, z = Dinv r; partials of r.z and r.r
__global__ __launch_bounds__(256) void k_cg_update(CgDesc d)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int s = blockIdx.x * kCgCols + lane;
    const bool live = s < d.S;
    double rz = 0.0, rr = 0.0;
    if (live) {
        const double pap = d.pAp[s], alpha = pap > 0.0 ? d.rz[s] / pap : 0.0;
        for (int v = blockIdx.y * kCgRows + w; v < min(d.V, (int)(blockIdx.y + 1) * kCgRows); v += kCgRowLanes) {
            const size_t o = (size_t)v * d.S + s;
            d.x[o] += alpha * d.p[o];
            const double rv = d.r[o] - alpha * d.Ap[o], zv = rv * d.dinv[v];
            d.r[o] = rv; d.z[o] = zv;
            rz += rv * zv;
            rr += rv * rv;
        }
    }
    const size_t nb = gridDim.y;
    block_rows_reduce(rz, rr, d.part + (size_t)blockIdx.y * d.S, d.part + (nb + blockIdx.y) * d.S, s, d.S, live);
}

// p = z + beta p; rz <- rz_new
__global__ __launch_bounds__(256) void k_cg_direction(CgDesc d)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int s = blockIdx.x * kCgCols + lane;
    if (s >= d.S) return;
    const double rz = d.rz[s], beta = rz > 0.0 ? d.rz_new[s] / rz : 0.0;
    for (int v = blockIdx.y * kCgRows + w; v < min(d.V, (int)(blockIdx.y + 1) * kCgRows); v += kCgRowLanes) {
        const size_t o = (size_t)v * d.S + s;
        d.p[o] = d.z[o] + beta * d.p[o];
    }
}
// (after every workgroup of k_cg_direction has read rz)
__global__ __launch_bounds__(256) void k_cg_roll(int S, double *__restrict__ rz, const double *__restrict__ rz_new, const double *__restrict__ rr,
                                                 const double *__restrict__ bb, double *__restrict__ worst)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    double rel = 0.0;
    if (s < S) {
        rz[s] = rz_new[s];
        rel = bb[s] > 0.0 ? rr[s] / bb[s] : 0.0;
    }
    // max over the workgroup -> one double per workgroup (the host takes the max of a handful of values)
    __shared__ double s_m[256];
    s_m[threadIdx.x] = rel;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) s_m[threadIdx.x] = fmax(s_m[threadIdx.x], s_m[threadIdx.x + k]);
        __syncthreads();
    }
    if (threadIdx.x == 0) worst[blockIdx.x] = s_m[0];
}

// X^T[3 f + c][s] = -(grad u_s)_f / |(grad u_s)_f|, grad u = sum_k u[f_k] G[f][k]   (G[f][k] = N x e_k / (2 A), host, float64)
__global__ __launch_bounds__(256) void k_heat_face_dirs(int F, int S, const int32_t *__restrict__ faces, const double *__restrict__ G,
                                                        const double *__restrict__ U, double *__restrict__ XT)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int s = blockIdx.x * kCgCols + lane;
    const int f = blockIdx.y * 4 + w;
    if (s >= S || f >= F) return;
    const double u0 = U[(size_t)faces[3 * f] * S + s], u1 = U[(size_t)faces[3 * f + 1] * S + s], u2 = U[(size_t)faces[3 * f + 2] * S + s];
    const double *g = G + (size_t)f * 9;
    const double gx = (u0 * g[0] + u1 * g[3]) + u2 * g[6], gy = (u0 * g[1] + u1 * g[4]) + u2 * g[7], gz = (u0 * g[2] + u1 * g[5]) + u2 * g[8];
    const double n = fmax(sqrt((gx * gx + gy * gy) + gz * gz), 1e-300);
    XT[((size_t)3 * f + 0) * S + s] = -gx / n;
    XT[((size_t)3 * f + 1) * S + s] = -gy / n;
    XT[((size_t)3 * f + 2) * S + s] = -gz / n;
}

constexpr int kSelMaxK = 16;
// score[m][s] (row stride ld): for source vertex v0 + s the K + 1 nodes of smallest score, ties towards the lower node index;
// weights (1 - e_k / e_K)^2 with Euclidean distances to the node positions, row-normalised (dynamic_sugar.py:842-861)
__global__ __launch_bounds__(256) void k_graph_select(int S, int M, int K, const double *__restrict__ score, int ld, int v0,
                                                      const float *__restrict__ verts, const float *__restrict__ node_xyz,
                                                      int64_t *__restrict__ idx, float *__restrict__ weights)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    double bd[kSelMaxK + 1];
    int bi[kSelMaxK + 1];
    const int K1 = K + 1;
    for (int k = 0; k < K1; ++k) { bd[k] = 1.0e300; bi[k] = -1; }
    for (int m = 0; m < M; ++m) {
        const double dm = score[(size_t)m * ld + s];
        if (dm < bd[K1 - 1]) {
            int k = K1 - 1;
            while (k > 0 && dm < bd[k - 1]) { bd[k] = bd[k - 1]; bi[k] = bi[k - 1]; --k; }
            bd[k] = dm;
            bi[k] = m;
        }
    }
    const size_t v = (size_t)v0 + s;
    const float px = verts[3 * v], py = verts[3 * v + 1], pz = verts[3 * v + 2];
    float e[kSelMaxK + 1];
    for (int k = 0; k < K1; ++k) {
        const int m = bi[k] < 0 ? 0 : bi[k];
        const float dx = px - node_xyz[3 * (size_t)m], dy = py - node_xyz[3 * (size_t)m + 1], dz = pz - node_xyz[3 * (size_t)m + 2];
        e[k] = sqrtf((dx * dx + dy * dy) + dz * dz);
    }
    float w[kSelMaxK], sum = 0.f;
    for (int k = 0; k < K; ++k) {
        const float t = 1.0f - e[k] / e[K];
        w[k] = t * t;
        sum += w[k];
    }
    for (int k = 0; k < K; ++k) {
        idx[v * K + k] = bi[k];
        weights[v * K + k] = w[k] / sum;
    }
}

}  // namespace dm4d

using namespace dm4d;

extern "C" {

static size_t cg_vec(int V, int S) { return ((size_t)V * S * 8 + 255) / 256 * 256; }
static int cg_row_blocks(int V) { return (V + kCgRows - 1) / kCgRows; }

/* r, z, p, Ap (V x S doubles each) + two stages of partial dot products + 5 S-vectors + per-workgroup maxima */
size_t dm4d_cg_batched_scratch_bytes(int32_t V, int32_t S)
{
    if (V <= 0 || S <= 0) return 256;
    return 4 * cg_vec(V, S) + ((size_t)2 * cg_row_blocks(V) * S * 8 + 255) / 256 * 256 + (size_t)6 * ((size_t)S * 8 + 255) / 256 * 256 + 4096;
}

int dm4d_cg_batched_f64(int32_t V, int32_t S, const int32_t *csr_offsets, const int32_t *csr_cols, const double *csr_vals,
                        const double *diag_inv, const double *B, double *X, void *scratch, int32_t max_iter, double tol,
                        int32_t check_every, double *final_rel_residual, dm4d_stream_t stream)
{
    if (V <= 0 || S <= 0 || max_iter <= 0 || check_every <= 0 || !(tol > 0.0)) { set_error("cg: bad V / S / max_iter / check_every / tol"); return DM4D_ERR_INVALID; }
    if (!csr_offsets || !csr_cols || !csr_vals || !diag_inv || !B || !X || !scratch) { set_error("cg: null tensor"); return DM4D_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    CgDesc d;
    d.V = V; d.S = S; d.off = csr_offsets; d.col = csr_cols; d.val = csr_vals; d.dinv = diag_inv; d.x = X;
    char *p = (char *)scratch;
    const size_t vec = cg_vec(V, S), svec = ((size_t)S * 8 + 255) / 256 * 256;
    const int nb = cg_row_blocks(V);
    d.r = (double *)p; p += vec;
    d.z = (double *)p; p += vec;
    d.p = (double *)p; p += vec;
    d.Ap = (double *)p; p += vec;
    d.part = (double *)p; p += ((size_t)2 * nb * S * 8 + 255) / 256 * 256;
    d.pAp = (double *)p; p += svec;
    d.rz = (double *)p; p += svec;
    d.rz_new = (double *)p; p += svec;
    d.rr = (double *)p; p += svec;
    d.bb = (double *)p; p += svec;
    double *worst = (double *)p;
    const dim3 grid((S + kCgCols - 1) / kCgCols, nb), sgrid((S + 255) / 256);
    if (sgrid.x > 256) { set_error("cg: at most %d right-hand sides per call", 256 * 256); return DM4D_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(k_cg_init, grid, dim3(256), 0, st, d, B);
    hipLaunchKernelGGL(k_cg_reduce, sgrid, dim3(256), 0, st, S, nb, (const double *)d.part, d.rz, d.bb);
    DM4D_HIP_CHECK(hipGetLastError());
    double rel2 = 1.0;
    int it = 0;
    double host_worst[256];
    while (it < max_iter) {
        for (int k = 0; k < check_every && it < max_iter; ++k, ++it) {
            hipLaunchKernelGGL(k_cg_spmm, grid, dim3(256), 0, st, d);
            hipLaunchKernelGGL(k_cg_reduce, sgrid, dim3(256), 0, st, S, nb, (const double *)d.part, d.pAp, (double *)nullptr);
            hipLaunchKernelGGL(k_cg_update, grid, dim3(256), 0, st, d);
            hipLaunchKernelGGL(k_cg_reduce, sgrid, dim3(256), 0, st, S, nb, (const double *)d.part, d.rz_new, d.rr);
            hipLaunchKernelGGL(k_cg_direction, grid, dim3(256), 0, st, d);
            hipLaunchKernelGGL(k_cg_roll, sgrid, dim3(256), 0, st, S, d.rz, (const double *)d.rz_new, (const double *)d.rr, (const double *)d.bb, worst);
        }
        DM4D_HIP_CHECK(hipGetLastError());
        DM4D_HIP_CHECK(hipMemcpyAsync(host_worst, worst, sgrid.x * sizeof(double), hipMemcpyDeviceToHost, st));
        DM4D_HIP_CHECK(hipStreamSynchronize(st));
        rel2 = 0.0;
        for (unsigned k = 0; k < sgrid.x; ++k) rel2 = host_worst[k] > rel2 ? host_worst[k] : rel2;
        if (!(rel2 == rel2)) { set_error("cg: NaN residual after %d iterations (is the matrix positive semi-definite?)", it); return DM4D_ERR_INVALID; }
        if (rel2 <= tol * tol) break;
    }
    if (final_rel_residual) *final_rel_residual = sqrt(rel2);
    return it;        /* iterations run (>= 0) */
}

int dm4d_heat_face_directions(int32_t F, int32_t S, const int32_t *faces, const double *G, const double *U, double *XT, dm4d_stream_t stream)
{
    if (F <= 0 || S <= 0 || !faces || !G || !U || !XT) { set_error("heat: bad arguments"); return DM4D_ERR_INVALID; }
    hipLaunchKernelGGL(k_heat_face_dirs, dim3((S + kCgCols - 1) / kCgCols, (F + 3) / 4), dim3(256), 0, (hipStream_t)stream, F, S, faces, G, U, XT);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_graph_select_knn(int32_t S, int32_t M, int32_t K, const double *score, int32_t ld, int32_t first_vertex, const float *verts,
                          const float *node_xyz, int64_t *neighbor_idx, float *neighbor_weights, dm4d_stream_t stream)
{
    if (S <= 0 || M <= 0 || K <= 0 || K > kSelMaxK || K + 1 > M || ld < S || first_vertex < 0) {
        set_error("select: need S > 0, M > K > 0, K <= %d, ld >= S", kSelMaxK);
        return DM4D_ERR_INVALID;
    }
    if (!score || !verts || !node_xyz || !neighbor_idx || !neighbor_weights) { set_error("select: null tensor"); return DM4D_ERR_INVALID; }
    hipLaunchKernelGGL(k_graph_select, dim3((S + 255) / 256), dim3(256), 0, (hipStream_t)stream, S, M, K, score, ld, first_vertex, verts, node_xyz,
                       neighbor_idx, neighbor_weights);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // extern "C"
