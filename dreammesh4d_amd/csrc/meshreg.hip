// meshreg.hip -- mesh regularisers of the dynamic stage on the deformed vertices (gfx950).
//
// As-rigid-as-possible energy with GIVEN vertex rotations (the skinned rotations), as the reference's
// ARAPCoach.compute_arap_energy(xyz_prime, vert_rotations) is called every iteration for the key frames and for
// 10 densely sampled inter-frames (custom/threestudio-dreammesh4d/utils/arap_utils.py:183-224,
// system/sugar_4dgen.py:304-311,331-385):
//
//     E_t = sum_i sum_{j in N(i)} w_ij || (x'_i - x'_j) - R_i (x_i - x_j) ||^2
//
// The reference materialises [V, max_valence, 3] edge tensors and runs bmm / norm / sum per timestamp in a Python
// loop; here one launch covers all T timestamps over a static CSR adjacency (weights and rest edges precomputed
// once).  The backward gathers: vertex i adds the terms of its own edges and, through the reverse-edge index,
// of the edges that point at it -- no atomics, deterministic.
#include <mutex>
#include "common.h"
#include "../../include/dm4d.h"

namespace dm4d {

struct ArapAdj {
    int V;
    const int32_t *off, *nbr, *rev;   // CSR [V+1], [E], [E] (rev[e] = index of the edge nbr[e] -> i)
    const float *w, *e;               // [E], [E,3]: weight, rest edge x_i - x_j
};

__device__ __forceinline__ void edge_residual(const float *__restrict__ xi, const float *__restrict__ xj,
                                              const float *__restrict__ R, const float *__restrict__ e, float s[3])
{
#pragma unroll
    for (int a = 0; a < 3; ++a) s[a] = (xi[a] - xj[a]) - ((R[3 * a] * e[0] + R[3 * a + 1] * e[1]) + R[3 * a + 2] * e[2]);
}

// A vertex is walked by a GROUP of kSub consecutive lanes (edge e0 + sub, + kSub, ...), their partial sums meet in a xor butterfly
// inside the group: one thread per vertex made the kernel as long as its highest valence (the two poles of the bench's uv
// sphere have ~360 edges against a mean of 6: 33 / 75 us forward / backward, and 343 us for the normal-consistency gather
// below, whose poles touch ~700 pair roles).  The order of the sum is fixed by the lane assignment: deterministic.
constexpr int kSub = 8;
template <int N>
__device__ __forceinline__ void group_sum(float (&v)[N])
{
#pragma unroll
    for (int m = 1; m < kSub; m <<= 1)
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] += __shfl_xor(v[k], m, kSub);
}

// per (timestamp, vertex): energy of its outgoing edges
__global__ __launch_bounds__(256) void k_arap_fwd(ArapAdj a, const float *__restrict__ xyz, const float *__restrict__ rot,
                                                  float *__restrict__ energy /* [T][V] */)
{
    const int gid = blockIdx.x * 256 + threadIdx.x, i = gid / kSub, sub = gid % kSub;
    if (i >= a.V) return;                      // (whole groups leave together: 256 % kSub == 0)
    const size_t t = blockIdx.y;
    xyz += t * a.V * 3;
    rot += t * a.V * 9;
    float R[9], xi[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = rot[9 * (size_t)i + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) xi[k] = xyz[3 * (size_t)i + k];
    float acc[1] = {0.f};
    const int e1 = a.off[i + 1];
#pragma unroll 2
    for (int eidx = a.off[i] + sub; eidx < e1; eidx += kSub) {
        const int j = a.nbr[eidx];
        float s[3];
        edge_residual(xi, xyz + 3 * (size_t)j, R, a.e + 3 * (size_t)eidx, s);
        acc[0] += a.w[eidx] * ((s[0] * s[0] + s[1] * s[1]) + s[2] * s[2]);
    }
    group_sum(acc);
    if (sub == 0) energy[t * a.V + i] = acc[0];
}

// per (timestamp, vertex): dE/dx'_i and dE/dR_i, scaled by the upstream gradient of E_t
__global__ __launch_bounds__(256) void k_arap_bwd(ArapAdj a, const float *__restrict__ xyz, const float *__restrict__ rot,
                                                  const float *__restrict__ g_energy /* [T] */, float *__restrict__ g_xyz,
                                                  float *__restrict__ g_rot)
{
    const int gid = blockIdx.x * 256 + threadIdx.x, i = gid / kSub, sub = gid % kSub;
    if (i >= a.V) return;
    const size_t t = blockIdx.y;
    xyz += t * a.V * 3;
    rot += t * a.V * 9;
    const float ge = 2.0f * g_energy[t];
    float R[9], xi[3], g[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // gx[3] | gR[9]
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = rot[9 * (size_t)i + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) xi[k] = xyz[3 * (size_t)i + k];
    const int e1 = a.off[i + 1];
#pragma unroll 2
    for (int eidx = a.off[i] + sub; eidx < e1; eidx += kSub) {
        const int j = a.nbr[eidx], m = a.rev[eidx];
        const float *xj = xyz + 3 * (size_t)j;
        const float *e = a.e + 3 * (size_t)eidx;
        float s[3], sr[3];
        edge_residual(xi, xj, R, e, s);                                        // own edge i -> j
        edge_residual(xj, xi, rot + 9 * (size_t)j, a.e + 3 * (size_t)m, sr);   // reverse edge j -> i
        const float w = a.w[eidx], wr = a.w[m];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            g[c] += w * s[c] - wr * sr[c];
#pragma unroll
            for (int b = 0; b < 3; ++b) g[3 + 3 * c + b] -= w * s[c] * e[b];
        }
    }
    group_sum(g);
    if (sub != 0) return;
    if (g_xyz) {
        float *o = g_xyz + (t * a.V + i) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = ge * g[c];
    }
    if (g_rot) {
        float *o = g_rot + (t * a.V + i) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) o[k] = ge * g[3 + k];
    }
}


// ---------------------------------------------------------------------------------------- normal consistency
// pytorch3d.loss.mesh_normal_consistency of the T deformed meshes (system/sugar_4dgen.py:214-226, lambda 100 in
// configs/sugar_dynamic_dg.yaml:146).  For every PAIR of faces that share an edge (v0, v1) with opposite vertices
// a and b:   n0 = (v1 - v0) x (a - v0),  n1 = -((v1 - v0) x (b - v0)),  term = 1 - cos(n0, n1);
// loss_t = mean of the terms (pytorch3d then averages over the meshes of the batch).  The pairs are static (the
// topology never changes), so the forward is one thread per (mesh, pair) and the backward a gather per (mesh,
// vertex) over the (pair, role) items that touch it -- role 0..3 = v0, v1, a, b -- no atomics, deterministic.
struct NcPairs { int P; const int32_t *v; /* [P][4] = v0, v1, a, b */ };

__device__ __forceinline__ void cross3(const float a[3], const float b[3], float o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
// e = v1 - v0, a = va - v0, b = vb - v0;  n0 = e x a, m = b x e (= -(e x b)); returns cos, norms clamped at 1e-8
// the way torch.cosine_similarity clamps them
__device__ __forceinline__ float nc_pair(const float *__restrict__ xyz, const int32_t *__restrict__ q, float e[3], float a[3],
                                         float b[3], float n0[3], float m[3], float &l0, float &l1)
{
    const float *p0 = xyz + 3 * (size_t)q[0], *p1 = xyz + 3 * (size_t)q[1], *pa = xyz + 3 * (size_t)q[2], *pb = xyz + 3 * (size_t)q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { e[k] = p1[k] - p0[k]; a[k] = pa[k] - p0[k]; b[k] = pb[k] - p0[k]; }
    cross3(e, a, n0);
    cross3(b, e, m);
    l0 = fmaxf(sqrtf((n0[0] * n0[0] + n0[1] * n0[1]) + n0[2] * n0[2]), 1e-8f);
    l1 = fmaxf(sqrtf((m[0] * m[0] + m[1] * m[1]) + m[2] * m[2]), 1e-8f);
    return ((n0[0] * m[0] + n0[1] * m[1]) + n0[2] * m[2]) / (l0 * l1);
}

__global__ __launch_bounds__(256) void k_nc_fwd(NcPairs pr, int V, const float *__restrict__ xyz, float *__restrict__ terms /* [T][P] */)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= pr.P) return;
    const size_t t = blockIdx.y;
    float e[3], a[3], b[3], n0[3], m[3], l0, l1;
    const float c = nc_pair(xyz + t * V * 3, pr.v + 4 * (size_t)p, e, a, b, n0, m, l0, l1);
    terms[t * pr.P + p] = 1.0f - c;
}

// Backward in two launches: the four vertices' gradient vectors of every pair (one thread per pair: the pair's normals, norms
// and cosine are evaluated ONCE, not once per vertex that gathers them -- with their correctly rounded divisions and square roots
// they were a 48-64 us kernel beside a 5 us forward), then the gather below.
__global__ __launch_bounds__(256) void k_nc_bwd_pairs(NcPairs pr, int V, const float *__restrict__ xyz, float *__restrict__ roles /* [T][P][4][3] */)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= pr.P) return;
    const size_t t = blockIdx.y;
    float e[3], a[3], b[3], n0[3], m[3], l0, l1;
    const float c = nc_pair(xyz + t * V * 3, pr.v + 4 * (size_t)p, e, a, b, n0, m, l0, l1);
    // d(1 - c)/dn0 = -(m / (l0 l1) - c n0 / l0^2),  d(1 - c)/dm = -(n0 / (l0 l1) - c m / l1^2)
    float g0[3], g1[3];
    const float r01 = 1.0f / (l0 * l1), c00 = c / (l0 * l0), c11 = c / (l1 * l1);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        g0[d] = -(m[d] * r01 - c00 * n0[d]);
        g1[d] = -(n0[d] * r01 - c11 * m[d]);
    }
    // n0 = e x a: d/de = a x g0, d/da = g0 x e;   m = b x e: d/db = e x g1, d/de = g1 x b
    float dE0[3], dE1[3], dA[3], dB[3];
    cross3(a, g0, dE0);
    cross3(g1, b, dE1);
    cross3(g0, e, dA);
    cross3(e, g1, dB);
    float4 *o = reinterpret_cast<float4 *>(roles + (t * pr.P + p) * 12);
    float r[12];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float dE = dE0[d] + dE1[d];
        r[d] = -((dE + dA[d]) + dB[d]);      // role 0: v0
        r[3 + d] = dE;                        // role 1: v1
        r[6 + d] = dA[d];                     // role 2: a
        r[9 + d] = dB[d];                     // role 3: b
    }
    o[0] = make_float4(r[0], r[1], r[2], r[3]);
    o[1] = make_float4(r[4], r[5], r[6], r[7]);
    o[2] = make_float4(r[8], r[9], r[10], r[11]);
}

// items of vertex i: item = pair * 4 + role, CSR offsets [V+1]; a group of kSub lanes per (mesh, vertex), see k_arap_fwd
__global__ __launch_bounds__(256) void k_nc_bwd(NcPairs pr, int V, const int32_t *__restrict__ off, const int32_t *__restrict__ items,
                                                const float *__restrict__ roles, const float *__restrict__ g_loss /* [T] */,
                                                float *__restrict__ g_xyz)
{
    const int gid = blockIdx.x * 256 + threadIdx.x, i = gid / kSub, sub = gid % kSub;
    if (i >= V) return;
    const size_t t = blockIdx.y;
    const float *rt = roles + t * pr.P * 12;
    float acc[3] = {0.f, 0.f, 0.f};
    const int k1 = off[i + 1];
#pragma unroll 4
    for (int k = off[i] + sub; k < k1; k += kSub) {
        const float *r = rt + 3 * (size_t)items[k];      // (item = pair * 4 + role: the role's vector of the pair)
        acc[0] += r[0]; acc[1] += r[1]; acc[2] += r[2];
    }
    group_sum(acc);
    if (sub != 0) return;
    const float s = g_loss[t] / (float)pr.P;
#pragma unroll
    for (int d = 0; d < 3; ++d) g_xyz[(t * V + i) * 3 + d] = acc[d] * s;
}


// ---------------------------------------------------------------------------------------- unit quaternion -> rotation matrix
// get_timed_vertex_rotation(return_matrix=True) (dynamic_sugar.py:640-655: a pypose .matrix()) for the ARAP term: R [n][3][3] of
// q [n][4] = (x, y, z, w), the arithmetic of the torch expression it replaces (ops.py::_MatrixPypose: ~45 elementwise launches
// per iteration for 9 polynomials), and pypose's backward: dL/dq = (sum_i (R e_i) x G[:, i], 0).
__global__ __launch_bounds__(256) void k_quat_matrix_fwd(size_t n, const float *__restrict__ q, float *__restrict__ R)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 v = reinterpret_cast<const float4 *>(q)[i];
    const float x = v.x, y = v.y, z = v.z, w = v.w;
    float *o = R + 9 * i;
    o[0] = 1.0f - 2.0f * (y * y + z * z); o[1] = 2.0f * (x * y - z * w);        o[2] = 2.0f * (x * z + y * w);
    o[3] = 2.0f * (x * y + z * w);        o[4] = 1.0f - 2.0f * (x * x + z * z); o[5] = 2.0f * (y * z - x * w);
    o[6] = 2.0f * (x * z - y * w);        o[7] = 2.0f * (y * z + x * w);        o[8] = 1.0f - 2.0f * (x * x + y * y);
}

__global__ __launch_bounds__(256) void k_quat_matrix_bwd(size_t n, const float *__restrict__ R, const float *__restrict__ G, float *__restrict__ gq)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *r = R + 9 * i, *g = G + 9 * i;
    float t[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {          // column c of R x column c of G, summed over the columns in order
        const float a[3] = {r[c], r[3 + c], r[6 + c]}, b[3] = {g[c], g[3 + c], g[6 + c]};
        float o[3];
        cross3(a, b, o);
#pragma unroll
        for (int d = 0; d < 3; ++d) t[d] = c == 0 ? o[d] : t[d] + o[d];
    }
    reinterpret_cast<float4 *>(gq)[i] = make_float4(t[0], t[1], t[2], 0.f);
}


// ---------------------------------------------------------------------------------------- Laplacian smoothing
// pytorch3d.loss.mesh_laplacian_smoothing(meshes, "uniform") (static stage: lambda 1, configs/sugar_static_refine.yaml:122;
// system/sugar_static.py:246-254; dynamic stage: system/sugar_4dgen.py:227-230, lambda 0 as shipped):
//   term_i = || (1 / deg_i) sum_{j in N(i)} v_j - v_i ||,  loss_t = mean_i term_i  (pytorch3d: then mean over meshes).
// Forward: one thread per (mesh, vertex), keeps the unit vector u_i of the Laplacian for the backward.  Backward
// (gather over the symmetric one-ring): d loss_t / d v_i = (-u_i + sum_{j in N(i)} u_j / deg_j) / V.
__global__ __launch_bounds__(256) void k_lap_fwd(int V, const int32_t *__restrict__ off, const int32_t *__restrict__ nbr,
                                                 const float *__restrict__ xyz, float *__restrict__ terms, float *__restrict__ unit)
{
    // a group of kSub lanes per (mesh, vertex), as in k_arap_fwd (one thread per vertex: 63 / 84 us forward / backward at 8.3k
    // vertices -- the kernel was as long as the highest valence's chain of dependent loads)
    const int gid = blockIdx.x * 256 + threadIdx.x, i = gid / kSub, sub = gid % kSub;
    if (i >= V) return;
    const size_t t = blockIdx.y;
    const float *x = xyz + t * V * 3;
    float s[3] = {0.f, 0.f, 0.f};
    const int e0 = off[i], e1 = off[i + 1];
#pragma unroll 4
    for (int e = e0 + sub; e < e1; e += kSub) {
        const float *xj = x + 3 * (size_t)nbr[e];
        s[0] += xj[0]; s[1] += xj[1]; s[2] += xj[2];
    }
    group_sum(s);
    if (sub != 0) return;
    const float inv = e1 > e0 ? 1.0f / (float)(e1 - e0) : 0.f;
    float d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) d[k] = e1 > e0 ? s[k] * inv - x[3 * (size_t)i + k] : 0.f;     // isolated vertex: row of zeros
    const float n = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    terms[t * V + i] = n;
    const float r = n > 0.f ? 1.0f / n : 0.f;                                                 // norm's subgradient at 0
#pragma unroll
    for (int k = 0; k < 3; ++k) unit[(t * V + i) * 3 + k] = d[k] * r;
}

__global__ __launch_bounds__(256) void k_lap_bwd(int V, const int32_t *__restrict__ off, const int32_t *__restrict__ nbr,
                                                 const float *__restrict__ unit, const float *__restrict__ g_loss, float *__restrict__ g_xyz)
{
    const int gid = blockIdx.x * 256 + threadIdx.x, i = gid / kSub, sub = gid % kSub;
    if (i >= V) return;
    const size_t t = blockIdx.y;
    const float *u = unit + t * V * 3;
    float a[3] = {0.f, 0.f, 0.f};
    const int e1 = off[i + 1];
#pragma unroll 4
    for (int e = off[i] + sub; e < e1; e += kSub) {
        const int j = nbr[e];
        const float w = 1.0f / (float)(off[j + 1] - off[j]);
        a[0] += u[3 * (size_t)j] * w; a[1] += u[3 * (size_t)j + 1] * w; a[2] += u[3 * (size_t)j + 2] * w;
    }
    group_sum(a);
    if (sub != 0) return;
    const float s = g_loss[t] / (float)V;
#pragma unroll
    for (int k = 0; k < 3; ++k) g_xyz[(t * V + i) * 3 + k] = (a[k] - u[3 * (size_t)i + k]) * s;
}

static int arap_check(int T, int V, const void *off, const void *nbr, const void *rev, const void *w, const void *e,
                      const void *xyz, const void *rot)
{
    if (T < 0 || V < 0) { set_error("arap: negative T/V"); return DM4D_ERR_INVALID; }
    if (T > 65535) { set_error("arap: more than 65535 timestamps per call"); return DM4D_ERR_INVALID; }
    if (T > 0 && V > 0 && (!off || !nbr || !rev || !w || !e || !xyz || !rot)) { set_error("arap: null tensor"); return DM4D_ERR_INVALID; }
    return DM4D_OK;
}

}  // namespace dm4d

using namespace dm4d;

extern "C" {

int dm4d_arap_energy_forward(int32_t T, int32_t V, const int32_t *csr_offsets, const int32_t *neighbors,
                             const int32_t *reverse_edge, const float *weights, const float *rest_edges,
                             const float *xyz_prime, const float *rotations, float *vertex_energy, dm4d_stream_t stream)
{
    int rc = arap_check(T, V, csr_offsets, neighbors, reverse_edge, weights, rest_edges, xyz_prime, rotations);
    if (rc) return rc;
    if (T == 0 || V == 0) return DM4D_OK;
    if (!vertex_energy) { set_error("arap: null output"); return DM4D_ERR_INVALID; }
    ArapAdj a{V, csr_offsets, neighbors, reverse_edge, weights, rest_edges};
    hipLaunchKernelGGL(k_arap_fwd, dim3((unsigned)(((size_t)V * kSub + 255) / 256), T), dim3(256), 0, (hipStream_t)stream, a, xyz_prime, rotations, vertex_energy);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_arap_energy_backward(int32_t T, int32_t V, const int32_t *csr_offsets, const int32_t *neighbors,
                              const int32_t *reverse_edge, const float *weights, const float *rest_edges,
                              const float *xyz_prime, const float *rotations, const float *g_energy, float *g_xyz,
                              float *g_rotations, dm4d_stream_t stream)
{
    int rc = arap_check(T, V, csr_offsets, neighbors, reverse_edge, weights, rest_edges, xyz_prime, rotations);
    if (rc) return rc;
    if (T == 0 || V == 0) return DM4D_OK;
    if (!g_energy) { set_error("arap: null upstream gradient"); return DM4D_ERR_INVALID; }
    ArapAdj a{V, csr_offsets, neighbors, reverse_edge, weights, rest_edges};
    hipLaunchKernelGGL(k_arap_bwd, dim3((unsigned)(((size_t)V * kSub + 255) / 256), T), dim3(256), 0, (hipStream_t)stream, a, xyz_prime, rotations, g_energy,
                       g_xyz, g_rotations);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_normal_consistency_forward(int32_t T, int32_t V, int32_t P, const int32_t *pairs, const float *xyz, float *terms,
                                    dm4d_stream_t stream)
{
    if (T < 0 || V < 0 || P < 0) { set_error("normal consistency: negative size"); return DM4D_ERR_INVALID; }
    if (T == 0 || P == 0) return DM4D_OK;
    if (!pairs || !xyz || !terms) { set_error("normal consistency: null tensor"); return DM4D_ERR_INVALID; }
    NcPairs pr{P, pairs};
    hipLaunchKernelGGL(k_nc_fwd, dim3((P + 255) / 256, T), dim3(256), 0, (hipStream_t)stream, pr, V, xyz, terms);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_normal_consistency_backward(int32_t T, int32_t V, int32_t P, const int32_t *pairs, const int32_t *vert_offsets,
                                     const int32_t *vert_items, const float *xyz, const float *g_loss, float *g_xyz,
                                     dm4d_stream_t stream)
{
    // rounds 1-3's signature: the scratch is the library's (see include/dm4d.h) -- one per DEVICE (a pointer of another device is not
    // addressable), guarded by a mutex (two host threads), never grown while the stream is capturing (hipFree / hipMalloc are illegal
    // there; callers that capture use the _scratch form), and grown only after the device has drained (another stream may still
    // read the old block)
    static std::mutex mu;
    static float *own_dev[64] = {};
    static size_t own_floats_dev[64] = {};
    int dev = 0;
    DM4D_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { set_error("normal consistency: device index %d", dev); return DM4D_ERR_UNSUPPORTED; }
    const size_t need = (T > 0 && P > 0) ? (size_t)T * (size_t)P * 12 : 0;
    std::lock_guard<std::mutex> lock(mu);
    float *&own = own_dev[dev];
    size_t &own_floats = own_floats_dev[dev];
    if (need > own_floats) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
            set_error("normal consistency backward: the library-owned scratch cannot grow during stream capture; call "
                      "dm4d_normal_consistency_backward_scratch with a caller-owned scratch");
            return DM4D_ERR_UNSUPPORTED;
        }
        if (own) { DM4D_HIP_CHECK(hipDeviceSynchronize()); DM4D_HIP_CHECK(hipFree(own)); own = nullptr; own_floats = 0; }
        DM4D_HIP_CHECK(hipMalloc((void **)&own, need * sizeof(float)));
        own_floats = need;
    }
    return dm4d_normal_consistency_backward_scratch(T, V, P, pairs, vert_offsets, vert_items, xyz, g_loss, g_xyz, own, stream);
}

int dm4d_normal_consistency_backward_scratch(int32_t T, int32_t V, int32_t P, const int32_t *pairs, const int32_t *vert_offsets,
                                             const int32_t *vert_items, const float *xyz, const float *g_loss, float *g_xyz,
                                             float *scratch, dm4d_stream_t stream)
{
    if (T < 0 || V < 0 || P < 0) { set_error("normal consistency: negative size"); return DM4D_ERR_INVALID; }
    if (T == 0 || V == 0) return DM4D_OK;
    if (!pairs || !vert_offsets || !vert_items || !xyz || !g_loss || !g_xyz) { set_error("normal consistency: null tensor"); return DM4D_ERR_INVALID; }
    if (P > 0 && (!scratch || ((uintptr_t)scratch & 15) != 0)) { set_error("normal consistency: the backward needs a 16-byte aligned scratch of T P 12 floats"); return DM4D_ERR_INVALID; }
    NcPairs pr{P > 0 ? P : 1, pairs};
    if (P > 0) {
        hipLaunchKernelGGL(k_nc_bwd_pairs, dim3((P + 255) / 256, T), dim3(256), 0, (hipStream_t)stream, pr, V, xyz, scratch);
        DM4D_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_nc_bwd, dim3((unsigned)(((size_t)V * kSub + 255) / 256), T), dim3(256), 0, (hipStream_t)stream, pr, V, vert_offsets, vert_items, scratch,
                       g_loss, g_xyz);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_quat_to_matrix_forward(int64_t n, const float *quat_xyzw, float *matrices, dm4d_stream_t stream)
{
    if (n < 0) { set_error("quat_to_matrix: negative size"); return DM4D_ERR_INVALID; }
    if (n == 0) return DM4D_OK;
    if (!quat_xyzw || !matrices) { set_error("quat_to_matrix: null tensor"); return DM4D_ERR_INVALID; }
    if (((uintptr_t)quat_xyzw & 15) != 0) { set_error("quat_to_matrix: quaternions must be 16-byte aligned"); return DM4D_ERR_INVALID; }
    hipLaunchKernelGGL(k_quat_matrix_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (size_t)n, quat_xyzw, matrices);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_quat_to_matrix_backward_pypose(int64_t n, const float *matrices, const float *g_matrices, float *g_quat, dm4d_stream_t stream)
{
    if (n < 0) { set_error("quat_to_matrix: negative size"); return DM4D_ERR_INVALID; }
    if (n == 0) return DM4D_OK;
    if (!matrices || !g_matrices || !g_quat) { set_error("quat_to_matrix: null tensor"); return DM4D_ERR_INVALID; }
    if (((uintptr_t)g_quat & 15) != 0) { set_error("quat_to_matrix: the quaternion gradient must be 16-byte aligned"); return DM4D_ERR_INVALID; }
    hipLaunchKernelGGL(k_quat_matrix_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (size_t)n, matrices, g_matrices, g_quat);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_laplacian_smoothing_forward(int32_t T, int32_t V, const int32_t *csr_offsets, const int32_t *neighbors, const float *xyz,
                                     float *terms, float *unit, dm4d_stream_t stream)
{
    if (T < 0 || V < 0) { set_error("laplacian smoothing: negative size"); return DM4D_ERR_INVALID; }
    if (T == 0 || V == 0) return DM4D_OK;
    if (!csr_offsets || !neighbors || !xyz || !terms || !unit) { set_error("laplacian smoothing: null tensor"); return DM4D_ERR_INVALID; }
    hipLaunchKernelGGL(k_lap_fwd, dim3((unsigned)(((size_t)V * kSub + 255) / 256), T), dim3(256), 0, (hipStream_t)stream, V, csr_offsets, neighbors, xyz, terms, unit);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_laplacian_smoothing_backward(int32_t T, int32_t V, const int32_t *csr_offsets, const int32_t *neighbors, const float *unit,
                                      const float *g_loss, float *g_xyz, dm4d_stream_t stream)
{
    if (T < 0 || V < 0) { set_error("laplacian smoothing: negative size"); return DM4D_ERR_INVALID; }
    if (T == 0 || V == 0) return DM4D_OK;
    if (!csr_offsets || !neighbors || !unit || !g_loss || !g_xyz) { set_error("laplacian smoothing: null tensor"); return DM4D_ERR_INVALID; }
    hipLaunchKernelGGL(k_lap_bwd, dim3((unsigned)(((size_t)V * kSub + 255) / 256), T), dim3(256), 0, (hipStream_t)stream, V, csr_offsets, neighbors, unit, g_loss, g_xyz);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // extern "C"
