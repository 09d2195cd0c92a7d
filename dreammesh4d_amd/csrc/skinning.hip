// skinning.hip -- sparse-control skinning of mesh vertices and the face -> Gaussian transform
// (gfx950).  Replaces, per (frame, view) unit, the pypose / pytorch3d / fancy-index graph of
//   custom/threestudio-dreammesh4d/geometry/dynamic_sugar.py:408-465  (node attributes)
//   .../dynamic_sugar.py:487-613   (_get_timed_vertex_attributes_from_dg: LBS / DQS / hybrid,
//                                   vertex rotation Exp(sum_k w_k Log q_k))
//   .../utils/dual_quaternions.py:94-131,184-197,224-231
//   .../dynamic_sugar.py:657-676,726-743,877-889 (Gaussian means, fused rotations)
//   .../dynamic_sugar.py:330-364   (deformed face normals, one per Gaussian)
// which materialises [V,K,3,3] tensors, two bmm's and ~40 elementwise launches per call, with
// SIX kernels: forward = 1 thread/vertex + 1 thread/Gaussian; backward = gather formulations
// over static adjacency (face->corner records -> vertex, vertex->neighbour records -> node), so
// there are no atomics and the gradients are bit-reproducible.
//
// Everything is HBM-bound gather work on a few hundred KB: vertex and node tables stay in L2.
// Gradients, two conventions (DESIGN.md "gradient convention"):
//   exact   the Euclidean derivatives of the forward function;
//   pypose  (DM4D_GRAD_PYPOSE or-ed into `method` / `G`) what the reference's autograd returns: pypose LieTensor
//           operations (SO3 Act / Log / Mul, so3 Exp) hand back LEFT-PERTURBATION tangent gradients zero-padded into the
//           quaternion storage, and take the first three components of an incoming quaternion-storage gradient as such
//           a tangent gradient; torch ops in between (F.normalize, .tensor(), the dual-quaternion algebra) stay Euclidean.
//           Restated from pypose 0.6.7's published backward rules (pypose/lietensor/operation.py: SO3_Log, so3_Exp,
//           SO3_Act, SO3_Mul; so3_Jl / so3_Jl_inv); the package is not in the tree: PARITY UNPINNED.
#include "common.h"
#include "raster.h"

namespace dm4d {

struct q4 { float x, y, z, w; };   // (x, y, z, w) storage, as pypose SO3
struct v3 { float x, y, z; };

__device__ __forceinline__ v3 mk3(float x, float y, float z) { return v3{x, y, z}; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ v3 operator*(float s, v3 a) { return v3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ v3 cross(v3 a, v3 b) { return v3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ v3 ld3(const float *p, size_t i) { return v3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
__device__ __forceinline__ void st3(float *p, size_t i, v3 a) { p[3 * i] = a.x; p[3 * i + 1] = a.y; p[3 * i + 2] = a.z; }
__device__ __forceinline__ q4 ldq(const float *p, size_t i) { const float4 t = reinterpret_cast<const float4 *>(p)[i]; return q4{t.x, t.y, t.z, t.w}; }
__device__ __forceinline__ v3 qv(q4 q) { return v3{q.x, q.y, q.z}; }
__device__ __forceinline__ float qdot(q4 a, q4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ q4 qscale(float s, q4 a) { return q4{s * a.x, s * a.y, s * a.z, s * a.w}; }
__device__ __forceinline__ q4 qadd(q4 a, q4 b) { return q4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__device__ __forceinline__ q4 qconj(q4 a) { return q4{-a.x, -a.y, -a.z, a.w}; }
// Hamilton product
__device__ __forceinline__ q4 qmul(q4 a, q4 b)
{
    return q4{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
              a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
// R(q) p = p + 2 w (v x p) + 2 v x (v x p)   (pypose SO3 Act)
__device__ __forceinline__ v3 qact(q4 q, v3 p)
{
    const v3 v = qv(q);
    const v3 uv = 2.f * cross(v, p);
    return p + q.w * uv + cross(v, uv);
}
// gradient of h . qact(q, y) w.r.t. q (as a free 4-vector) and y
__device__ __forceinline__ q4 qact_grad_q(q4 q, v3 y, v3 h)
{
    const v3 v = qv(q);
    const v3 gv = (2.f * q.w) * cross(y, h) + 2.f * (dot(v, y) * h + dot(h, v) * y - (2.f * dot(h, y)) * v);
    return q4{gv.x, gv.y, gv.z, 2.f * dot(h, cross(v, y))};
}
__device__ __forceinline__ v3 qact_grad_p(q4 q, v3 h)   // R(q)^T h for unit q; exact transpose in general
{
    // d/dp [p + 2w (v x p) + 2 v x (v x p)]^T h = h - 2w (v x h) + 2 v x (v x h)
    const v3 v = qv(q);
    const v3 uv = 2.f * cross(v, h);
    return h - q.w * uv + cross(v, uv);
}

constexpr float kEps = 1.1920928955078125e-07f;   // float32 eps: pypose's series/generic branch threshold

// SO3 Log (quaternion -> rotation vector): 2 atan(|v|/w)/|v| * v
__device__ __forceinline__ v3 so3_log(q4 q)
{
    const v3 v = qv(q);
    const float u = sqrtf(dot(v, v));
    const float f = (u < kEps) ? (2.f / q.w - (2.f / 3.f) * u * u / (q.w * q.w * q.w)) : 2.f * atanf(u / q.w) / u;
    return f * v;
}
// given g = dL/dLog, returns dL/dq (free 4-vector)
__device__ __forceinline__ q4 so3_log_grad(q4 q, v3 g)
{
    const v3 v = qv(q);
    const float u2 = dot(v, v), u = sqrtf(u2), w = q.w;
    float f, dfu_over_u;
    if (u < 1e-4f) {
        f = 2.f / w - (2.f / 3.f) * u2 / (w * w * w);
        dfu_over_u = -4.f / (3.f * w * w * w);
    } else {
        const float at = atanf(u / w);
        f = 2.f * at / u;
        dfu_over_u = 2.f * (w * u / (u2 + w * w) - at) / (u2 * u);
    }
    const float vg = dot(v, g);
    const v3 gv = f * g + (dfu_over_u * vg) * v;
    return q4{gv.x, gv.y, gv.z, -2.f / (u2 + w * w) * vg};
}
// so3 Exp (rotation vector -> quaternion)
__device__ __forceinline__ q4 so3_exp(v3 r)
{
    const float t2 = dot(r, r), t = sqrtf(t2);
    float a, c;
    if (t < kEps) { a = 0.5f - t2 / 48.f + t2 * t2 / 3840.f; c = 1.f - t2 / 8.f + t2 * t2 / 384.f; }
    else { a = sinf(0.5f * t) / t; c = cosf(0.5f * t); }
    return q4{a * r.x, a * r.y, a * r.z, c};
}
// given g = dL/dq (4-vector) at q = Exp(r), returns dL/dr
__device__ __forceinline__ v3 so3_exp_grad(v3 r, q4 g)
{
    const float t2 = dot(r, r), t = sqrtf(t2);
    float a, da_over_t;
    if (t < 1e-3f) { a = 0.5f - t2 / 48.f; da_over_t = -1.f / 24.f + t2 / 960.f; }
    else {
        const float s = sinf(0.5f * t), c = cosf(0.5f * t);
        a = s / t;
        da_over_t = (0.5f * t * c - s) / (t2 * t);
    }
    const v3 gx = qv(g);
    return a * gx + (da_over_t * dot(r, gx) - 0.5f * a * g.w) * r;
}

// ---- pypose's convention -------------------------------------------------------------------------------------------
// g Jl(x) and g Jl^-1(x) (row vector times the left Jacobian of SO(3) / its inverse), K = hat(x):
//   Jl = I + (1 - cos t)/t^2 K + (t - sin t)/t^3 K^2      Jl^-1 = I - K/2 + (1/t^2 - (1 + cos t)/(2 t sin t)) K^2
//   g K = g x x,  g K^2 = (g x x) x x
__device__ __forceinline__ v3 row_times_Jl(v3 x, v3 g)
{
    const float t2 = dot(x, x), t = sqrtf(t2);
    float c1, c2;
    if (t < 1e-3f) { c1 = 0.5f - t2 / 24.f; c2 = 1.f / 6.f - t2 / 120.f; }
    else { c1 = (1.f - cosf(t)) / t2; c2 = (t - sinf(t)) / (t2 * t); }
    const v3 gk = cross(g, x), gkk = cross(gk, x);
    return g + c1 * gk + c2 * gkk;
}
__device__ __forceinline__ v3 row_times_Jl_inv(v3 x, v3 g)
{
    const float t2 = dot(x, x), t = sqrtf(t2);
    float c2;
    if (t < 1e-3f) c2 = 1.f / 12.f + t2 / 720.f;
    else c2 = (1.f - 0.5f * t * (1.f + cosf(t)) / sinf(t)) / t2;
    const v3 gk = cross(g, x), gkk = cross(gk, x);
    return g - 0.5f * gk + c2 * gkk;
}
// SO3_Log.backward: (g Jl^-1(Log q), 0)
__device__ __forceinline__ q4 so3_log_grad_pp(q4 q, v3 g)
{
    const v3 r = row_times_Jl_inv(so3_log(q), g);
    return q4{r.x, r.y, r.z, 0.f};
}
// so3_Exp.backward: g[:3] Jl(r)
__device__ __forceinline__ v3 so3_exp_grad_pp(v3 r, q4 g) { return row_times_Jl(r, qv(g)); }
// SO3_Act.backward w.r.t. the rotation: (h (-hat(R y)), 0) = ((R y) x h, 0)
__device__ __forceinline__ q4 qact_grad_q_pp(q4 q, v3 y, v3 h)
{
    const v3 r = cross(qact(q, y), h);
    return q4{r.x, r.y, r.z, 0.f};
}
constexpr int kPypose = 0x100;     // == DM4D_GRAD_PYPOSE
// Gradient of the quaternion product Z = A * B of the dual-quaternion algebra (dual_quaternions.py:115-131: q_d = (t / 2) * q_r;
// :224-231: translation = (2 q_d) * conj(q_r)) given g = dL/dZ.  exact: gA = g * conj(B), gB = conj(A) * g.  pypose: the operands
// are pp.SO3 LieTensors (of NON-unit quaternions) and `*` is SO3_Mul, whose backward hands back
//     X_grad = (g[:3], 0)        Y_grad = (g[:3] SO3_Adj(X), 0),   SO3_Adj(X) = I + 2 w hat(v) + 2 hat(v)^2 of X's components as they are
// (pypose 0.6.7, lietensor/operation.py; the row vector times that matrix is qact(conj(X), g[:3]), also for a non-unit X).
__device__ __forceinline__ void dqs_mul_grad(const int pypose, const q4 A, const q4 B, const q4 g, q4 &gA, q4 &gB)
{
    if (pypose) {
        gA = q4{g.x, g.y, g.z, 0.f};
        const v3 r = qact(qconj(A), qv(g));
        gB = q4{r.x, r.y, r.z, 0.f};
    } else {
        gA = qmul(g, qconj(B));
        gB = qmul(qconj(A), g);
    }
}

// ---- node attributes from the raw deformation-network outputs --------------------------------
struct NodeAttr { q4 q; float pn; v3 t; float S[9]; float o; };
__device__ __forceinline__ NodeAttr node_attr(int m, const float *dx, const float *dr, const float *ds, const float *dop)
{
    NodeAttr a;
    a.t = ld3(dx, m);
    const q4 p = ldq(dr, m);
    const q4 pp = q4{p.x, p.y, p.z, p.w + 1.f};            // + identity (dynamic_sugar.py:449-451)
    a.pn = fmaxf(sqrtf(qdot(pp, pp)), 1e-12f);             // F.normalize eps
    a.q = qscale(1.f / a.pn, pp);
    if (ds) {
        const float *s = ds + 6 * (size_t)m;
        a.S[0] = 1.f + s[0]; a.S[1] = s[3]; a.S[2] = s[4];
        a.S[3] = s[3]; a.S[4] = 1.f + s[1]; a.S[5] = s[5];
        a.S[6] = s[4]; a.S[7] = s[5]; a.S[8] = 1.f + s[2];
    } else {
        a.S[0] = a.S[4] = a.S[8] = 1.f;
        a.S[1] = a.S[2] = a.S[3] = a.S[5] = a.S[6] = a.S[7] = 0.f;
    }
    a.o = dop ? 1.f / (1.f + __expf(-dop[m])) : 0.f;
    return a;
}

constexpr int kSkinThreads = 256;
struct ZeroList { char *base; size_t stride; int n; };     // n blocks of 64 words at base + i * stride to clear (k_face_fwd)
enum { kLbs = 0, kDqs = 1, kHybrid = 2 };
constexpr int kMaxK = 8;
constexpr int kNodeRec = 14;   // dx3 dr4 ds6 do1

struct SkinArgs {
    int method, V, M, K;
    int pypose;                          // gradient convention of the backward (see the file header)
    const float *verts; const int32_t *nbr_idx; const float *nbr_w;
    const float *dx, *dr, *ds, *dop;     // [B][M][3|4|6|1]: blockIdx.y selects the view
};
// per-view slices of the batched node tables
__device__ __forceinline__ SkinArgs skin_view(SkinArgs a, int b)
{
    const size_t o = (size_t)b * a.M;
    a.dx += o * 3;
    a.dr += o * 4;
    if (a.ds) a.ds += o * 6;
    if (a.dop) a.dop += o;
    return a;
}

// ---------------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(kSkinThreads) void k_skin_fwd(SkinArgs a0, float *__restrict__ out_xyz,
                                                           float *__restrict__ out_rot)
{
    const int v = blockIdx.x * kSkinThreads + threadIdx.x;
    if (v >= a0.V) return;
    const SkinArgs a = skin_view(a0, blockIdx.y);
    out_xyz += (size_t)blockIdx.y * a.V * 3;
    out_rot += (size_t)blockIdx.y * a.V * 4;
    const v3 p = ld3(a.verts, v);
    v3 x_lbs = mk3(0, 0, 0), rho = mk3(0, 0, 0);
    q4 br = q4{0, 0, 0, 0}, bd = q4{0, 0, 0, 0};
    float eta = 0.f;
    for (int k = 0; k < a.K; ++k) {
        const int m = a.nbr_idx[(size_t)v * a.K + k];
        const float w = a.nbr_w[(size_t)v * a.K + k];
        const NodeAttr n = node_attr(m, a.dx, a.dr, a.ds, a.dop);
        if (a.method != kDqs) {
            const v3 y = mk3(n.S[0] * p.x + n.S[1] * p.y + n.S[2] * p.z, n.S[3] * p.x + n.S[4] * p.y + n.S[5] * p.z,
                             n.S[6] * p.x + n.S[7] * p.y + n.S[8] * p.z);
            x_lbs = x_lbs + w * (qact(n.q, y) + n.t);
        }
        if (a.method != kLbs) {
            const float qn = sqrtf(qdot(n.q, n.q));
            const q4 qr = qscale(1.f / qn, n.q);
            const q4 d = qmul(q4{0.5f * n.t.x, 0.5f * n.t.y, 0.5f * n.t.z, 0.f}, qr);
            br = qadd(br, qscale(w, qr));
            bd = qadd(bd, qscale(w, d));
        }
        eta += w * n.o;
        rho = rho + w * so3_log(n.q);
    }
    v3 x;
    if (a.method == kLbs) {
        x = x_lbs;
    } else {
        const float nn = sqrtf(qdot(br, br));
        const q4 rh = qscale(1.f / nn, br), dh = qscale(1.f / nn, bd);
        const q4 trq = qmul(qscale(2.f, dh), qconj(rh));
        const v3 x_dqs = qact(rh, p) + qv(trq);
        if (a.method == kDqs) x = x_dqs;
        else {
            const float e = fminf(eta + 0.4f, 1.0f);
            x = e * x_lbs + (1.f - e) * x_dqs;
        }
    }
    st3(out_xyz, v, x);
    const q4 vr = so3_exp(rho);
    reinterpret_cast<float4 *>(out_rot)[v] = make_float4(vr.x, vr.y, vr.z, vr.w);
}

// ---------------------------------------------------------------------------------------- backward 1
// per vertex: gradients w.r.t. the raw outputs of its K neighbour nodes -> rec[v][k][14]
__global__ __launch_bounds__(kSkinThreads) void k_skin_bwd_vertex(SkinArgs a0, const float *__restrict__ g_xyz,
                                                                  const float *__restrict__ g_rot,
                                                                  float *__restrict__ rec)
{
    const int v = blockIdx.x * kSkinThreads + threadIdx.x;
    if (v >= a0.V) return;
    const SkinArgs a = skin_view(a0, blockIdx.y);
    if (g_xyz) g_xyz += (size_t)blockIdx.y * a.V * 3;
    if (g_rot) g_rot += (size_t)blockIdx.y * a.V * 4;
    rec += (size_t)blockIdx.y * a.V * a.K * kNodeRec;
    const v3 p = ld3(a.verts, v);
    const v3 gx = g_xyz ? ld3(g_xyz, v) : mk3(0, 0, 0);
    const q4 gq = g_rot ? ldq(g_rot, v) : q4{0, 0, 0, 0};
    // ---- recompute the forward blend state ----
    v3 x_lbs = mk3(0, 0, 0), rho = mk3(0, 0, 0);
    q4 br = q4{0, 0, 0, 0}, bd = q4{0, 0, 0, 0};
    float eta_raw = 0.f;
    for (int k = 0; k < a.K; ++k) {
        const int m = a.nbr_idx[(size_t)v * a.K + k];
        const float w = a.nbr_w[(size_t)v * a.K + k];
        const NodeAttr n = node_attr(m, a.dx, a.dr, a.ds, a.dop);
        if (a.method != kDqs) {
            const v3 y = mk3(n.S[0] * p.x + n.S[1] * p.y + n.S[2] * p.z, n.S[3] * p.x + n.S[4] * p.y + n.S[5] * p.z,
                             n.S[6] * p.x + n.S[7] * p.y + n.S[8] * p.z);
            x_lbs = x_lbs + w * (qact(n.q, y) + n.t);
        }
        if (a.method != kLbs) {
            const float qn = sqrtf(qdot(n.q, n.q));
            const q4 qr = qscale(1.f / qn, n.q);
            br = qadd(br, qscale(w, qr));
            bd = qadd(bd, qscale(w, qmul(q4{0.5f * n.t.x, 0.5f * n.t.y, 0.5f * n.t.z, 0.f}, qr)));
        }
        eta_raw += w * n.o;
        rho = rho + w * so3_log(n.q);
    }
    // ---- gradients of the blended quantities ----
    float eta = 1.f, g_eta = 0.f;
    v3 g_lbs = gx, g_dqs = mk3(0, 0, 0);
    q4 g_br = q4{0, 0, 0, 0}, g_bd = q4{0, 0, 0, 0};
    if (a.method != kLbs) {
        const float nn = sqrtf(qdot(br, br));
        const q4 rh = qscale(1.f / nn, br), dh = qscale(1.f / nn, bd);
        if (a.method == kDqs) { g_dqs = gx; g_lbs = mk3(0, 0, 0); }
        else {
            const v3 x_dqs = qact(rh, p) + qv(qmul(qscale(2.f, dh), qconj(rh)));
            eta = fminf(eta_raw + 0.4f, 1.0f);
            g_lbs = eta * gx;
            g_dqs = (1.f - eta) * gx;
            g_eta = (eta_raw + 0.4f < 1.0f) ? dot(gx, x_lbs - x_dqs) : 0.f;
        }
        // x_dqs = R(rh) p + (2 dh * conj(rh)).xyz
        q4 g_rh = a.pypose ? qact_grad_q_pp(rh, p, g_dqs) : qact_grad_q(rh, p, g_dqs);     // q_r.matrix() of transform_point_simple
        const q4 G = q4{g_dqs.x, g_dqs.y, g_dqs.z, 0.f};
        q4 g_dh, g_b;                                                // translation = (2 dh) * conj(rh): a = 2 dh, b = conj(rh)
        dqs_mul_grad(a.pypose, qscale(2.f, dh), qconj(rh), G, g_dh, g_b);
        g_dh = qscale(2.f, g_dh);
        g_rh = qadd(g_rh, q4{-g_b.x, -g_b.y, -g_b.z, g_b.w});
        g_bd = qscale(1.f / nn, g_dh);
        g_br = qadd(qscale(1.f / nn, g_rh), qscale(-(qdot(g_rh, rh) + qdot(g_dh, dh)) / nn, rh));
    }
    const v3 g_rho = a.pypose ? so3_exp_grad_pp(rho, gq) : so3_exp_grad(rho, gq);
    // ---- per-neighbour gradients ----
    for (int k = 0; k < a.K; ++k) {
        const int m = a.nbr_idx[(size_t)v * a.K + k];
        const float w = a.nbr_w[(size_t)v * a.K + k];
        const NodeAttr n = node_attr(m, a.dx, a.dr, a.ds, a.dop);
        v3 g_t = mk3(0, 0, 0);
        q4 g_q = a.pypose ? so3_log_grad_pp(n.q, w * g_rho) : so3_log_grad(n.q, w * g_rho);
        float g_S[6] = {0, 0, 0, 0, 0, 0};
        if (a.method != kDqs) {
            const v3 h = w * g_lbs;
            const v3 y = mk3(n.S[0] * p.x + n.S[1] * p.y + n.S[2] * p.z, n.S[3] * p.x + n.S[4] * p.y + n.S[5] * p.z,
                             n.S[6] * p.x + n.S[7] * p.y + n.S[8] * p.z);
            g_t = g_t + h;
            g_q = qadd(g_q, a.pypose ? qact_grad_q_pp(n.q, y, h) : qact_grad_q(n.q, y, h));
            const v3 gy = qact_grad_p(n.q, h);
            g_S[0] = gy.x * p.x; g_S[1] = gy.y * p.y; g_S[2] = gy.z * p.z;
            g_S[3] = gy.x * p.y + gy.y * p.x;
            g_S[4] = gy.x * p.z + gy.z * p.x;
            g_S[5] = gy.y * p.z + gy.z * p.y;
        }
        if (a.method != kLbs) {
            const float qn = sqrtf(qdot(n.q, n.q));
            const q4 qr = qscale(1.f / qn, n.q);
            const q4 av = q4{0.5f * n.t.x, 0.5f * n.t.y, 0.5f * n.t.z, 0.f};
            const q4 g_d = qscale(w, g_bd);
            q4 g_a, g_qr;                                               // q_d = (t / 2) * q_r
            dqs_mul_grad(a.pypose, av, qr, g_d, g_a, g_qr);
            g_t = g_t + 0.5f * qv(g_a);
            g_qr = qadd(g_qr, qscale(w, g_br));
            g_q = qadd(g_q, qscale(1.f / qn, qadd(g_qr, qscale(-qdot(g_qr, qr), qr))));
        }
        // through q = pp / |pp|
        const q4 g_p = qscale(1.f / n.pn, qadd(g_q, qscale(-qdot(g_q, n.q), n.q)));
        const float g_o = w * g_eta;
        float *r = rec + ((size_t)v * a.K + k) * kNodeRec;
        r[0] = g_t.x; r[1] = g_t.y; r[2] = g_t.z;
        r[3] = g_p.x; r[4] = g_p.y; r[5] = g_p.z; r[6] = g_p.w;
        r[7] = g_S[0]; r[8] = g_S[1]; r[9] = g_S[2]; r[10] = g_S[3]; r[11] = g_S[4]; r[12] = g_S[5];
        r[13] = g_o * n.o * (1.f - n.o);
    }
}

// K == 4 (the shipped configuration): the four neighbours of a vertex on the four lanes of a DPP quad.  The vertex
// kernel above is one long dependent chain per thread on V x frames = 67 k threads (one wave per SIMD: nothing hides
// its latency); here every lane evaluates ONE neighbour (its node attributes once, not once per phase), the blended
// quantities are quad sums ((k0 + k1) + (k2 + k3), fixed order), the vertex-level part is replicated and every lane
// writes its own record: 4x the threads, a third of the chain.
__device__ __forceinline__ float quad_sum(float x)
{
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    return x;
}
__device__ __forceinline__ v3 quad_sum3(v3 a) { return mk3(quad_sum(a.x), quad_sum(a.y), quad_sum(a.z)); }
__device__ __forceinline__ q4 quad_sum4(q4 a) { return q4{quad_sum(a.x), quad_sum(a.y), quad_sum(a.z), quad_sum(a.w)}; }

__global__ __launch_bounds__(kSkinThreads) void k_skin_bwd_vertex_k4(SkinArgs a0, const float *__restrict__ g_xyz,
                                                                     const float *__restrict__ g_rot,
                                                                     float *__restrict__ rec)
{
    const int gid = blockIdx.x * kSkinThreads + threadIdx.x;
    const int k = gid & 3;
    const bool live = (gid >> 2) < a0.V;
    const int v = live ? (gid >> 2) : a0.V - 1;       // whole quads stay active (DPP); dead quads do not store
    const SkinArgs a = skin_view(a0, blockIdx.y);
    if (g_xyz) g_xyz += (size_t)blockIdx.y * a.V * 3;
    if (g_rot) g_rot += (size_t)blockIdx.y * a.V * 4;
    rec += (size_t)blockIdx.y * a.V * 4 * kNodeRec;
    const v3 p = ld3(a.verts, v);
    const v3 gx = g_xyz ? ld3(g_xyz, v) : mk3(0, 0, 0);
    const q4 gq = g_rot ? ldq(g_rot, v) : q4{0, 0, 0, 0};
    const int m = a.nbr_idx[(size_t)v * 4 + k];
    const float w = a.nbr_w[(size_t)v * 4 + k];
    const NodeAttr n = node_attr(m, a.dx, a.dr, a.ds, a.dop);
    // ---- the forward blend state: this neighbour's terms, summed over the quad ----
    const v3 y = mk3(n.S[0] * p.x + n.S[1] * p.y + n.S[2] * p.z, n.S[3] * p.x + n.S[4] * p.y + n.S[5] * p.z,
                     n.S[6] * p.x + n.S[7] * p.y + n.S[8] * p.z);
    const float qn = sqrtf(qdot(n.q, n.q));
    const q4 qr = qscale(1.f / qn, n.q);
    const q4 av = q4{0.5f * n.t.x, 0.5f * n.t.y, 0.5f * n.t.z, 0.f};
    v3 x_lbs = mk3(0, 0, 0);
    q4 br = q4{0, 0, 0, 0}, bd = q4{0, 0, 0, 0};
    if (a.method != kDqs) x_lbs = quad_sum3(w * (qact(n.q, y) + n.t));
    if (a.method != kLbs) {
        br = quad_sum4(qscale(w, qr));
        bd = quad_sum4(qscale(w, qmul(av, qr)));
    }
    const float eta_raw = quad_sum(w * n.o);
    const v3 rho = quad_sum3(w * so3_log(n.q));
    // ---- gradients of the blended quantities (the same in the four lanes) ----
    float eta = 1.f, g_eta = 0.f;
    v3 g_lbs = gx, g_dqs = mk3(0, 0, 0);
    q4 g_br = q4{0, 0, 0, 0}, g_bd = q4{0, 0, 0, 0};
    if (a.method != kLbs) {
        const float nn = sqrtf(qdot(br, br));
        const q4 rh = qscale(1.f / nn, br), dh = qscale(1.f / nn, bd);
        if (a.method == kDqs) { g_dqs = gx; g_lbs = mk3(0, 0, 0); }
        else {
            const v3 x_dqs = qact(rh, p) + qv(qmul(qscale(2.f, dh), qconj(rh)));
            eta = fminf(eta_raw + 0.4f, 1.0f);
            g_lbs = eta * gx;
            g_dqs = (1.f - eta) * gx;
            g_eta = (eta_raw + 0.4f < 1.0f) ? dot(gx, x_lbs - x_dqs) : 0.f;
        }
        q4 g_rh = a.pypose ? qact_grad_q_pp(rh, p, g_dqs) : qact_grad_q(rh, p, g_dqs);     // q_r.matrix() of transform_point_simple
        const q4 G = q4{g_dqs.x, g_dqs.y, g_dqs.z, 0.f};
        q4 g_dh, g_b;                                                // translation = (2 dh) * conj(rh): a = 2 dh, b = conj(rh)
        dqs_mul_grad(a.pypose, qscale(2.f, dh), qconj(rh), G, g_dh, g_b);
        g_dh = qscale(2.f, g_dh);
        g_rh = qadd(g_rh, q4{-g_b.x, -g_b.y, -g_b.z, g_b.w});
        g_bd = qscale(1.f / nn, g_dh);
        g_br = qadd(qscale(1.f / nn, g_rh), qscale(-(qdot(g_rh, rh) + qdot(g_dh, dh)) / nn, rh));
    }
    const v3 g_rho = a.pypose ? so3_exp_grad_pp(rho, gq) : so3_exp_grad(rho, gq);
    // ---- this neighbour's gradients ----
    v3 g_t = mk3(0, 0, 0);
    q4 g_q = a.pypose ? so3_log_grad_pp(n.q, w * g_rho) : so3_log_grad(n.q, w * g_rho);
    float g_S[6] = {0, 0, 0, 0, 0, 0};
    if (a.method != kDqs) {
        const v3 h = w * g_lbs;
        g_t = g_t + h;
        g_q = qadd(g_q, a.pypose ? qact_grad_q_pp(n.q, y, h) : qact_grad_q(n.q, y, h));
        const v3 gy = qact_grad_p(n.q, h);
        g_S[0] = gy.x * p.x; g_S[1] = gy.y * p.y; g_S[2] = gy.z * p.z;
        g_S[3] = gy.x * p.y + gy.y * p.x;
        g_S[4] = gy.x * p.z + gy.z * p.x;
        g_S[5] = gy.y * p.z + gy.z * p.y;
    }
    if (a.method != kLbs) {
        const q4 g_d = qscale(w, g_bd);
        q4 g_a, g_qr;                                                   // q_d = (t / 2) * q_r
        dqs_mul_grad(a.pypose, av, qr, g_d, g_a, g_qr);
        g_t = g_t + 0.5f * qv(g_a);
        g_qr = qadd(g_qr, qscale(w, g_br));
        g_q = qadd(g_q, qscale(1.f / qn, qadd(g_qr, qscale(-qdot(g_qr, qr), qr))));
    }
    const q4 g_p = qscale(1.f / n.pn, qadd(g_q, qscale(-qdot(g_q, n.q), n.q)));
    const float g_o = w * g_eta;
    if (!live) return;
    float *r = rec + ((size_t)v * 4 + k) * kNodeRec;
    r[0] = g_t.x; r[1] = g_t.y; r[2] = g_t.z;
    r[3] = g_p.x; r[4] = g_p.y; r[5] = g_p.z; r[6] = g_p.w;
    r[7] = g_S[0]; r[8] = g_S[1]; r[9] = g_S[2]; r[10] = g_S[3]; r[11] = g_S[4]; r[12] = g_S[5];
    r[13] = g_o * n.o * (1.f - n.o);
}

// ---------------------------------------------------------------------------------------- backward 2
// per node: fixed-order sum of the records of every (vertex, k) that references it (static CSR)
__global__ __launch_bounds__(64) void k_skin_bwd_node(int M, const int32_t *__restrict__ csr_off,
                                                      const int32_t *__restrict__ csr_item,
                                                      const float *__restrict__ rec, float *__restrict__ g_dx,
                                                      float *__restrict__ g_dr, float *__restrict__ g_ds,
                                                      float *__restrict__ g_do, size_t rec_view_stride)
{
    // one WAVE per node: lanes stride over the node's (vertex, k) records, then a fixed-order
    // butterfly sums the 64 partials (deterministic)
    const int m = blockIdx.x, lane = threadIdx.x;
    if (m >= M) return;
    {
        const size_t bv = blockIdx.y, o = bv * M;
        rec += bv * rec_view_stride;
        if (g_dx) g_dx += o * 3;
        if (g_dr) g_dr += o * 4;
        if (g_ds) g_ds += o * 6;
        if (g_do) g_do += o;
    }
    float acc[kNodeRec];
#pragma unroll
    for (int i = 0; i < kNodeRec; ++i) acc[i] = 0.f;
    for (int e = csr_off[m] + lane; e < csr_off[m + 1]; e += 64) {
        const float *r = rec + (size_t)csr_item[e] * kNodeRec;
#pragma unroll
        for (int i = 0; i < kNodeRec; ++i) acc[i] += r[i];
    }
#pragma unroll
    for (int i = 0; i < kNodeRec; ++i) acc[i] = wave_sum_row3(acc[i]);   // fixed order, on the VALU; total in row 3
    if (lane != 63) return;
    if (g_dx) { g_dx[3 * m] = acc[0]; g_dx[3 * m + 1] = acc[1]; g_dx[3 * m + 2] = acc[2]; }
    if (g_dr) { g_dr[4 * m] = acc[3]; g_dr[4 * m + 1] = acc[4]; g_dr[4 * m + 2] = acc[5]; g_dr[4 * m + 3] = acc[6]; }
    if (g_ds) { for (int i = 0; i < 6; ++i) g_ds[6 * m + i] = acc[7 + i]; }
    if (g_do) g_do[m] = acc[13];
}

// ---------------------------------------------------------------------------------------- face -> Gaussians
constexpr int kMaxPerFace = 6;
__constant__ float c_bary[4][kMaxPerFace][3] = {
    {{1.f / 3, 1.f / 3, 1.f / 3}},
    {{1.f / 2, 1.f / 4, 1.f / 4}, {1.f / 4, 1.f / 2, 1.f / 4}, {1.f / 4, 1.f / 4, 1.f / 2}},
    {{1.f / 3, 1.f / 3, 1.f / 3}, {2.f / 3, 1.f / 6, 1.f / 6}, {1.f / 6, 2.f / 3, 1.f / 6}, {1.f / 6, 1.f / 6, 2.f / 3}},
    {{2.f / 3, 1.f / 6, 1.f / 6}, {1.f / 6, 2.f / 3, 1.f / 6}, {1.f / 6, 1.f / 6, 2.f / 3},
     {1.f / 6, 5.f / 12, 5.f / 12}, {5.f / 12, 1.f / 6, 5.f / 12}, {5.f / 12, 5.f / 12, 1.f / 6}}};   // sugar.py:235-276
__device__ __forceinline__ int bary_row(int G) { return G == 1 ? 0 : G == 3 ? 1 : G == 4 ? 2 : 3; }

// normals are written at normals[i * nstride + 0..2] (nstride 3, or 6 when they are the second half of a
// fused [N,6] colour buffer whose first half receives `rgb`)
__global__ __launch_bounds__(kSkinThreads) void k_face_fwd(int F, int G, int V, const int32_t *__restrict__ faces,
                                                           const float *__restrict__ vxyz,
                                                           const float *__restrict__ vrot,
                                                           const float *__restrict__ q_static /* [N,4] wxyz */,
                                                           float *__restrict__ means, float *__restrict__ rots,
                                                           float *__restrict__ normals, int nstride,
                                                           const float *__restrict__ rgb, float *__restrict__ colors6, ZeroList zl)
{
    // side job of the first workgroup (batched path): clear the rasterizer's per-view counters, which K1 -- the next launch --
    // adds to; as a launch of its own this was 5.6 us of the step's serial chain
    if (zl.n > 0 && blockIdx.x == 0 && blockIdx.y == 0)
        for (int t = threadIdx.x; t < zl.n * 64; t += kSkinThreads) reinterpret_cast<uint32_t *>(zl.base + (size_t)(t >> 6) * zl.stride)[t & 63] = 0u;
    const int i = blockIdx.x * kSkinThreads + threadIdx.x;
    if (i >= F * G) return;
    {
        const size_t bv = blockIdx.y, n = (size_t)F * G;
        vxyz += bv * V * 3;
        vrot += bv * V * 4;
        means += bv * n * 3;
        rots += bv * n * 4;
        if (normals) normals += bv * n * nstride;
        if (colors6) colors6 += bv * n * 6;
    }
    const int f = i / G, s = i - f * G;
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    const float *b = c_bary[bary_row(G)][s];
    const v3 x0 = ld3(vxyz, i0), x1 = ld3(vxyz, i1), x2 = ld3(vxyz, i2);
    st3(means, i, (b[0] * x0 + b[1] * x1) + b[2] * x2);
    const v3 r = (b[0] * so3_log(ldq(vrot, i0)) + b[1] * so3_log(ldq(vrot, i1))) + b[2] * so3_log(ldq(vrot, i2));
    const q4 qd = so3_exp(r);
    const float4 qs = reinterpret_cast<const float4 *>(q_static)[i];          // w,x,y,z
    const q4 Q = qmul(qd, q4{qs.y, qs.z, qs.w, qs.x});
    const float inv = 1.f / fmaxf(sqrtf(qdot(Q, Q)), 1e-12f);
    reinterpret_cast<float4 *>(rots)[i] = make_float4(Q.w * inv, Q.x * inv, Q.y * inv, Q.z * inv);
    if (normals) {
        const v3 c = cross(x1 - x0, x2 - x0);
        const v3 nn = (1.f / fmaxf(sqrtf(dot(c, c)), 1e-12f)) * c;
        float *o = normals + (size_t)i * nstride;
        o[0] = nn.x; o[1] = nn.y; o[2] = nn.z;
    }
    if (colors6) {
        float *o = colors6 + (size_t)i * 6;
        o[0] = rgb[3 * i]; o[1] = rgb[3 * i + 1]; o[2] = rgb[3 * i + 2];
    }
}

constexpr int kCornerRec = 6;   // dL/dx (3) + dL/d(rotation-vector blend) (3) per face corner
// The face part of the face -> Gaussian backward for the workgroup's faces, given every thread's (= Gaussian's, slot sl of face
// f) upstream gradients ALREADY SUMMED over the views of the frame: gm = dL/dmean, go = dL/drotation (x, y, z, w), gn =
// dL/dnormal.  vxyz / vrot / rec: the frame's.  Called by every thread of the workgroup (two barriers inside).
// The Exp / Exp-gradient of a slot is the heavy part and runs G-wide in parallel; the G slot results of a face meet in LDS and
// the slot-0 thread adds them in slot order (fixed: deterministic) and finishes the face.
__device__ __forceinline__ void face_bwd_finish(const int F, const int G, const int32_t *__restrict__ faces,
                                                const float *__restrict__ vxyz, const float *__restrict__ vrot,
                                                const float *__restrict__ q_static, const bool has_means, const bool has_rots,
                                                const bool has_normals, const bool live, const int f, const int sl, const int base,
                                                const v3 gm, const q4 go, const v3 gn_in, float *__restrict__ rec, const int pypose,
                                                float (*s_log)[3] /* LDS [kSkinThreads][3]: per thread of a face's first three slots: Log of vertex (tid % G) */,
                                                float (*s_val)[9] /* LDS [kSkinThreads][9]: per slot: gm (3), gr (3), gn (3) */)
{
    const int tid = threadIdx.x;
    // phase A: Log of the three vertex rotations, one per thread (G >= 3) or all by slot 0 (G == 1)
    if (live && has_rots) {
        if (G >= 3) {
            if (sl < 3) {
                const v3 L = so3_log(ldq(vrot, faces[3 * f + sl]));
                s_log[base + sl][0] = L.x; s_log[base + sl][1] = L.y; s_log[base + sl][2] = L.z;
            }
        }      // G == 1: the single slot computes the three logs itself (below)
    }
    __syncthreads();
    // phase B: per slot
    if (live) {
        const size_t i = (size_t)f * G + sl;
        const float *b = c_bary[bary_row(G)][sl];
        v3 gr = mk3(0, 0, 0);
        if (has_rots) {
            v3 L0, L1, L2;
            if (G >= 3) {
                L0 = mk3(s_log[base][0], s_log[base][1], s_log[base][2]);
                L1 = mk3(s_log[base + 1][0], s_log[base + 1][1], s_log[base + 1][2]);
                L2 = mk3(s_log[base + 2][0], s_log[base + 2][1], s_log[base + 2][2]);
            } else {
                L0 = so3_log(ldq(vrot, faces[3 * f]));
                L1 = so3_log(ldq(vrot, faces[3 * f + 1]));
                L2 = so3_log(ldq(vrot, faces[3 * f + 2]));
            }
            const v3 r = (b[0] * L0 + b[1] * L1) + b[2] * L2;
            const q4 qd = so3_exp(r);
            const float4 qs4 = reinterpret_cast<const float4 *>(q_static)[i];
            const q4 qs = q4{qs4.y, qs4.z, qs4.w, qs4.x};
            const q4 Q = qmul(qd, qs);
            const float nq = fmaxf(sqrtf(qdot(Q, Q)), 1e-12f);
            const q4 out = qscale(1.f / nq, Q);
            const q4 gQ = qscale(1.f / nq, qadd(go, qscale(-qdot(go, out), out)));
            if (pypose) gr = so3_exp_grad_pp(r, gQ);     // SO3_Mul.backward: X_grad = (g[:3], 0); so3_Exp.backward: g[:3] Jl(r)
            else gr = so3_exp_grad(r, qmul(gQ, qconj(qs)));
        }
        float *o = s_val[tid];
        o[0] = gm.x; o[1] = gm.y; o[2] = gm.z; o[3] = gr.x; o[4] = gr.y; o[5] = gr.z; o[6] = gn_in.x; o[7] = gn_in.y; o[8] = gn_in.z;
    }
    __syncthreads();
    // phase C: slot 0 adds the slots in order and finishes the face
    if (!live || sl != 0) return;
    v3 X[3] = {mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0)}, R[3] = {mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0)};
    v3 gn = mk3(0, 0, 0);
    for (int s2 = 0; s2 < G; ++s2) {
        const float *b = c_bary[bary_row(G)][s2];
        const float *o = s_val[base + s2];
        const v3 gm2 = mk3(o[0], o[1], o[2]), gr = mk3(o[3], o[4], o[5]);
        if (has_means) { X[0] = X[0] + b[0] * gm2; X[1] = X[1] + b[1] * gm2; X[2] = X[2] + b[2] * gm2; }
        if (has_rots) { R[0] = R[0] + b[0] * gr; R[1] = R[1] + b[1] * gr; R[2] = R[2] + b[2] * gr; }
        gn = gn + mk3(o[6], o[7], o[8]);
    }
    if (has_normals) {
        const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
        const v3 x0 = ld3(vxyz, i0), x1 = ld3(vxyz, i1), x2 = ld3(vxyz, i2);
        const v3 e1 = x1 - x0, e2 = x2 - x0, c = cross(e1, e2);
        const float cn = fmaxf(sqrtf(dot(c, c)), 1e-12f);
        const v3 nn = (1.f / cn) * c;
        const v3 gc = (1.f / cn) * (gn - dot(gn, nn) * nn);
        const v3 ge1 = cross(e2, gc), ge2 = cross(gc, e1);
        X[1] = X[1] + ge1;
        X[2] = X[2] + ge2;
        X[0] = X[0] - (ge1 + ge2);
    }
    float *o = rec + (size_t)f * 3 * kCornerRec;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        o[j * kCornerRec + 0] = X[j].x; o[j * kCornerRec + 1] = X[j].y; o[j * kCornerRec + 2] = X[j].z;
        o[j * kCornerRec + 3] = R[j].x; o[j * kCornerRec + 4] = R[j].y; o[j * kCornerRec + 5] = R[j].z;
    }
}

__global__ __launch_bounds__(kSkinThreads) void k_face_bwd_face(int F, int G, int V, const int32_t *__restrict__ faces,
                                                                const float *__restrict__ vxyz,
                                                                const float *__restrict__ vrot,
                                                                const float *__restrict__ q_static,
                                                                const float *__restrict__ g_means,
                                                                const float *__restrict__ g_rots,
                                                                const float *__restrict__ g_normals, int nstride,
                                                                float *__restrict__ rec /* [F][3][6] */,
                                                                const int32_t *__restrict__ frame_index, int n_views, int pypose)
{
    // blockIdx.y = frame; the upstream gradients are per VIEW: summed here over the views of the frame in view
    // order (the backward is linear in them).  frame_index == nullptr: view == frame.
    // One thread per (face, slot) = per Gaussian (coalesced reads of the per-Gaussian gradients).
    const int tid = threadIdx.x;
    const int faces_per_wg = kSkinThreads / G;
    const int fl = tid / G, sl = tid - fl * G;      // face in workgroup, slot
    const int f = blockIdx.x * faces_per_wg + fl;
    const bool live = fl < faces_per_wg && f < F;
    const size_t n = (size_t)F * G;
    const int frame = blockIdx.y;
    {
        const size_t bv = blockIdx.y;
        vxyz += bv * V * 3;
        vrot += bv * V * 4;
        rec += bv * F * 3 * kCornerRec;
    }
    const int b0 = frame_index ? 0 : frame, b1 = frame_index ? n_views : frame + 1;
    v3 gm = mk3(0, 0, 0), gn = mk3(0, 0, 0);
    q4 go = q4{0.f, 0.f, 0.f, 0.f};
    if (live) {
        const size_t i = (size_t)f * G + sl;
        for (int bv = b0; bv < b1; ++bv) {
            if (frame_index && frame_index[bv] != frame) continue;
            if (g_means) gm = gm + ld3(g_means + (size_t)bv * n * 3, i);
            if (g_rots) {
                const float4 go4 = reinterpret_cast<const float4 *>(g_rots + (size_t)bv * n * 4)[i];   // grads in w,x,y,z order
                go = qadd(go, q4{go4.y, go4.z, go4.w, go4.x});
            }
            if (g_normals) {
                const float *pn = g_normals + ((size_t)bv * n + i) * nstride;
                gn = gn + mk3(pn[0], pn[1], pn[2]);
            }
        }
    }
    __shared__ float s_log[kSkinThreads][3];
    __shared__ float s_val[kSkinThreads][9];
    face_bwd_finish(F, G, faces, vxyz, vrot, q_static, g_means != nullptr, g_rots != nullptr, g_normals != nullptr, live, f, sl, fl * G,
                    gm, go, gn, rec, pypose, s_log, s_val);
}

// per vertex: fixed-order sum over incident face corners (static CSR), then Log backward
// ext_*: optional extra upstream gradients on the deformed vertices (mesh regularisers), added here
// 8 lanes per vertex: lane c sums the corners e = c, c + 8, ... (a pole of a UV sphere has hundreds of incident
// faces: one thread per vertex made that vertex the critical path of the whole launch), then a fixed-order
// butterfly over the 8 lanes; lane 0 finishes the vertex.
__global__ __launch_bounds__(kSkinThreads) void k_face_bwd_vertex(int V, int F, const int32_t *__restrict__ csr_off,
                                                                  const int32_t *__restrict__ csr_item /* 3f+j */,
                                                                  const float *__restrict__ vrot,
                                                                  const float *__restrict__ rec,
                                                                  const float *__restrict__ ext_xyz,
                                                                  const float *__restrict__ ext_rot,
                                                                  float *__restrict__ g_vxyz, float *__restrict__ g_vrot, int pypose,
                                                                  const int32_t *__restrict__ view_frame, int n_view_recs)
{
    // n_view_recs > 0 (round 4: the fused record gather + face kernel with a thread per (VIEW, Gaussian), gather_face.hip): `rec` holds
    // one set of corner records per VIEW, [n_view_recs][F][3][6]; this frame's are those of the views b with view_frame[b] == frame
    // (view_frame == NULL: view == frame), added in view order.  Otherwise one set per frame.
    const int gid = blockIdx.x * kSkinThreads + threadIdx.x;
    const int v = gid >> 3, c = gid & 7;
    const bool live = v < V;
    const int frame = blockIdx.y;
    {
        const size_t bv = blockIdx.y;
        vrot += bv * V * 4;
        if (n_view_recs <= 0) rec += bv * F * 3 * kCornerRec;
        g_vxyz += bv * V * 3;
        g_vrot += bv * V * 4;
        if (ext_xyz) ext_xyz += bv * V * 3;
        if (ext_rot) ext_rot += bv * V * 4;
    }
    float a[kCornerRec] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
        const int nb = n_view_recs > 0 ? n_view_recs : 1;
        for (int b = 0; b < nb; ++b) {
            if (n_view_recs > 0 && (view_frame ? view_frame[b] : b) != frame) continue;      // (uniform)
            const float *__restrict__ rb = rec + (n_view_recs > 0 ? (size_t)b * F * 3 * kCornerRec : 0);
            for (int e = csr_off[v] + c; e < csr_off[v + 1]; e += 8) {
                const float2 *r = reinterpret_cast<const float2 *>(rb + (size_t)csr_item[e] * kCornerRec);
                const float2 r0 = r[0], r1 = r[1], r2 = r[2];
                a[0] += r0.x; a[1] += r0.y; a[2] += r1.x; a[3] += r1.y; a[4] += r2.x; a[5] += r2.y;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < kCornerRec; ++i) {
        a[i] = dpp_add<0xB1>(a[i]);     // lane ^ 1 (quad_perm [1,0,3,2])
        a[i] = dpp_add<0x4E>(a[i]);     // lane ^ 2 (quad_perm [2,3,0,1])
        a[i] = dpp_add<0x141>(a[i]);    // the other quad of the 8 lanes (row_half_mirror: its lanes all hold that quad's sum)
    }
    if (!live || c != 0) return;
    v3 X = mk3(a[0], a[1], a[2]);
    if (ext_xyz) X = X + ld3(ext_xyz, v);
    st3(g_vxyz, v, X);
    q4 g = pypose ? so3_log_grad_pp(ldq(vrot, v), mk3(a[3], a[4], a[5])) : so3_log_grad(ldq(vrot, v), mk3(a[3], a[4], a[5]));
    if (ext_rot) g = qadd(g, ldq(ext_rot, v));
    reinterpret_cast<float4 *>(g_vrot)[v] = make_float4(g.x, g.y, g.z, g.w);
}

}  // namespace dm4d

using namespace dm4d;

namespace dm4d {

int skin_forward_launch(int B, int method, int V, int M, int K, const float *verts, const int32_t *idx, const float *w,
                        const float *dx, const float *dr, const float *ds, const float *dop, float *out_xyz,
                        float *out_rot, hipStream_t st)
{
    if (V <= 0 || B <= 0) return DM4D_OK;
    method &= 0xff;
    SkinArgs a{method, V, M, K, 0, verts, idx, w, dx, dr, method == kDqs ? nullptr : ds, method == kHybrid ? dop : nullptr};
    ProfScope prof_(kKSkinFwd, st);
    hipLaunchKernelGGL(k_skin_fwd, dim3((V + kSkinThreads - 1) / kSkinThreads, B), dim3(kSkinThreads), 0, st, a, out_xyz, out_rot);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int skin_backward_launch(int B, int method, int V, int M, int K, const float *verts, const int32_t *idx, const float *w,
                         const float *dx, const float *dr, const float *ds, const float *dop, const float *g_xyz,
                         const float *g_rot, const int32_t *csr_off, const int32_t *csr_items, float *scratch,
                         float *o_dx, float *o_dr, float *o_ds, float *o_do, hipStream_t st)
{
    if (B <= 0) return DM4D_OK;
    const int pypose = (method & kPypose) ? 1 : 0;
    method &= 0xff;
    SkinArgs a{method, V, M, K, pypose, verts, idx, w, dx, dr, method == kDqs ? nullptr : ds, method == kHybrid ? dop : nullptr};
    ProfScope prof_(kKSkinBwd, st);
    if (V > 0) {
        if (K == 4)
            hipLaunchKernelGGL(k_skin_bwd_vertex_k4, dim3((4 * V + kSkinThreads - 1) / kSkinThreads, B), dim3(kSkinThreads), 0, st, a,
                               g_xyz, g_rot, scratch);
        else
            hipLaunchKernelGGL(k_skin_bwd_vertex, dim3((V + kSkinThreads - 1) / kSkinThreads, B), dim3(kSkinThreads), 0, st, a,
                               g_xyz, g_rot, scratch);
        DM4D_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_skin_bwd_node, dim3(M, B), dim3(64), 0, st, M, csr_off, csr_items,
                       (const float *)scratch, o_dx, o_dr, o_ds, o_do, (size_t)V * K * kNodeRec);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int face_forward_launch(int B, int F, int G, int V, const int32_t *faces, const float *vxyz, const float *vrot,
                        const float *qs, float *means, float *rots, float *normals, int nstride, const float *rgb,
                        float *colors6, hipStream_t st, char *zero_base, size_t zero_stride, int zero_n)
{
    if (F <= 0 || B <= 0) return zero_n > 0 ? DM4D_ERR_INVALID : DM4D_OK;
    G &= 0xff;
    ProfScope prof_(kKFaceFwd, st);
    const ZeroList zl = {zero_base, zero_stride, zero_base ? zero_n : 0};
    hipLaunchKernelGGL(k_face_fwd, dim3((F * G + kSkinThreads - 1) / kSkinThreads, B), dim3(kSkinThreads), 0, st, F, G, V,
                       faces, vxyz, vrot, qs, means, rots, normals, nstride, rgb, colors6, zl);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int face_backward_launch(int B, int F, int G, int V, const int32_t *faces, const float *vxyz, const float *vrot,
                         const float *qs, const float *g_means, const float *g_rots, const float *g_normals, int nstride,
                         const int32_t *csr_off, const int32_t *csr_items, float *scratch, const float *ext_xyz,
                         const float *ext_rot, float *o_vxyz, float *o_vrot, const int32_t *frame_index, int n_views,
                         hipStream_t st)
{
    // B = frames; with frame_index the upstream gradients g_* are per view ([n_views, N, .])
    if (B <= 0) return DM4D_OK;
    const int pypose = (G & kPypose) ? 1 : 0;
    G &= 0xff;
    ProfScope prof_(kKFaceBwd, st);
    // n_views < 0: the corner records were already written by the fused gather + face kernel -- -1: one set per frame; -(n + 1): one
    // set per VIEW (n views), summed over a frame's views by the vertex kernel
    if (F > 0 && !(n_views < 0)) {
        const int fpw = kSkinThreads / G;      // faces per workgroup (one thread per Gaussian)
        hipLaunchKernelGGL(k_face_bwd_face, dim3((F + fpw - 1) / fpw, B), dim3(kSkinThreads), 0, st, F, G,
                           V, faces, vxyz, vrot, qs, g_means, g_rots, g_normals, nstride, scratch, frame_index, n_views, pypose);
        DM4D_HIP_CHECK(hipGetLastError());
    }
    if (V > 0) {
        const int n_view_recs = n_views < -1 ? -n_views - 1 : 0;
        hipLaunchKernelGGL(k_face_bwd_vertex, dim3((8 * V + kSkinThreads - 1) / kSkinThreads, B), dim3(kSkinThreads), 0, st, V,
                           F, csr_off, csr_items, vrot, (const float *)scratch, ext_xyz, ext_rot, o_vxyz, o_vrot, pypose,
                           frame_index, n_view_recs);
        DM4D_HIP_CHECK(hipGetLastError());
    }
    return DM4D_OK;
}

int skin_check(int method, int V, int M, int K, const void *verts, const void *idx, const void *w, const void *dx,
               const void *dr, const void *ds, const void *dop)
{
    if ((method & ~kPypose) < 0 || (method & ~kPypose) > 2) { set_error("method must be 0 (lbs), 1 (dqs) or 2 (hybrid), optionally | DM4D_GRAD_PYPOSE"); return DM4D_ERR_INVALID; }
    method &= 0xff;
    if (V < 0 || M <= 0 || K <= 0 || K > kMaxK) { set_error("bad V/M/K (%d/%d/%d)", V, M, K); return DM4D_ERR_INVALID; }
    if (V > 0 && (!verts || !idx || !w || !dx || !dr)) { set_error("null input"); return DM4D_ERR_INVALID; }
    if (method != kDqs && !ds) { set_error("lbs/hybrid need the strain output ds"); return DM4D_ERR_INVALID; }
    if (method == kHybrid && !dop) { set_error("hybrid needs the opacity output"); return DM4D_ERR_INVALID; }
    return DM4D_OK;
}

int face_check(int F, int G, const void *faces, const void *vxyz, const void *vrot, const void *qs)
{
    G &= ~kPypose;
    if (F < 0 || !(G == 1 || G == 3 || G == 4 || G == 6)) { set_error("bad F/G (%d/%d); G must be 1, 3, 4 or 6", F, G); return DM4D_ERR_INVALID; }
    if (F > 0 && (!faces || !vxyz || !vrot || !qs)) { set_error("null input"); return DM4D_ERR_INVALID; }
    return DM4D_OK;
}

}  // namespace dm4d

using namespace dm4d;

extern "C" {

int dm4d_skin_vertices_forward(int32_t method, int32_t V, int32_t M, int32_t K, const float *verts,
                               const int32_t *nbr_idx, const float *nbr_w, const float *dx, const float *dr,
                               const float *ds, const float *d_opacity, float *out_xyz, float *out_rot,
                               dm4d_stream_t stream)
{
    int rc = skin_check(method, V, M, K, verts, nbr_idx, nbr_w, dx, dr, ds, d_opacity);
    if (rc) return rc;
    if (V == 0) return DM4D_OK;
    if (!out_xyz || !out_rot) { set_error("null output"); return DM4D_ERR_INVALID; }
    return skin_forward_launch(1, method, V, M, K, verts, nbr_idx, nbr_w, dx, dr, ds, d_opacity, out_xyz, out_rot,
                               (hipStream_t)stream);
}

size_t dm4d_skin_scratch_bytes(int32_t V, int32_t K) { return (size_t)(V > 0 ? V : 1) * K * kNodeRec * 4; }

int dm4d_skin_vertices_backward(int32_t method, int32_t V, int32_t M, int32_t K, const float *verts,
                                const int32_t *nbr_idx, const float *nbr_w, const float *dx, const float *dr,
                                const float *ds, const float *d_opacity, const float *dL_dxyz, const float *dL_drot,
                                const int32_t *node_csr_offsets, const int32_t *node_csr_items, void *scratch,
                                float *dL_ddx, float *dL_ddr, float *dL_dds, float *dL_ddo, dm4d_stream_t stream)
{
    int rc = skin_check(method, V, M, K, verts, nbr_idx, nbr_w, dx, dr, ds, d_opacity);
    if (rc) return rc;
    if (!node_csr_offsets || !node_csr_items || !scratch) { set_error("null csr/scratch"); return DM4D_ERR_INVALID; }
    return skin_backward_launch(1, method, V, M, K, verts, nbr_idx, nbr_w, dx, dr, ds, d_opacity, dL_dxyz, dL_drot,
                                node_csr_offsets, node_csr_items, (float *)scratch, dL_ddx, dL_ddr, dL_dds, dL_ddo,
                                (hipStream_t)stream);
}

int dm4d_face_gaussians_forward(int32_t F, int32_t G, const int32_t *faces, const float *vxyz, const float *vrot,
                                const float *q_static_wxyz, float *means, float *rotations_wxyz, float *normals,
                                dm4d_stream_t stream)
{
    int rc = face_check(F, G, faces, vxyz, vrot, q_static_wxyz);
    if (rc) return rc;
    if (F == 0) return DM4D_OK;
    if (!means || !rotations_wxyz) { set_error("null output"); return DM4D_ERR_INVALID; }
    return face_forward_launch(1, F, G, 0, faces, vxyz, vrot, q_static_wxyz, means, rotations_wxyz, normals, 3, nullptr,
                               nullptr, (hipStream_t)stream, nullptr, 0, 0);
}

size_t dm4d_face_scratch_bytes(int32_t F) { return (size_t)(F > 0 ? F : 1) * 3 * kCornerRec * 4; }

int dm4d_face_gaussians_backward(int32_t F, int32_t G, int32_t V, const int32_t *faces, const float *vxyz,
                                 const float *vrot, const float *q_static_wxyz, const float *dL_dmeans,
                                 const float *dL_drotations_wxyz, const float *dL_dnormals,
                                 const int32_t *vert_csr_offsets, const int32_t *vert_csr_items, void *scratch,
                                 float *dL_dvxyz, float *dL_dvrot, dm4d_stream_t stream)
{
    int rc = face_check(F, G, faces, vxyz, vrot, q_static_wxyz);
    if (rc) return rc;
    if (V < 0 || !vert_csr_offsets || !vert_csr_items || !scratch || !dL_dvxyz || !dL_dvrot) { set_error("null csr/scratch/output"); return DM4D_ERR_INVALID; }
    return face_backward_launch(1, F, G, V, faces, vxyz, vrot, q_static_wxyz, dL_dmeans, dL_drotations_wxyz, dL_dnormals,
                                3, vert_csr_offsets, vert_csr_items, (float *)scratch, nullptr, nullptr, dL_dvxyz,
                                dL_dvrot, nullptr, 1, (hipStream_t)stream);
}

}  // extern "C"
