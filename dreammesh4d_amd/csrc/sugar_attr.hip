// sugar_attr.hip -- the SuGaR geometry's per-Gaussian attributes from its parameters, one launch each way (gfx950).
//
// What a renderer call of the STATIC stage reads from `SuGaRModel` (custom/threestudio-dreammesh4d/geometry/sugar.py:471-570: get_xyz,
// get_scaling, get_rotation (quaternions), get_opacity (strengths), the degree-0 SH colour, the face normals) are ~100 torch operators
// on 50 k-element tensors, ~200 more in their backward (every parameter is learnt there): ~1 ms of a 12.7 ms iteration at the launch
// floor.  Here, for Gaussian i = slot s of face f = (v0, v1, v2):
//   mean     = b[s][0] v0 + b[s][1] v1 + b[s][2] v2                                            (surface_triangle_bary_coords)
//   scale    = (thickness, exp(ls0), exp(ls1));   opacity = sigmoid(density);   rgb = max(SH_C0 clip(sh, -c, c) + 0.5, 0)
//   normal   = n = normalize((v1 - v0) x (v2 - v0))
//   rotation = normalize(matrix_to_quaternion([n | r1 | r2])),  b1 = normalize(v0 - v1), b2 = normalize(n x b1), c = normalize(complex),
//              r1 = c0 b1 + c1 b2, r2 = -c1 b1 + c0 b2    (matrix_to_quaternion: pytorch3d's largest-component construction, w >= 0)
// forward: a thread per Gaussian; outputs are the rasterizer's inputs (means [N,3], rotations [N,4] (w,x,y,z), scales [N,3], opacities
// [N], colors [N,6] = rgb | normal).  backward: a thread per FACE walks its G Gaussians, keeps the face's vertex gradients in registers
// and adds them to dL/dpoints with 9 float atomics (the torch composition's index_add does the same); the per-Gaussian parameter
// gradients are plain stores.  Checked against the torch composition and its autograd (tests/test_static_stage_gpu.py).
#include "common.h"
#include "../../include/dm4d.h"

namespace dm4d {

struct SA3 { float x, y, z; };
__device__ __forceinline__ SA3 sa3(float x, float y, float z) { return SA3{x, y, z}; }
__device__ __forceinline__ SA3 operator+(SA3 a, SA3 b) { return SA3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ SA3 operator-(SA3 a, SA3 b) { return SA3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ SA3 operator*(float s, SA3 a) { return SA3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float sdot(SA3 a, SA3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ SA3 scross(SA3 a, SA3 b) { return SA3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ SA3 sload(const float *p, int i) { return SA3{p[3 * (size_t)i], p[3 * (size_t)i + 1], p[3 * (size_t)i + 2]}; }
// F.normalize: x / max(|x|, 1e-12); its backward: (g - xhat (xhat . g)) / |x| where the clamp is inactive, g / eps where it is
__device__ __forceinline__ SA3 snormalize(SA3 v, float &len) { len = fmaxf(sqrtf(sdot(v, v)), 1e-12f); return (1.0f / len) * v; }
__device__ __forceinline__ SA3 snormalize_bwd(SA3 vhat, float len, SA3 g)
{
    if (len <= 1e-12f) return (1.0f / len) * g;
    return (1.0f / len) * (g - sdot(vhat, g) * vhat);
}

constexpr float kSHC0 = 0.28209479177387814f;

struct FaceFrame { SA3 v0, v1, v2, e1, e2, cr, n, d01, b1, cb, b2; float ln, l1, l2; };
__device__ __forceinline__ FaceFrame face_frame(const float *points, const int64_t *faces, int f)
{
    FaceFrame F;
    const int i0 = (int)faces[3 * (size_t)f], i1 = (int)faces[3 * (size_t)f + 1], i2 = (int)faces[3 * (size_t)f + 2];
    F.v0 = sload(points, i0); F.v1 = sload(points, i1); F.v2 = sload(points, i2);
    F.e1 = F.v1 - F.v0; F.e2 = F.v2 - F.v0;
    F.cr = scross(F.e1, F.e2);
    F.n = snormalize(F.cr, F.ln);
    F.d01 = F.v0 - F.v1;
    F.b1 = snormalize(F.d01, F.l1);
    F.cb = scross(F.n, F.b1);
    F.b2 = snormalize(F.cb, F.l2);
    return F;
}

// pytorch3d.transforms.matrix_to_quaternion on R = [n | r1 | r2] (columns), standardised to w >= 0, then F.normalize
struct QuatFwd { float q[4]; float raw[4]; float x[4], qa[4]; int best; float den, sign, qlen; };
__device__ __forceinline__ QuatFwd quat_from_columns(SA3 n, SA3 r1, SA3 r2)
{
    // m[row][col]: col 0 = n, col 1 = r1, col 2 = r2
    const float m00 = n.x, m01 = r1.x, m02 = r2.x, m10 = n.y, m11 = r1.y, m12 = r2.y, m20 = n.z, m21 = r1.z, m22 = r2.z;
    QuatFwd o;
    o.x[0] = 1.0f + m00 + m11 + m22; o.x[1] = 1.0f + m00 - m11 - m22; o.x[2] = 1.0f - m00 + m11 - m22; o.x[3] = 1.0f - m00 - m11 + m22;
    int best = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { o.qa[k] = o.x[k] > 0.f ? sqrtf(o.x[k]) : 0.f; if (o.qa[k] > o.qa[best]) best = k; }      // argmax: the first maximum
    o.best = best;
    float c[4];
    if (best == 0) { c[0] = o.qa[0] * o.qa[0]; c[1] = m21 - m12; c[2] = m02 - m20; c[3] = m10 - m01; }
    else if (best == 1) { c[0] = m21 - m12; c[1] = o.qa[1] * o.qa[1]; c[2] = m10 + m01; c[3] = m02 + m20; }
    else if (best == 2) { c[0] = m02 - m20; c[1] = m10 + m01; c[2] = o.qa[2] * o.qa[2]; c[3] = m12 + m21; }
    else { c[0] = m10 - m01; c[1] = m20 + m02; c[2] = m21 + m12; c[3] = o.qa[3] * o.qa[3]; }
    o.den = 2.0f * fmaxf(o.qa[best], 0.1f);
#pragma unroll
    for (int k = 0; k < 4; ++k) o.raw[k] = c[k] / o.den;
    o.sign = o.raw[0] < 0.f ? -1.0f : 1.0f;
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { o.raw[k] *= o.sign; s2 += o.raw[k] * o.raw[k]; }
    o.qlen = fmaxf(sqrtf(s2), 1e-12f);
#pragma unroll
    for (int k = 0; k < 4; ++k) o.q[k] = o.raw[k] / o.qlen;
    return o;
}

struct SAArgs {
    int F, G, V;
    const float *points;          // [V,3]
    const int64_t *faces;         // [F,3]
    const float *bary;            // [G,3]
    const float *cx;              // [N,2] complex numbers
    const float *log_scales;      // [N,2]
    const float *densities;       // [N]
    const float *sh_dc;           // [N,3]
    float thickness, clip;
};

__global__ __launch_bounds__(256) void k_sugar_attr_fwd(SAArgs a, float *__restrict__ means, float *__restrict__ rots, float *__restrict__ scales,
                                                        float *__restrict__ opac, float *__restrict__ colors)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.F * a.G) return;
    const int f = i / a.G, s = i - f * a.G;
    const FaceFrame F = face_frame(a.points, a.faces, f);
    const float *b = a.bary + 3 * s;
    const SA3 mean = (b[0] * F.v0 + b[1] * F.v1) + b[2] * F.v2;
    means[3 * (size_t)i] = mean.x; means[3 * (size_t)i + 1] = mean.y; means[3 * (size_t)i + 2] = mean.z;
    float cl;
    const float c0r = a.cx[2 * (size_t)i], c1r = a.cx[2 * (size_t)i + 1];
    cl = fmaxf(sqrtf(c0r * c0r + c1r * c1r), 1e-12f);
    const float c0 = c0r / cl, c1 = c1r / cl;
    const SA3 r1 = c0 * F.b1 + c1 * F.b2, r2 = (-c1) * F.b1 + c0 * F.b2;
    const QuatFwd q = quat_from_columns(F.n, r1, r2);
#pragma unroll
    for (int k = 0; k < 4; ++k) rots[4 * (size_t)i + k] = q.q[k];
    scales[3 * (size_t)i] = a.thickness;
    scales[3 * (size_t)i + 1] = expf(a.log_scales[2 * (size_t)i]);
    scales[3 * (size_t)i + 2] = expf(a.log_scales[2 * (size_t)i + 1]);
    opac[i] = 1.0f / (1.0f + expf(-a.densities[i]));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float sh = fminf(fmaxf(a.sh_dc[3 * (size_t)i + k], -a.clip), a.clip);
        colors[6 * (size_t)i + k] = fmaxf(sh * kSHC0 + 0.5f, 0.0f);
    }
    colors[6 * (size_t)i + 3] = F.n.x; colors[6 * (size_t)i + 4] = F.n.y; colors[6 * (size_t)i + 5] = F.n.z;
}

// thread per face
__global__ __launch_bounds__(128) void k_sugar_attr_bwd(SAArgs a, const float *__restrict__ g_means, const float *__restrict__ g_rots,
                                                        const float *__restrict__ g_scales, const float *__restrict__ g_opac,
                                                        const float *__restrict__ g_colors, const float *__restrict__ scales_out,
                                                        const float *__restrict__ opac_out, float *__restrict__ g_points /* zeroed */,
                                                        float *__restrict__ g_cx, float *__restrict__ g_ls, float *__restrict__ g_den,
                                                        float *__restrict__ g_sh)
{
    const int f = blockIdx.x * 128 + threadIdx.x;
    if (f >= a.F) return;
    const FaceFrame F = face_frame(a.points, a.faces, f);
    SA3 gv0 = sa3(0, 0, 0), gv1 = sa3(0, 0, 0), gv2 = sa3(0, 0, 0);
    SA3 gn = sa3(0, 0, 0), gb1 = sa3(0, 0, 0), gb2 = sa3(0, 0, 0);       // dL/dn, dL/db1, dL/db2 summed over the face's Gaussians
    for (int s = 0; s < a.G; ++s) {
        const size_t i = (size_t)f * a.G + s;
        const float *b = a.bary + 3 * s;
        // mean
        if (g_means) {
            const SA3 gm = sa3(g_means[3 * i], g_means[3 * i + 1], g_means[3 * i + 2]);
            gv0 = gv0 + b[0] * gm; gv1 = gv1 + b[1] * gm; gv2 = gv2 + b[2] * gm;
        }
        // normal (colour channels 3..5)
        if (g_colors) gn = gn + sa3(g_colors[6 * i + 3], g_colors[6 * i + 4], g_colors[6 * i + 5]);
        // elementwise parameters
        if (g_ls) {
            g_ls[2 * i] = g_scales ? g_scales[3 * i + 1] * scales_out[3 * i + 1] : 0.f;           // exp backward: grad * result
            g_ls[2 * i + 1] = g_scales ? g_scales[3 * i + 2] * scales_out[3 * i + 2] : 0.f;
        }
        if (g_den) { const float o = opac_out[i]; g_den[i] = g_opac ? g_opac[i] * ((1.0f - o) * o) : 0.f; }       // sigmoid backward
        if (g_sh) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float sh = a.sh_dc[3 * i + k];
                const float cs = fminf(fmaxf(sh, -a.clip), a.clip);
                const bool pass = (sh >= -a.clip && sh <= a.clip) && (cs * kSHC0 + 0.5f >= 0.0f);      // clamp / clamp_min masks (bounds included, as torch)
                g_sh[3 * i + k] = (g_colors && pass) ? g_colors[6 * i + k] * kSHC0 : 0.f;
            }
        }
        // rotation
        const float c0r = a.cx[2 * i], c1r = a.cx[2 * i + 1];
        const float cn = sqrtf(c0r * c0r + c1r * c1r), cl = fmaxf(cn, 1e-12f);
        const float c0 = c0r / cl, c1 = c1r / cl;
        float gc0 = 0.f, gc1 = 0.f;
        if (g_rots) {
            const SA3 r1 = c0 * F.b1 + c1 * F.b2, r2 = (-c1) * F.b1 + c0 * F.b2;
            const QuatFwd q = quat_from_columns(F.n, r1, r2);
            // normalize backward, sign, division by den
            float go[4], dotq = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { go[k] = g_rots[4 * i + k]; dotq += q.q[k] * go[k]; }
            float graw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) graw[k] = (q.qlen <= 1e-12f ? go[k] / q.qlen : (go[k] - q.q[k] * dotq) / q.qlen) * q.sign;
            // raw_k (before the sign) = c_k / den:  g c_k = graw_k / den;  g den = - sum graw_k c_k / den^2
            float gcand[4], gden = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { gcand[k] = graw[k] / q.den; gden -= graw[k] * (q.raw[k] * q.sign) / q.den; }
            // den = 2 max(qa_best, 0.1); the diagonal candidate is qa_best^2
            float gqa = 2.0f * q.qa[q.best] * gcand[q.best];
            if (q.qa[q.best] > 0.1f) gqa += 2.0f * gden;
            const float gx = q.x[q.best] > 0.f ? gqa / (2.0f * q.qa[q.best]) : 0.f;
            // x_best = 1 + s0 m00 + s1 m11 + s2 m22
            float g00, g11, g22, g01 = 0.f, g02 = 0.f, g10 = 0.f, g12 = 0.f, g20 = 0.f, g21 = 0.f;
            const int bst = q.best;
            g00 = (bst == 0 || bst == 1) ? gx : -gx;
            g11 = (bst == 0 || bst == 2) ? gx : -gx;
            g22 = (bst == 0 || bst == 3) ? gx : -gx;
            if (bst == 0) { g21 += gcand[1]; g12 -= gcand[1]; g02 += gcand[2]; g20 -= gcand[2]; g10 += gcand[3]; g01 -= gcand[3]; }
            else if (bst == 1) { g21 += gcand[0]; g12 -= gcand[0]; g10 += gcand[2]; g01 += gcand[2]; g02 += gcand[3]; g20 += gcand[3]; }
            else if (bst == 2) { g02 += gcand[0]; g20 -= gcand[0]; g10 += gcand[1]; g01 += gcand[1]; g12 += gcand[3]; g21 += gcand[3]; }
            else { g10 += gcand[0]; g01 -= gcand[0]; g20 += gcand[1]; g02 += gcand[1]; g21 += gcand[2]; g12 += gcand[2]; }
            // columns: n = (m00, m10, m20), r1 = (m01, m11, m21), r2 = (m02, m12, m22)
            gn = gn + sa3(g00, g10, g20);
            const SA3 gr1 = sa3(g01, g11, g21), gr2 = sa3(g02, g12, g22);
            // r1 = c0 b1 + c1 b2, r2 = -c1 b1 + c0 b2
            gb1 = gb1 + (c0 * gr1 + (-c1) * gr2);
            gb2 = gb2 + (c1 * gr1 + c0 * gr2);
            gc0 = sdot(gr1, F.b1) + sdot(gr2, F.b2);
            gc1 = sdot(gr1, F.b2) - sdot(gr2, F.b1);
        }
        if (g_cx) {
            // c = normalize(cx)
            float gx0, gx1;
            if (cn <= 1e-12f) { gx0 = gc0 / cl; gx1 = gc1 / cl; }
            else { const float d = c0 * gc0 + c1 * gc1; gx0 = (gc0 - c0 * d) / cl; gx1 = (gc1 - c1 * d) / cl; }
            g_cx[2 * i] = gx0; g_cx[2 * i + 1] = gx1;
        }
    }
    // b2 = normalize(n x b1)
    const SA3 gcb = snormalize_bwd(F.b2, F.l2, gb2);
    gn = gn + scross(F.b1, gcb);                 // d(n x b1) / dn:  g_n = b1 x g
    gb1 = gb1 + scross(gcb, F.n);                // d(n x b1) / db1: g_b1 = g x n
    // b1 = normalize(v0 - v1)
    const SA3 gd01 = snormalize_bwd(F.b1, F.l1, gb1);
    gv0 = gv0 + gd01; gv1 = gv1 - gd01;
    // n = normalize(e1 x e2), e1 = v1 - v0, e2 = v2 - v0
    const SA3 gcr = snormalize_bwd(F.n, F.ln, gn);
    const SA3 ge1 = scross(F.e2, gcr), ge2 = scross(gcr, F.e1);
    gv1 = gv1 + ge1; gv2 = gv2 + ge2; gv0 = gv0 - (ge1 + ge2);
    if (g_points) {
        const int i0 = (int)a.faces[3 * (size_t)f], i1 = (int)a.faces[3 * (size_t)f + 1], i2 = (int)a.faces[3 * (size_t)f + 2];
        atomicAdd(g_points + 3 * (size_t)i0, gv0.x); atomicAdd(g_points + 3 * (size_t)i0 + 1, gv0.y); atomicAdd(g_points + 3 * (size_t)i0 + 2, gv0.z);
        atomicAdd(g_points + 3 * (size_t)i1, gv1.x); atomicAdd(g_points + 3 * (size_t)i1 + 1, gv1.y); atomicAdd(g_points + 3 * (size_t)i1 + 2, gv1.z);
        atomicAdd(g_points + 3 * (size_t)i2, gv2.x); atomicAdd(g_points + 3 * (size_t)i2 + 1, gv2.y); atomicAdd(g_points + 3 * (size_t)i2 + 2, gv2.z);
    }
}

}  // namespace dm4d

using namespace dm4d;

extern "C" {

static int sa_check(int F, int G, int V, const void *points, const void *faces, const void *bary, const void *cx, const void *ls, const void *den,
                    const void *sh)
{
    if (F < 0 || V <= 0 || (G != 1 && G != 3 && G != 4 && G != 6)) { set_error("sugar attributes: F %d, V %d, G %d", F, V, G); return DM4D_ERR_INVALID; }
    if (F == 0) return DM4D_OK;
    if (!points || !faces || !bary || !cx || !ls || !den || !sh) { set_error("sugar attributes: null tensor"); return DM4D_ERR_INVALID; }
    return DM4D_OK;
}

int dm4d_sugar_attributes_forward(int32_t F, int32_t G, int32_t V, const float *points, const int64_t *faces, const float *bary, const float *complex_numbers,
                                  const float *log_scales, const float *densities, const float *sh_dc, float thickness, float color_clip,
                                  float *means3D, float *rotations, float *scales, float *opacities, float *colors6, dm4d_stream_t stream)
{
    int rc = sa_check(F, G, V, points, faces, bary, complex_numbers, log_scales, densities, sh_dc);
    if (rc != DM4D_OK || F == 0) return rc;
    if (!means3D || !rotations || !scales || !opacities || !colors6) { set_error("sugar attributes: null output"); return DM4D_ERR_INVALID; }
    SAArgs a{F, G, V, points, faces, bary, complex_numbers, log_scales, densities, sh_dc, thickness, color_clip};
    const int N = F * G;
    hipLaunchKernelGGL(k_sugar_attr_fwd, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, means3D, rotations, scales, opacities, colors6);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_sugar_attributes_backward(int32_t F, int32_t G, int32_t V, const float *points, const int64_t *faces, const float *bary,
                                   const float *complex_numbers, const float *log_scales, const float *densities, const float *sh_dc, float thickness,
                                   float color_clip, const float *scales, const float *opacities, const float *dL_dmeans3D, const float *dL_drotations,
                                   const float *dL_dscales, const float *dL_dopacities, const float *dL_dcolors6, float *dL_dpoints,
                                   float *dL_dcomplex, float *dL_dlog_scales, float *dL_ddensities, float *dL_dsh_dc, dm4d_stream_t stream)
{
    int rc = sa_check(F, G, V, points, faces, bary, complex_numbers, log_scales, densities, sh_dc);
    if (rc != DM4D_OK || F == 0) return rc;
    if (!scales || !opacities) { set_error("sugar attributes: backward needs the forward's scales / opacities"); return DM4D_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    if (dL_dpoints) DM4D_HIP_CHECK(hipMemsetAsync(dL_dpoints, 0, (size_t)V * 3 * sizeof(float), st));
    SAArgs a{F, G, V, points, faces, bary, complex_numbers, log_scales, densities, sh_dc, thickness, color_clip};
    hipLaunchKernelGGL(k_sugar_attr_bwd, dim3((F + 127) / 128), dim3(128), 0, st, a, dL_dmeans3D, dL_drotations, dL_dscales, dL_dopacities, dL_dcolors6,
                       scales, opacities, dL_dpoints, dL_dcomplex, dL_dlog_scales, dL_ddensities, dL_dsh_dc);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // extern "C"
