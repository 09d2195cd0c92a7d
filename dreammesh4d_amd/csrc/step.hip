// step.hip -- the STEP OBJECT: one C call each way for the whole per-(frame, view) path of a dynamic-stage step,
//
//   forward : deformation network of the nodes (dm4d_nodenet_forward) -> dm4d_views_forward
//   backward: dm4d_views_backward -> dm4d_nodenet_backward (parameter gradients into the caller's persistent buffers)
//
// for a FIXED problem (mesh, graph, image size, batch, capacities, parameter tensors): the object owns every struct, pointer
// table and host-side array of the two operators (deep copies: nothing of the descriptor has to outlive dm4d_step_create),
// the device memory stays the caller's, allocated ONCE.  What changes from step to step -- timestamps, cameras, the
// view -> frame map, the upstream image gradients -- are the arguments of the two calls.
//
// Why: at ~1.0 ms of GPU work per 8-view step the host side of `render_views` + `node_outputs` (two autograd Functions, ~60
// tensor allocations, ctypes struct marshalling: 0.76-0.88 ms per step, tools/host_bound.py) is as long as the step itself, so
// kernel time saved no longer shows in the step time.  With the step object the host's share of a step is the ~21 kernel
// launches and two Python -> C transitions (dreammesh4d_amd/step.py).
//
// Replaces the Python loop of custom/threestudio-dreammesh4d/renderer/gaussian_batch_renderer.py:21-76 around
// geometry/dynamic_sugar.py:367-431 (the per-timestamp deformation query) for one training step.
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "common.h"
#include "raster.h"

struct dm4d_step {
    dm4d_step_desc d;
    std::vector<int32_t> res;
    std::vector<float> aabb;
    std::vector<const float *> planes;
    std::vector<float *> g_planes;
    uint32_t magic;
};

static const uint32_t kStepMagic = 0x53544550u;   // "STEP"

extern "C" {

int dm4d_step_create(const dm4d_step_desc *desc, dm4d_step **out)
{
    using namespace dm4d;
    if (!desc || !out) { set_error("step: null descriptor"); return DM4D_ERR_INVALID; }
    *out = nullptr;
    const int S = desc->S;
    if (S <= 0 || S > 8) { set_error("step: %d HexPlane scales", S); return DM4D_ERR_INVALID; }
    if (!desc->res || !desc->aabb_host || !desc->planes || !desc->g_planes || !desc->nodes || !desc->feat || !desc->samples ||
        !desc->h_save || !desc->y_save || !desc->g_feat || !desc->net_scratch) {
        set_error("step: null tensor in the deformation-network part of the descriptor");
        return DM4D_ERR_INVALID;
    }
    const int nh = desc->w.n_heads;
    if (nh <= 0 || nh > 4) { set_error("step: %d MLP heads", nh); return DM4D_ERR_INVALID; }
    for (int k = 0; k < nh; ++k)
        if (!desc->node_out[k]) { set_error("step: null node output %d", k); return DM4D_ERR_INVALID; }
    const dm4d_views &v = desc->views;
    if (v.B <= 0 || v.M <= 0) { set_error("step: bad batch / node count"); return DM4D_ERR_INVALID; }
    if (!v.dx || !v.dr) { set_error("step: views.dx / views.dr must point at the node outputs"); return DM4D_ERR_INVALID; }
    if (desc->grads.dL_dopacity || desc->grads.dL_dscales) {
        set_error("step: the step object is the dynamic stage's (static appearance frozen): dL_dopacity / dL_dscales must be NULL");
        return DM4D_ERR_UNSUPPORTED;
    }
    dm4d_step *s = new (std::nothrow) dm4d_step;
    if (!s) { set_error("step: out of host memory"); return DM4D_ERR_INVALID; }
    s->d = *desc;
    s->res.assign(desc->res, desc->res + 4 * S);
    s->aabb.assign(desc->aabb_host, desc->aabb_host + 6);
    s->planes.assign(desc->planes, desc->planes + 6 * S);
    s->g_planes.assign(desc->g_planes, desc->g_planes + 6 * S);
    s->d.res = s->res.data();
    s->d.aabb_host = s->aabb.data();
    s->d.planes = s->planes.data();
    s->d.g_planes = s->g_planes.data();
    s->magic = kStepMagic;
    *out = s;
    return DM4D_OK;
}

void dm4d_step_destroy(dm4d_step *s)
{
    if (!s || s->magic != kStepMagic) return;
    s->magic = 0;
    delete s;
}

int dm4d_step_forward(dm4d_step *s, const float *times01, const float *viewmatrix, const float *projmatrix,
                      const int32_t *frame_index, dm4d_stream_t stream)
{
    using namespace dm4d;
    if (!s || s->magic != kStepMagic) { set_error("step: not a step object"); return DM4D_ERR_INVALID; }
    if (!times01 || !viewmatrix || !projmatrix) { set_error("step: null timestamps / cameras"); return DM4D_ERR_INVALID; }
    dm4d_step_desc &d = s->d;
    dm4d_views &v = d.views;
    const int NF = frame_index ? v.n_frames : v.B;
    if (frame_index && (NF <= 0 || NF > v.B)) { set_error("step: n_frames %d out of range", NF); return DM4D_ERR_INVALID; }
    d.times = times01;
    v.viewmatrix = viewmatrix;
    v.projmatrix = projmatrix;
    v.frame_index = frame_index;
    int rc = dm4d_nodenet_forward(d.S, v.M, NF, d.res, d.planes, d.hex_flags, d.aabb_host, d.nodes, times01, &d.w, d.feat, d.samples,
                                  d.h_save, d.y_save, d.node_out, d.net_scratch, stream);
    if (rc) return rc;
    return dm4d_views_forward(&v, stream);
}

static int step_backward_impl(dm4d_step *s, const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha,
                              const float *dL_dvxyz_ext, const float *dL_dvrot_ext, dm4d_stream_t stream, bool rgb_only);

int dm4d_step_backward(dm4d_step *s, const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha,
                       const float *dL_dvxyz_ext, const float *dL_dvrot_ext, dm4d_stream_t stream)
{
    return step_backward_impl(s, dL_dcolor, dL_ddepth, dL_dalpha, dL_dvxyz_ext, dL_dvrot_ext, stream, false);
}

int dm4d_step_backward_rgb(dm4d_step *s, const float *dL_dcolor, const float *dL_dalpha, const float *dL_dvxyz_ext,
                           const float *dL_dvrot_ext, dm4d_stream_t stream)
{
    return step_backward_impl(s, dL_dcolor, nullptr, dL_dalpha, dL_dvxyz_ext, dL_dvrot_ext, stream, true);
}

static int step_backward_impl(dm4d_step *s, const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha,
                              const float *dL_dvxyz_ext, const float *dL_dvrot_ext, dm4d_stream_t stream, bool rgb_only)
{
    using namespace dm4d;
    if (!s || s->magic != kStepMagic) { set_error("step: not a step object"); return DM4D_ERR_INVALID; }
    dm4d_step_desc &d = s->d;
    if (!d.times) { set_error("step: backward before forward"); return DM4D_ERR_INVALID; }
    dm4d_views_grads &g = d.grads;
    g.dL_dcolor = dL_dcolor; g.dL_ddepth = dL_ddepth; g.dL_dalpha = dL_dalpha;
    g.dL_dvxyz_ext = dL_dvxyz_ext; g.dL_dvrot_ext = dL_dvrot_ext;
    int rc = rgb_only ? dm4d_views_backward_rgb(&d.views, &g, stream) : dm4d_views_backward(&d.views, &g, stream);
    if (rc) return rc;
    const dm4d_views &v = d.views;
    const int NF = v.frame_index ? v.n_frames : v.B;
    return dm4d_nodenet_backward(d.S, v.M, NF, d.res, d.planes, d.hex_flags | d.hex_backward_flags, d.aabb_host, d.nodes, d.times, &d.w, d.feat,
                                 d.samples, d.h_save, d.y_save, d.node_gout, d.n_spatial, d.sp_scale, d.sp_plane, d.sp_texel, d.sp_off,
                                 d.sp_item, d.n_time, d.tp_scale, d.tp_plane, d.tp_col, d.tp_off, d.tp_item, d.g_feat, d.g_planes, &d.gw,
                                 d.net_scratch, stream);
}

const dm4d_views *dm4d_step_views(const dm4d_step *s)
{
    return (s && s->magic == kStepMagic) ? &s->d.views : nullptr;
}

}  // extern "C"
