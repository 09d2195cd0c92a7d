// views.hip -- the whole per-view hot path for a BATCH of (frame, view) units in 8 + 6 launches:
//
//   forward : skin vertices -> face->Gaussians (+ normals into the fused colour buffer) -> zero
//             counters -> preprocess -> colscan -> scatter -> tile sort -> blend (6 channels)
//   backward: blend bwd -> gather + preprocess bwd -> face bwd (faces, vertices) -> skin bwd
//             (vertices, nodes)
//
// Every launch covers all B views (grid.y = view), there is NO host synchronisation (the duplicate
// lists use a caller-chosen capacity; overflow raises a flag the host reads lazily), and the
// RGB pass and the normal pass of the reference's renderer
// (custom/threestudio-dreammesh4d/renderer/diff_sugar_rasterizer_temporal.py:161-217) share one
// binning and one blend.  This is what custom/.../renderer/gaussian_batch_renderer.py:21-76 does with a
// Python loop over views and two rasterizer calls (two host syncs) per view.
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "common.h"
#include "raster.h"

namespace dm4d {
int skin_forward_launch(int B, int method, int V, int M, int K, const float *verts, const int32_t *idx, const float *w,
                        const float *dx, const float *dr, const float *ds, const float *dop, float *out_xyz,
                        float *out_rot, hipStream_t st);
int skin_backward_launch(int B, int method, int V, int M, int K, const float *verts, const int32_t *idx, const float *w,
                         const float *dx, const float *dr, const float *ds, const float *dop, const float *g_xyz,
                         const float *g_rot, const int32_t *csr_off, const int32_t *csr_items, float *scratch,
                         float *o_dx, float *o_dr, float *o_ds, float *o_do, hipStream_t st);
int face_forward_launch(int B, int F, int G, int V, const int32_t *faces, const float *vxyz, const float *vrot,
                        const float *qs, float *means, float *rots, float *normals, int nstride, const float *rgb,
                        float *colors6, hipStream_t st, char *zero_base = nullptr, size_t zero_stride = 0, int zero_n = 0);
int face_backward_launch(int B, int F, int G, int V, const int32_t *faces, const float *vxyz, const float *vrot,
                         const float *qs, const float *g_means, const float *g_rots, const float *g_normals, int nstride,
                         const int32_t *csr_off, const int32_t *csr_items, float *scratch, const float *ext_xyz,
                         const float *ext_rot, float *o_vxyz, float *o_vrot, const int32_t *frame_index, int n_views,
                         hipStream_t st);
int launch_gather_face_bwd(const BatchDesc &d, int n_frames, int F, int G, int V, const int32_t *faces, const float *vxyz, const float *vrot,
                           const float *qs, float *face_scratch, hipStream_t st);
int skin_check(int method, int V, int M, int K, const void *verts, const void *idx, const void *w, const void *dx,
               const void *dr, const void *ds, const void *dop);
int face_check(int F, int G, const void *faces, const void *vxyz, const void *vrot, const void *qs);

static int views_check(const dm4d_views *v)
{
    if (!v) { set_error("null views"); return DM4D_ERR_INVALID; }
    if (v->B <= 0 || v->B > 65535) { set_error("bad batch size %d", v->B); return DM4D_ERR_INVALID; }
    if (v->frame_index && (v->n_frames <= 0 || v->n_frames > v->B)) { set_error("n_frames %d out of range (1..B)", v->n_frames); return DM4D_ERR_INVALID; }
    if (v->N != v->F * v->G) { set_error("N (%d) != F*G (%d*%d)", v->N, v->F, v->G); return DM4D_ERR_INVALID; }
    if (v->N > (1 << kGidBits)) { set_error("N = %d: at most %d Gaussians per view", v->N, 1 << kGidBits); return DM4D_ERR_UNSUPPORTED; }
    if (v->image_height <= 0 || v->image_width <= 0) { set_error("bad image size"); return DM4D_ERR_INVALID; }
    if ((int64_t)((v->image_height + kTile - 1) / kTile) * ((v->image_width + kTile - 1) / kTile) > kMaxTiles) {
        set_error("image has more than %d tiles", kMaxTiles);
        return DM4D_ERR_UNSUPPORTED;
    }
    if (v->capacity <= 0 || v->capacity > 0xFFFFFFF0ll) { set_error("capacity out of range"); return DM4D_ERR_INVALID; }
    if (v->record_mode != DM4D_RECORDS_CELL && v->record_mode != DM4D_RECORDS_TILE) { set_error("bad record_mode %d", v->record_mode); return DM4D_ERR_INVALID; }
    if (v->record_capacity <= 0 || v->record_capacity > 0xFFFFFFF0ll) { set_error("record_capacity out of range"); return DM4D_ERR_INVALID; }
    if (!v->bg || !v->viewmatrix || !v->projmatrix || !v->q_static || !v->scales || !v->opacities || !v->rgb ||
        !v->vxyz || !v->vrot || !v->means3D || !v->rotations || !v->colors || !v->radii || !v->geom || !v->binning ||
        !v->image) {
        set_error("null tensor in dm4d_views");
        return DM4D_ERR_INVALID;
    }
    int rc = skin_check(v->method, v->V, v->M, v->K, v->verts, v->nbr_idx, v->nbr_w, v->dx, v->dr, v->ds, v->d_opacity);
    if (rc) return rc;
    return face_check(v->F, v->G, v->faces, v->vxyz, v->vrot, v->q_static);
}

static BatchDesc views_batch(const dm4d_views *v)
{
    BatchDesc d;
    memset(&d, 0, sizeof(d));
    d.B = v->B; d.N = v->N; d.C = 6; d.W = v->image_width; d.H = v->image_height;
    d.frame_index = v->frame_index;
    d.tanfovx = v->tanfovx; d.tanfovy = v->tanfovy; d.scale_modifier = v->scale_modifier;
    d.bg = v->bg;
    d.view = v->viewmatrix; d.proj = v->projmatrix; d.cam_stride = 16;
    d.means3D = v->means3D; d.means_stride = (size_t)v->N * 3;
    d.rotations = v->rotations; d.rot_stride = (size_t)v->N * 4;
    d.colors = v->colors; d.color_stride = (size_t)v->N * 6;
    d.scales = v->scales; d.scale_stride = v->scales_per_frame ? (size_t)v->N * 3 : 0; d.scales_by_frame = v->scales_per_frame ? 1 : 0;
    d.opacities = v->opacities; d.opac_stride = 0;
    d.radii = v->radii; d.radii_stride = (size_t)v->N;
    d.geom = (char *)v->geom; d.geom_stride = geom_layout(v->N, v->image_height, v->image_width).total;
    d.binning = (char *)v->binning; d.bin_stride = binning_bytes(v->capacity); d.cap = (uint32_t)v->capacity;
    d.rec_cap = (uint32_t)v->record_capacity;
    d.tile_records = v->record_mode == DM4D_RECORDS_TILE ? 1 : 0;
    d.image = (char *)v->image; d.img_stride = image_bytes(v->image_height, v->image_width);
    d.out_color = v->out_color; d.out_depth = v->out_depth; d.out_alpha = v->out_alpha;
    return d;
}

// views [b0, b0 + nb) of a batch as a batch of their own
static BatchDesc sub_batch(const BatchDesc &d, int b0, int nb)
{
    BatchDesc s = d;
    const size_t b = (size_t)b0, N = (size_t)d.N, P = (size_t)d.H * d.W;
    s.B = nb;
    if (d.frame_index) s.frame_index = d.frame_index + b0;
    else {
        if (d.means3D) s.means3D = d.means3D + b * d.means_stride;
        if (d.rotations) s.rotations = d.rotations + b * d.rot_stride;
        if (d.colors) s.colors = d.colors + b * d.color_stride;
    }
    s.view = d.view + b * d.cam_stride; s.proj = d.proj + b * d.cam_stride;
    if (d.campos) s.campos = d.campos + b * d.campos_stride;
    if (d.scales && !(d.scales_by_frame && d.frame_index)) s.scales = d.scales + b * d.scale_stride;   // (per-frame: resolved through frame_index)
    if (d.opacities) s.opacities = d.opacities + b * d.opac_stride;
    if (d.shs) s.shs = d.shs + b * d.sh_stride;
    if (d.cov3D) s.cov3D = d.cov3D + b * d.cov_stride;
    if (d.radii) s.radii = d.radii + b * d.radii_stride;
    s.geom = d.geom + b * d.geom_stride;
    if (d.binning) s.binning = d.binning + b * d.bin_stride;
    if (d.image) s.image = d.image + b * d.img_stride;
    if (d.out_color) s.out_color = d.out_color + b * d.C * P;
    if (d.out_depth) s.out_depth = d.out_depth + b * P;
    if (d.out_alpha) s.out_alpha = d.out_alpha + b * P;
    if (d.dL_dcolor) s.dL_dcolor = d.dL_dcolor + b * d.C * P;
    if (d.dL_ddepth) s.dL_ddepth = d.dL_ddepth + b * P;
    if (d.dL_dalpha) s.dL_dalpha = d.dL_dalpha + b * P;
    if (d.dLq) s.dLq = d.dLq + b * d.dlq_stride;
    if (d.o.dL_dmeans2D) s.o.dL_dmeans2D = d.o.dL_dmeans2D + b * N * 3;
    if (d.o.dL_dmeans3D) s.o.dL_dmeans3D = d.o.dL_dmeans3D + b * N * 3;
    if (d.o.dL_dopacity) s.o.dL_dopacity = d.o.dL_dopacity + b * N;
    if (d.o.dL_dcolors) s.o.dL_dcolors = d.o.dL_dcolors + b * N * d.C;
    if (d.o.dL_dsh) s.o.dL_dsh = d.o.dL_dsh + b * N * d.sh_coeffs * 3;
    if (d.o.dL_dscales) s.o.dL_dscales = d.o.dL_dscales + b * N * 3;
    if (d.o.dL_drotations) s.o.dL_drotations = d.o.dL_drotations + b * N * 4;
    if (d.o.dL_dcov3D) s.o.dL_dcov3D = d.o.dL_dcov3D + b * N * 6;
    return s;
}

}  // namespace dm4d

using namespace dm4d;

extern "C" {

size_t dm4d_views_geom_bytes(int32_t B, int32_t N, int32_t H, int32_t W) { return (size_t)B * geom_layout(N, H, W).total; }
size_t dm4d_views_binning_bytes(int32_t B, int64_t capacity) { return (size_t)B * binning_bytes(capacity); }
size_t dm4d_views_image_bytes(int32_t B, int32_t H, int32_t W) { return (size_t)B * image_bytes(H, W); }
size_t dm4d_views_grad_bytes(int32_t B, int64_t record_capacity) { return (size_t)B * grad_bytes(record_capacity, 6); }
size_t dm4d_views_skin_scratch_bytes(int32_t B, int32_t V, int32_t K) { return (size_t)B * dm4d_skin_scratch_bytes(V, K); }
size_t dm4d_views_face_scratch_bytes(int32_t B, int32_t F) { return (size_t)B * dm4d_face_scratch_bytes(F); }

int dm4d_views_forward(const dm4d_views *v, dm4d_stream_t stream)
{
    int rc = views_check(v);
    if (rc) return rc;
    if (!v->out_color || !v->out_depth || !v->out_alpha) { set_error("null output image"); return DM4D_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    const int NF = v->frame_index ? v->n_frames : v->B;     // skinning + face transform once per frame
    rc = skin_forward_launch(NF, v->method, v->V, v->M, v->K, v->verts, v->nbr_idx, v->nbr_w, v->dx, v->dr, v->ds,
                             v->d_opacity, v->vxyz, v->vrot, st);
    if (rc) return rc;
    const BatchDesc d = views_batch(v);
    // (the face kernel's first workgroup also clears the B views' counters for K1: one launch less on the serial chain)
    rc = face_forward_launch(NF, v->F, v->G, v->V, v->faces, v->vxyz, v->vrot, v->q_static, v->means3D, v->rotations,
                             v->colors + 3, 6, v->rgb, v->colors, st, d.geom, d.geom_stride, d.B);
    if (rc) return rc;
    if ((rc = launch_preprocess(d, st))) return rc;
    if ((rc = launch_colscan(d, st))) return rc;
    if ((rc = launch_scatter(d, st))) return rc;
    return launch_sort_and_forward(d, st);
}

static int views_backward_impl(const dm4d_views *v, const dm4d_views_grads *gr, dm4d_stream_t stream, bool rgb_only);

int dm4d_views_backward(const dm4d_views *v, const dm4d_views_grads *gr, dm4d_stream_t stream)
{
    return views_backward_impl(v, gr, stream, false);
}

/* dL_dcolor's channels 3..5 (the normal pass) are declared zero and not read: dm4d.h */
int dm4d_views_backward_rgb(const dm4d_views *v, const dm4d_views_grads *gr, dm4d_stream_t stream)
{
    return views_backward_impl(v, gr, stream, true);
}

static int views_backward_impl(const dm4d_views *v, const dm4d_views_grads *gr, dm4d_stream_t stream, bool rgb_only)
{
    int rc = views_check(v);
    if (rc) return rc;
    if (!gr || !gr->dL_dcolor || !gr->grad_scratch || !gr->skin_scratch || !gr->face_scratch || !gr->node_csr_offsets ||
        !gr->node_csr_items || !gr->vert_csr_offsets || !gr->vert_csr_items || !gr->dL_dvxyz || !gr->dL_dvrot || !gr->dL_ddx || !gr->dL_ddr) {
        set_error("null tensor in dm4d_views_grads");
        return DM4D_ERR_INVALID;
    }
    // The per-VIEW Gaussian gradients are optional as a group: all three NULL = the caller only wants what flows on to the
    // vertices / nodes, and B2 + the face kernel run as one (gather_face.hip: nothing is written per view, dL_dmeans2D too
    // becomes optional); all three set = the two-kernel path that materialises them.
    const int n_view_grads = (gr->dL_dmeans3D ? 1 : 0) + (gr->dL_drotations ? 1 : 0) + (gr->dL_dcolors ? 1 : 0);
    if (n_view_grads != 0 && n_view_grads != 3) { set_error("dL_dmeans3D, dL_drotations, dL_dcolors: all or none"); return DM4D_ERR_INVALID; }
    const bool fused_face = n_view_grads == 0;
    if (fused_face && (gr->dL_dopacity || gr->dL_dscales)) { set_error("dL_dopacity / dL_dscales need the per-view gradient tensors"); return DM4D_ERR_INVALID; }
    if (!fused_face && !gr->dL_dmeans2D) { set_error("null dL_dmeans2D"); return DM4D_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    BatchDesc d = views_batch(v);
    d.dL_dcolor = gr->dL_dcolor; d.dL_ddepth = gr->dL_ddepth; d.dL_dalpha = gr->dL_dalpha;
    d.dLq = (float *)gr->grad_scratch; d.dlq_stride = grad_bytes(v->record_capacity, 6) / 4;
    d.o = BwdOutputs{gr->dL_dmeans2D, gr->dL_dmeans3D, gr->dL_dopacity, gr->dL_dcolors, nullptr, gr->dL_dscales,
                     gr->dL_drotations, nullptr};
    // static appearance frozen (the dynamic stage, static_learnable = False, C/geometry/dynamic_sugar.py:79-87): the
    // blend backward neither reduces nor records dL/dopacity and dL/d(rgb), 9 values per record instead of 13
    d.lean = gr->dL_dopacity ? 0 : 1;
    // ... and without a depth gradient (no depth loss in the shipped dynamic configuration) 8 values: 32-byte records
    static const bool no_lean2 = getenv("DM4D_NO_LEAN2") != nullptr;          // (A/B switch)
    // (the switch never applies to the rgb-only call: that call IS the 32-byte-record path, and a caller that selected it from the
    //  conditions above -- views.py, step.py -- must not get DM4D_ERR_UNSUPPORTED from a probe switch)
    if (d.lean == 1 && !gr->dL_ddepth && !d.tile_records && (!no_lean2 || rgb_only)) d.lean = 2;
    if (rgb_only) {
        // no loss reads the normal image (every normal weight 0 in C/configs/sugar_dynamic_dg.yaml:145-157; in the reference autograd
        // then never enters the normal pass's backward, C/renderer/diff_sugar_rasterizer_temporal.py:202-211): 5 per-entry sums, not 8
        if (d.lean != 2) {
            set_error("dm4d_views_backward_rgb: needs the static appearance frozen (dL_dopacity NULL), no depth gradient and cell records");
            return DM4D_ERR_UNSUPPORTED;
        }
        d.lean = 3;
    }
    // The blend backward writes one record per (Gaussian, cell) and the gather reads them back: in groups of views whose
    // records fit the 256 MB memory-side cache the round trip stays off HBM (8 views at once: 466 MB).
    static const int group_env = getenv("DM4D_BWD_GROUP") ? atoi(getenv("DM4D_BWD_GROUP")) : 0;
    const int group = group_env > 0 ? group_env : v->B;
    for (int b0 = 0; b0 < v->B; b0 += group) {
        BatchDesc sb = sub_batch(d, b0, v->B - b0 < group ? v->B - b0 : group);
        if (getenv("DM4D_BWD_REUSE")) sb.dLq = d.dLq;      // experiment: every group writes its records to the same scratch
        if ((rc = launch_render_bwd(sb, st))) return rc;
        if (!fused_face && (rc = launch_gather_bwd(sb, st))) return rc;
    }
    const int NF = v->frame_index ? v->n_frames : v->B;
    if (fused_face && (rc = launch_gather_face_bwd(d, NF, v->F, v->G | (v->method & 0x100), v->V, v->faces, v->vxyz, v->vrot, v->q_static,
                                                   (float *)gr->face_scratch, st))) return rc;
    rc = face_backward_launch(NF, v->F, v->G | (v->method & 0x100), v->V, v->faces, v->vxyz, v->vrot, v->q_static, gr->dL_dmeans3D,
                              gr->dL_drotations, fused_face ? nullptr : gr->dL_dcolors + 3, 6, gr->vert_csr_offsets, gr->vert_csr_items,
                              (float *)gr->face_scratch, gr->dL_dvxyz_ext, gr->dL_dvrot_ext, gr->dL_dvxyz, gr->dL_dvrot,
                              v->frame_index, fused_face ? -(v->B + 1) : v->B, st);      // (fused: per-VIEW corner records, face_backward_launch)
    if (rc) return rc;
    return skin_backward_launch(NF, v->method, v->V, v->M, v->K, v->verts, v->nbr_idx, v->nbr_w, v->dx, v->dr, v->ds,
                                v->d_opacity, gr->dL_dvxyz, gr->dL_dvrot, gr->node_csr_offsets, gr->node_csr_items,
                                (float *)gr->skin_scratch, gr->dL_ddx, gr->dL_ddr, gr->dL_dds, gr->dL_ddo, st);
}

// ---------------------------------------------------------------------------------- B views of one set of Gaussians
static int gviews_check(const dm4d_gviews *v)
{
    if (!v) { set_error("null gviews"); return DM4D_ERR_INVALID; }
    if (v->B <= 0 || v->B > 65535) { set_error("bad batch size %d", v->B); return DM4D_ERR_INVALID; }
    if (v->N <= 0 || v->N > (1 << kGidBits)) { set_error("N = %d: 1 .. %d Gaussians per view", v->N, 1 << kGidBits); return DM4D_ERR_UNSUPPORTED; }
    if (v->image_height <= 0 || v->image_width <= 0) { set_error("bad image size"); return DM4D_ERR_INVALID; }
    if ((int64_t)((v->image_height + kTile - 1) / kTile) * ((v->image_width + kTile - 1) / kTile) > kMaxTiles) {
        set_error("image has more than %d tiles", kMaxTiles);
        return DM4D_ERR_UNSUPPORTED;
    }
    if (v->capacity <= 0 || v->capacity > 0xFFFFFFF0ll) { set_error("capacity out of range"); return DM4D_ERR_INVALID; }
    if (v->record_capacity <= 0 || v->record_capacity > 0xFFFFFFF0ll) { set_error("record_capacity out of range"); return DM4D_ERR_INVALID; }
    if (v->record_mode != DM4D_RECORDS_CELL && v->record_mode != DM4D_RECORDS_TILE) { set_error("bad record_mode %d", v->record_mode); return DM4D_ERR_INVALID; }
    if (!v->bg || !v->viewmatrix || !v->projmatrix || !v->means3D || !v->rotations || !v->scales || !v->opacities || !v->colors ||
        !v->radii || !v->geom || !v->binning || !v->image) {
        set_error("null tensor in dm4d_gviews");
        return DM4D_ERR_INVALID;
    }
    return DM4D_OK;
}

static BatchDesc gviews_batch(const dm4d_gviews *v)
{
    BatchDesc d;
    memset(&d, 0, sizeof(d));
    d.B = v->B; d.N = v->N; d.C = 6; d.W = v->image_width; d.H = v->image_height;
    d.tanfovx = v->tanfovx; d.tanfovy = v->tanfovy; d.scale_modifier = v->scale_modifier;
    d.bg = v->bg;
    d.view = v->viewmatrix; d.proj = v->projmatrix; d.cam_stride = 16;
    d.means3D = v->means3D; d.rotations = v->rotations; d.colors = v->colors; d.scales = v->scales; d.opacities = v->opacities;   // strides 0: shared
    d.radii = v->radii; d.radii_stride = (size_t)v->N;
    d.geom = (char *)v->geom; d.geom_stride = geom_layout(v->N, v->image_height, v->image_width).total;
    d.binning = (char *)v->binning; d.bin_stride = binning_bytes(v->capacity); d.cap = (uint32_t)v->capacity;
    d.rec_cap = (uint32_t)v->record_capacity;
    d.tile_records = v->record_mode == DM4D_RECORDS_TILE ? 1 : 0;
    d.image = (char *)v->image; d.img_stride = image_bytes(v->image_height, v->image_width);
    d.out_color = v->out_color; d.out_depth = v->out_depth; d.out_alpha = v->out_alpha;
    return d;
}

int dm4d_gviews_forward(const dm4d_gviews *v, dm4d_stream_t stream)
{
    int rc = gviews_check(v);
    if (rc) return rc;
    if (!v->out_color || !v->out_depth || !v->out_alpha) { set_error("null output image"); return DM4D_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    const BatchDesc d = gviews_batch(v);
    if ((rc = launch_zero_counters(d, st))) return rc;
    if ((rc = launch_preprocess(d, st))) return rc;
    if ((rc = launch_colscan(d, st))) return rc;
    if ((rc = launch_scatter(d, st))) return rc;
    return launch_sort_and_forward(d, st);
}

int dm4d_gviews_backward(const dm4d_gviews *v, const dm4d_gviews_grads *gr, dm4d_stream_t stream)
{
    int rc = gviews_check(v);
    if (rc) return rc;
    if (!gr || !gr->dL_dcolor || !gr->grad_scratch || !gr->dL_dmeans2D || !gr->dL_dmeans3D || !gr->dL_drotations || !gr->dL_dscales ||
        !gr->dL_dopacity || !gr->dL_dcolors) {
        set_error("null tensor in dm4d_gviews_grads");
        return DM4D_ERR_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    BatchDesc d = gviews_batch(v);
    d.dL_dcolor = gr->dL_dcolor; d.dL_ddepth = gr->dL_ddepth; d.dL_dalpha = gr->dL_dalpha;
    d.dLq = (float *)gr->grad_scratch; d.dlq_stride = grad_bytes(v->record_capacity, 6) / 4;
    d.o = BwdOutputs{gr->dL_dmeans2D, gr->dL_dmeans3D, gr->dL_dopacity, gr->dL_dcolors, nullptr, gr->dL_dscales, gr->dL_drotations, nullptr};
    d.lean = 0;                                 // every appearance gradient: the static stage learns them all
    if ((rc = launch_render_bwd(d, st))) return rc;
    return launch_gather_bwd(d, st);
}

/* per view: num_rendered, num_records, overflow flags (synchronises the stream). */
int dm4d_views_counters(const dm4d_views *v, int64_t *num_rendered, int64_t *num_records, int32_t *overflowed,
                        dm4d_stream_t stream)
{
    if (!v || !v->geom) { set_error("null views"); return DM4D_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    const size_t stride = geom_layout(v->N, v->image_height, v->image_width).total;
    std::vector<uint32_t> tmp((size_t)v->B * 4);
    for (int b = 0; b < v->B; ++b)
        DM4D_HIP_CHECK(hipMemcpyAsync(&tmp[4 * b], (const char *)v->geom + b * stride, 16, hipMemcpyDeviceToHost, st));
    DM4D_HIP_CHECK(hipStreamSynchronize(st));
    for (int b = 0; b < v->B; ++b) {
        if (num_rendered) num_rendered[b] = tmp[4 * b + kCntD];
        if (num_records) num_records[b] = tmp[4 * b + kCntR];
        if (overflowed) overflowed[b] = (tmp[4 * b + kCntOverflow] ? 1 : 0) | (tmp[4 * b + kCntRecOverflow] ? 2 : 0);
    }
    return DM4D_OK;
}

}  // extern "C"
