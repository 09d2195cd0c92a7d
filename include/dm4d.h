/*
 * dm4d.h -- C ABI of libdm4d_hip.so, the MI355X (gfx950) hot path of DreamMesh4D's
 * dynamic stage.  Plain pointers and sizes only; every pointer marked [dev] is a
 * DEVICE pointer (HBM) owned by the caller, every call is enqueued on the caller's
 * hipStream_t (passed as void*), no call allocates device memory, and the library
 * keeps no global device state (re-entrant per stream).
 *
 * Each entry point names the reference interface it replaces
 * (paths relative to the DreamMesh4D tree, C/ = custom/threestudio-dreammesh4d/):
 *
 *   dm4d_rasterize_*      diff_gaussian_rasterization.GaussianRasterizer (un-vendored CUDA
 *                         package, requirements.txt:49) as called at
 *                         C/renderer/diff_sugar_rasterizer_temporal.py:129-178,202-211 and
 *                         C/renderer/diff_sugar_rasterizer_normal.py:117-132,161-170,186-195
 *   dm4d_mark_visible     GaussianRasterizer.markVisible (same package)
 *   dm4d_dist2_knn3       simple_knn._C.distCUDA2 (requirements.txt:50), call site
 *                         C/geometry/gaussian_base.py:435-438
 *   dm4d_skin_*           C/geometry/dynamic_sugar.py:408-465,487-613 (+ C/utils/dual_quaternions.py)
 *   dm4d_face_gaussians_* C/geometry/dynamic_sugar.py:657-706,726-743,877-889,330-364 and
 *                         C/geometry/sugar.py:479-518
 *
 * Return convention: >= 0 success (some calls return a count), < 0 error code;
 * dm4d_last_error() returns a thread-local description.
 */
#ifndef DM4D_H
#define DM4D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DM4D_OK 0
#define DM4D_ERR_INVALID -1    /* bad argument */
#define DM4D_ERR_HIP -2        /* a HIP runtime call failed */
#define DM4D_ERR_CAPACITY -3   /* a caller-provided buffer is too small */
#define DM4D_ERR_UNSUPPORTED -4

#define DM4D_TILE 16           /* BLOCK_X == BLOCK_Y of the reference rasterizer */

typedef void *dm4d_stream_t;   /* hipStream_t */

/* ABI version = 100 * major + round.  A binding built against this header checks dm4d_version() == DM4D_ABI_VERSION when it loads the
 * library (dreammesh4d_amd/_lib.py does).  Entry points are never changed in place from round 5 on: a new argument is a new symbol
 * (dm4d_adamw_step beside dm4d_adamw_message, dm4d_normal_consistency_backward_scratch beside dm4d_normal_consistency_backward). */
#define DM4D_ABI_VERSION 105
int dm4d_version(void);
const char *dm4d_last_error(void);
/* Number of HIP devices visible / name of device `dev` (host helpers for the loader). */
int dm4d_device_count(void);
int dm4d_device_arch(int dev, char *buf, int buflen);

/* Per-kernel timing with HIP events recorded on the launch stream (used by bench.py for the
 * roofline of the dominant kernel).  kernel ids: 0 preprocess, 1 colscan, 2 scatter,
 * 3 tile_sort, 4 render_fwd, 5 render_bwd, 6 gather_bwd, 7 skin_fwd, 8 skin_bwd, 9 face_fwd,
 * 10 face_bwd, 11 knn.  Process-global debug facility; off by default. */
void dm4d_profile_enable(unsigned kernel_mask);
int64_t dm4d_profile_collect(int kernel_id, double *total_ms);

/* ------------------------------------------------------------------ rasterizer */

/* GaussianRasterizationSettings (C/renderer/diff_sugar_rasterizer_temporal.py:129-142).
 * Matrices are in the reference's row-vector ("transposed") convention
 * (threestudio/utils/ops.py:398-413): x' = M[0]*x + M[4]*y + M[8]*z + M[12]. */
typedef struct dm4d_raster_settings {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    int32_t sh_degree;          /* only 0 is implemented (the reference never uses more) */
    int32_t prefiltered;
    int32_t debug;
    const float *bg;            /* [dev] [3]  */
    const float *viewmatrix;    /* [dev] [16] */
    const float *projmatrix;    /* [dev] [16] */
    const float *campos;        /* [dev] [3]  */
} dm4d_raster_settings;

/* Per-Gaussian inputs of one rasterizer call.  Exactly one of (shs | colors_precomp)
 * and one of ((scales, rotations) | cov3D_precomp), as upstream. */
typedef struct dm4d_raster_inputs {
    int32_t N;
    int32_t sh_coeffs;            /* M of shs[N,M,3]; 0 when colors_precomp is used */
    int32_t n_channels;           /* colour channels C of colors_precomp: 3 (0 means 3) or 6.  C = 6 blends the
                                     RGB pass and the normal pass of one view together (identical geometry,
                                     C/renderer/diff_sugar_rasterizer_temporal.py:169-178 + 202-211): colours are
                                     [N,6], out_color / dL_dcolor [6,H,W], dL_dcolors [N,6]. */
    const float *means3D;         /* [dev] [N,3] */
    const float *shs;             /* [dev] [N,M,3] or NULL */
    const float *colors_precomp;  /* [dev] [N,C] or NULL */
    const float *opacities;       /* [dev] [N]   */
    const float *scales;          /* [dev] [N,3] or NULL */
    const float *rotations;       /* [dev] [N,4] (w,x,y,z) or NULL */
    const float *cov3D_precomp;   /* [dev] [N,6] or NULL */
} dm4d_raster_inputs;

/* Workspace sizes (bytes).  geom: per-Gaussian + per-tile state; binning: per
 * (Gaussian,tile) duplicate, `capacity` duplicates; image: per-pixel state;
 * grad: backward scratch for `n_records` (Gaussian, 4x4-pixel cell) records
 * (dm4d_rasterize_num_records). */
size_t dm4d_raster_geom_bytes(int32_t N, int32_t image_height, int32_t image_width);
size_t dm4d_raster_binning_bytes(int64_t capacity);
size_t dm4d_raster_image_bytes(int32_t image_height, int32_t image_width);
size_t dm4d_raster_grad_bytes(int64_t n_records, int32_t n_channels);

/* Stage 1 (no host sync): preprocess every Gaussian, count duplicates per tile, scan.
 * Writes radii[N] and the geom workspace (which must be 16-byte aligned; it need not be
 * zeroed).  After it the number of duplicates D ("num_rendered") is in device memory. */
int dm4d_rasterize_prepare(const dm4d_raster_settings *s, const dm4d_raster_inputs *in,
                           int32_t *radii /* [dev] [N] */, void *geom /* [dev] */, size_t geom_bytes,
                           dm4d_stream_t stream);

/* Host read of D: synchronises `stream` (this is the one sync upstream also has). */
int64_t dm4d_rasterize_num_rendered(const void *geom /* [dev] */, dm4d_stream_t stream);
/* Host read of R, the number of backward records (sum over the Gaussians of the 4x4-pixel cells their
 * alpha >= 1/255 support reaches); known after prepare, like D.  Synchronises `stream`. */
int64_t dm4d_rasterize_num_records(const void *geom /* [dev] */, dm4d_stream_t stream);
/* Both with one synchronisation. */
int dm4d_rasterize_counts(const void *geom /* [dev] */, int64_t *num_rendered, int64_t *num_records,
                          dm4d_stream_t stream);

/* Stage 2: scatter duplicates into per-tile segments, sort every tile by
 * (depth bits, Gaussian id) -- the order of a stable radix sort of tile<<32|depth --,
 * blend front to back.  `binning` must hold `capacity` >= D duplicates; if it does not,
 * nothing is rendered past the capacity and the overflow flag read by
 * dm4d_rasterize_overflowed() is set.  Outputs: color [C,H,W], depth [H,W], alpha [H,W]. */
int dm4d_rasterize_render(const dm4d_raster_settings *s, const dm4d_raster_inputs *in,
                          const int32_t *radii /* [dev] [N], from prepare */,
                          void *geom, void *binning, int64_t capacity, void *image,
                          float *out_color, float *out_depth, float *out_alpha, dm4d_stream_t stream);

/* 1 if the last render on this geom workspace dropped duplicates (syncs the stream). */
int dm4d_rasterize_overflowed(const void *geom, dm4d_stream_t stream);

/* Backward of prepare+render.  dL_ddepth / dL_dalpha may be NULL (treated as zero).
 * Output gradient pointers may be NULL when not wanted, except dL_dmeans2D and
 * dL_dmeans3D.  `grad` is scratch of dm4d_raster_grad_bytes(record_capacity), record_capacity >= R.
 * Deterministic: no floating-point atomics anywhere. */
int dm4d_rasterize_backward(const dm4d_raster_settings *s, const dm4d_raster_inputs *in,
                            const int32_t *radii, const void *geom, const void *binning, int64_t capacity,
                            const void *image, void *grad, int64_t record_capacity,
                            const float *dL_dcolor /* [C,H,W] */, const float *dL_ddepth /* [H,W] */,
                            const float *dL_dalpha /* [H,W] */,
                            float *dL_dmeans2D /* [N,3] */, float *dL_dmeans3D /* [N,3] */,
                            float *dL_dopacity /* [N] */, float *dL_dcolors /* [N,C] */,
                            float *dL_dsh /* [N,M,3] */, float *dL_dscales /* [N,3] */,
                            float *dL_drotations /* [N,4] */, float *dL_dcov3D /* [N,6] */,
                            dm4d_stream_t stream);

/* One-shot forward in the shape of upstream's RasterizeGaussiansCUDA: the library asks the
 * caller for the three workspaces through `alloc(ctx, which, bytes)` (which = 0 geom,
 * 1 binning, 2 image; must return a 16-byte aligned device pointer that stays valid until
 * the backward) and returns num_rendered.  Synchronises the stream once (to size binning). */
typedef void *(*dm4d_alloc_fn)(void *ctx, int which, size_t bytes);
int64_t dm4d_rasterize_forward(const dm4d_raster_settings *s, const dm4d_raster_inputs *in,
                               float *out_color, float *out_depth, float *out_alpha, int32_t *radii,
                               dm4d_alloc_fn alloc, void *alloc_ctx, dm4d_stream_t stream);

/* Debug / parity views into the workspaces (device-to-host copies, sync the stream). */
int dm4d_raster_read_sorted(const void *geom, const void *binning, int32_t N, int32_t image_height,
                            int32_t image_width, int64_t D, uint64_t *keys /* [host] [D] */,
                            uint32_t *values /* [host] [D] */, uint32_t *ranges /* [host] [tiles,2] */,
                            dm4d_stream_t stream);
int dm4d_raster_read_geom(const void *geom, int32_t N, int32_t image_height, int32_t image_width,
                          float *xy /* [host][N,2] */, float *depths /* [N] */,
                          float *conic_opacity /* [N,4] */, uint32_t *tiles_touched /* [N] */,
                          dm4d_stream_t stream);
int dm4d_raster_read_image_state(const void *geom, const void *binning, const void *image, int32_t N, int32_t image_height,
                                 int32_t image_width, int64_t D /* capacity given to dm4d_rasterize_render */,
                                 uint32_t *n_contrib /* [host][H,W]: upstream's meaning, last contributor's position in the
                                                        tile list + 1 (translated from the cell-list positions kept on device) */,
                                 float *final_T /* [host][H,W] */, dm4d_stream_t stream);

/* Debug: when `trace` (device, uint64 [blocks, 4]) is non-NULL every wave of the blend kernels records
 * {start, end} in 100 MHz ticks, {XCC_ID << 32 | HW_ID} and its iteration count at index
 * blockIdx.y * gridDim.x + blockIdx.x.  NULL switches it off (the default).  min_work > 0 makes waves
 * whose longest list is shorter exit at once (isolates the long ones; results are then incomplete). */
int dm4d_debug_trace(void *trace, uint32_t min_work);
/* Debug: per-tile phase timestamps of the tile sort, uint64 [views * tiles, 5] = {start, binned, sorted, end, n}. */
int dm4d_debug_sort_trace(void *trace);

/* markVisible: present[i] = view-space z > 0.2 */
int dm4d_mark_visible(int32_t N, const float *means3D, const float *viewmatrix, uint8_t *present,
                      dm4d_stream_t stream);

/* ------------------------------------------------------------------ Zero123 SDS step: GroupNorm (+ SiLU) over NHWC activations */

/* torch.nn.GroupNorm followed (silu != 0) by SiLU, for activations stored channels-last -- what
 * extern/ldm_zero123/modules/diffusionmodules/util.py:242-244 (GroupNorm32), openaimodel.py:259-275 (ResBlock
 * "GroupNorm, SiLU, conv") and diffusionmodules/model.py (Normalize, nonlinearity) evaluate ~100 times per SDS step:
 *     y[n, p, c] = act( (x[n, p, c] + add[n, c] - mean[n, g]) * rstd[n, g] * gamma[c] + beta[c] ),   g = c / (C / G)
 * x, y [N, HW, C]; add [N, C] or NULL (the ResBlock's timestep-embedding term); gamma, beta [C]; all of `dtype`
 * (DM4D_GN_F16: 16-bit floats, C % 8 == 0; DM4D_GN_F32: C % 4 == 0); statistics in float32 over the C / G channels of a
 * group and all HW positions, biased variance, as torch.  stats [N, G, 2] (out): mean, rstd -- what backward needs.
 * scratch [N, splits, G, 2] floats, uninitialised; 1 <= splits <= DM4D_GN_MAX_SPLITS = workgroups per sample (each
 * owns ceil(HW / splits) consecutive positions).  Results do not depend on run-to-run scheduling (no atomics). */
#define DM4D_GN_F16 0
#define DM4D_GN_F32 1
#define DM4D_GN_MAX_SPLITS 128
int dm4d_groupnorm_nhwc_forward(int32_t N, int32_t HW, int32_t C, int32_t G, int32_t dtype, const void *x, const void *add,
                                int32_t add_stride /* >= C (a multiple of 8): add is [N, C], rows add_stride elements apart; 0: add is [C], the same for every sample */,
                                const void *gamma, const void *beta, float eps, int32_t silu, void *y, float *stats,
                                float *scratch, int32_t splits, dm4d_stream_t stream);
/* dL/dx of the above for frozen gamma / beta and a constant `add` (the guidance model is not trained): x, add as given to
 * forward, stats from forward, dy / dx [N, HW, C]. */
int dm4d_groupnorm_nhwc_backward(int32_t N, int32_t HW, int32_t C, int32_t G, int32_t dtype, const void *x, const void *add,
                                 int32_t add_stride, const void *gamma, const void *beta, const float *stats, int32_t silu,
                                 const void *dy, void *dx, float *scratch, int32_t splits, dm4d_stream_t stream);
/* The same with dx += dx_add [N, HW, C] (NULL: nothing added): the gradient that reaches x beside the norm -- the residual
 * connection of a ResnetBlock (diffusionmodules/model.py:121-145: x + h(norm1(x))) -- added where dx is written instead of by a
 * launch of its own (autograd's accumulation: 13 adds of up to 67 MB per VAE encoder backward). */
int dm4d_groupnorm_nhwc_backward_add(int32_t N, int32_t HW, int32_t C, int32_t G, int32_t dtype, const void *x, const void *add,
                                     int32_t add_stride, const void *gamma, const void *beta, const float *stats, int32_t silu,
                                     const void *dy, const void *dx_add, void *dx, float *scratch, int32_t splits,
                                     dm4d_stream_t stream);

/* The arithmetic between the networks of a Zero123 SDS step (guidance/temporal_stable_zero123_guidance.py:299-374) as two launches;
 * csrc/sds_glue.hip has the expressions.  Every tensor comes with its four element strides (batch, channel, row, column): the
 * layouts are whatever the networks' last layers produce.  float16: moments [B,8,H,W], post (posterior noise) [B,4,H,W], c_concat
 * [L,4,H,W], x_in [2B,8,H,W], pred [2B,4,H,W], d_moments [B,8,H,W]; float32: noise, latents [B,4,H,W], alphas_cumprod [T], clip
 * (one number or NULL), loss, grad_norm; int64: t [B], frame_index [B], t2 [2B].
 *   prepare: latents = scale_factor (mean + exp(0.5 clamp(logvar, -30, 20)) post); noisy = sqrt(ac[t]) latents + sqrt(1 - ac[t]) noise;
 *            x_in = [noisy | 0] for the first B samples, [noisy | c_concat[frame_index]] for the second B; t2 = [t, t].
 *   finish : pred = uncond + guidance_scale (cond - uncond); grad = clip(nan_to_num((1 - ac[t]) (pred - noise)));
 *            loss = 0.5 sum (latents - (latents - grad))^2 / B; grad_norm = |grad|; d_moments = dloss/dmoments. */
int dm4d_sds_prepare(int32_t B, int32_t H, int32_t W, float scale_factor, const void *moments, const int64_t *moments_strides,
                     const void *post, const int64_t *post_strides, const float *noise, const int64_t *noise_strides, float *latents,
                     const int64_t *latents_strides, const int64_t *t, const float *alphas_cumprod, const void *c_concat,
                     const int64_t *c_concat_strides, const int64_t *frame_index, void *x_in, const int64_t *x_in_strides, int64_t *t2,
                     dm4d_stream_t stream);
int dm4d_sds_finish(int32_t B, int32_t H, int32_t W, float scale_factor, float guidance_scale, const void *pred,
                    const int64_t *pred_strides, const float *latents, const int64_t *latents_strides, const float *noise,
                    const int64_t *noise_strides, const int64_t *t, const float *alphas_cumprod, const float *clip, const void *moments,
                    const int64_t *moments_strides, const void *post, const int64_t *post_strides, void *d_moments,
                    const int64_t *d_moments_strides, float *loss, float *grad_norm, dm4d_stream_t stream);

/* y[r, c] = a[r, c] + b[r, c] + bias[c] over [rows, C] (the end of a ResBlock: skip + convolution output + that convolution's
 * bias -- openaimodel.py:259-275, diffusionmodules/model.py ResnetBlock); dtype as above, C % 8 == 0 (F16) / % 4 (F32). */
int dm4d_add_bias_nhwc(int64_t rows, int32_t C, int32_t dtype, const void *a, const void *b, const void *bias, void *y,
                       dm4d_stream_t stream);
/* GEGLU (extern/ldm_zero123/modules/attention.py:48-56): y[r, d] = proj[r, d] * gelu(proj[r, D + d]), exact (erf) GELU;
 * proj [rows, 2 D], y [rows, D]. */
int dm4d_geglu(int64_t rows, int32_t D, int32_t dtype, const void *proj, void *y, dm4d_stream_t stream);
/* Residual add + LayerNorm of float16 rows (the UNet's transformer blocks, extern/ldm_zero123/modules/attention.py:196-213, without
 * gradients):  s = x[r] (+ tok[r / rows_per_sample], a row per sample broadcast over its positions: the single-token
 * cross-attention);  normed[r] = LayerNorm(s) gamma + beta (float32 statistics);  xb[r] = s (+ bias2) -- the residual operand of
 * the next GEMM with that GEMM's output bias already added.  tok, bias2, xb may be NULL.  C % 8 == 0, C <= 2048. */
int dm4d_add_layernorm_f16(int64_t rows, int32_t C, int32_t rows_per_sample, const void *x, const void *tok, const void *gamma,
                           const void *beta, float eps, const void *bias2, void *normed, void *xb, dm4d_stream_t stream);

/* ------------------------------------------------------------------ simple-knn */

/* simple_knn._C.distCUDA2 (C/geometry/gaussian_base.py:435-438): out[i] = mean of the squared distances
 * from point i to its 3 nearest OTHER points (self excluded by index).  points [N,3], out [N]. */
int dm4d_dist2_knn3(int32_t N, const float *points, float *out, dm4d_stream_t stream);
/* The same values (bit for bit) through upstream's structure -- Morton-ordered boxes of 1024 points, a point only visits
 * the boxes that can hold something closer than its current third-best (simple-knn's coord2Morton / boxMinMax /
 * boxMeanDist) -- for large clouds: dm4d_dist2_knn3 above is an O(N^2) exhaustive search.  `scratch` [dev]:
 * dm4d_knn_scratch_bytes(N) bytes, 256-byte aligned, uninitialised. */
size_t dm4d_knn_scratch_bytes(int32_t N);
int dm4d_dist2_knn3_ws(int32_t N, const float *points, float *out, void *scratch, size_t scratch_bytes, dm4d_stream_t stream);

/* ------------------------------------------------------------------ skinning / face -> Gaussians */

/* Gradient convention of the skinning / face->Gaussian BACKWARD entries: the default is the exact (Euclidean) gradient of the
 * forward function; with DM4D_GRAD_PYPOSE or-ed into `method` (dm4d_skin_vertices_backward, dm4d_views.method) or into `G`
 * (dm4d_face_gaussians_backward) the rotation operations return what the reference's pypose LieTensor autograd returns
 * (SO3 Log / Act / Mul, so3 Exp: left-perturbation tangent gradients zero-padded into the quaternion storage,
 * C/geometry/dynamic_sugar.py:461,530-586,669-676,877-889) -- the gradient the reference actually trains its rotation
 * head with.  Restated from pypose 0.6.7's published rules (the package is not in the tree): parity unpinned.  The two
 * SO3 products of the DQS branch's dual-quaternion algebra (C/utils/dual_quaternions.py:115-131,184-231, NON-unit
 * operands) follow the same SO3_Mul rule in this mode (X: (g[:3], 0), Y: (g[:3] Adj(X), 0)). */
#define DM4D_GRAD_PYPOSE 0x100

/* Sparse-control skinning of the V mesh vertices by M deformation-graph nodes, K neighbours each
 * (C/geometry/dynamic_sugar.py:408-465 node attributes, :487-613 vertex skinning;
 * C/utils/dual_quaternions.py:94-131,184-197,224-231).  method: 0 = "lbs", 1 = "dqs", 2 = "hybrid"
 * (C/configs/sugar_dynamic_dg.yaml:86).  Inputs are the RAW outputs of the deformation network for
 * one timestamp: dx [M,3] translation, dr [M,4] rotation delta (x,y,z,w; the kernel adds the identity
 * and normalises), ds [M,6] strain (lbs/hybrid), d_opacity [M] logit (hybrid).  nbr_idx [V,K] int32,
 * nbr_w [V,K] row-normalised.  Outputs: xyz [V,3], rot [V,4] (x,y,z,w unit quaternion =
 * Exp(sum_k w_k Log q_k)). */
int dm4d_skin_vertices_forward(int32_t method, int32_t V, int32_t M, int32_t K, const float *verts,
                               const int32_t *nbr_idx, const float *nbr_w, const float *dx, const float *dr,
                               const float *ds, const float *d_opacity, float *out_xyz, float *out_rot,
                               dm4d_stream_t stream);
size_t dm4d_skin_scratch_bytes(int32_t V, int32_t K);
/* The `d_scale: true` branch (C/geometry/dynamic_sugar.py:593-611, 697-704; csrc/dscale.hip).  Vertex scale matrices
 * out [n_frames, V, 9] (row-major 3x3) from the strain head ds [n_frames, M, 6] (and, hybrid, the opacity logits
 * d_opacity [n_frames, M]); method 0 = lbs, 2 = hybrid (the reference defines no vertex scale for dqs).  Gaussian scales
 * out [n_frames, F*G, 3] = (sum_c bary[g, c] vertex_scales[faces[f, c]]) scaling[f*G + g]; bary [G, 3]; G <= 6.
 * Backward: node_csr_* as above (item = v*K + k), vert_csr_* the static inverse of faces (item = face*3 + corner);
 * dL_ddo / dL_dvertex_scales / dL_dscaling may be NULL. */
int dm4d_vertex_scales_forward(int32_t method, int32_t n_frames, int32_t V, int32_t M, int32_t K, const int32_t *nbr_idx, const float *nbr_w,
                               const float *ds, const float *d_opacity, float *out, dm4d_stream_t stream);
int dm4d_vertex_scales_backward(int32_t method, int32_t n_frames, int32_t V, int32_t M, int32_t K, const int32_t *nbr_idx, const float *nbr_w,
                                const float *ds, const float *d_opacity, const int32_t *node_csr_offsets, const int32_t *node_csr_items,
                                const float *dL_dout, float *dL_dds, float *dL_ddo, dm4d_stream_t stream);
int dm4d_gaussian_scales_forward(int32_t n_frames, int32_t F, int32_t G, int32_t V, const int32_t *faces, const float *bary,
                                 const float *vertex_scales, const float *scaling, float *out, dm4d_stream_t stream);
int dm4d_gaussian_scales_backward(int32_t n_frames, int32_t F, int32_t G, int32_t V, const int32_t *faces, const float *bary,
                                  const float *vertex_scales, const float *scaling, const int32_t *vert_csr_offsets,
                                  const int32_t *vert_csr_items, const float *dL_dout, float *dL_dvertex_scales, float *dL_dscaling,
                                  dm4d_stream_t stream);
/* Backward.  node_csr_* is the static inverse of nbr_idx: node m is referenced by the items
 * node_csr_items[node_csr_offsets[m] .. node_csr_offsets[m+1]) where item = v*K + k.
 * dL_dxyz / dL_drot may be NULL (zero).  Output pointers may be NULL when not wanted.
 * Deterministic (gather formulation, no atomics). */
int dm4d_skin_vertices_backward(int32_t method, int32_t V, int32_t M, int32_t K, const float *verts,
                                const int32_t *nbr_idx, const float *nbr_w, const float *dx, const float *dr,
                                const float *ds, const float *d_opacity, const float *dL_dxyz, const float *dL_drot,
                                const int32_t *node_csr_offsets, const int32_t *node_csr_items, void *scratch,
                                float *dL_ddx, float *dL_ddr, float *dL_dds, float *dL_ddo, dm4d_stream_t stream);

/* Mesh-bound Gaussians of a deformed mesh: F faces x G Gaussians (G in {1,3,4,6}, barycentric tables of
 * C/geometry/sugar.py:235-276), face-major order.  means = sum_j b_gj x_j (C/geometry/dynamic_sugar.py:726-743);
 * rotation = normalize(Exp(sum_j b_gj Log q_j) (x) q_static) in (w,x,y,z) order (:669-676,877-889);
 * normals (optional) = unit face normal of the deformed mesh repeated per Gaussian (:330-364). */
int dm4d_face_gaussians_forward(int32_t F, int32_t G, const int32_t *faces, const float *vxyz, const float *vrot,
                                const float *q_static_wxyz, float *means, float *rotations_wxyz, float *normals,
                                dm4d_stream_t stream);
size_t dm4d_face_scratch_bytes(int32_t F);
/* Backward.  vert_csr_*: vertex v is corner (item % 3) of face (item / 3) for the items
 * vert_csr_items[vert_csr_offsets[v] .. vert_csr_offsets[v+1]).  Any of the three upstream gradients
 * may be NULL.  Outputs dL_dvxyz [V,3], dL_dvrot [V,4] (x,y,z,w). */
int dm4d_face_gaussians_backward(int32_t F, int32_t G, int32_t V, const int32_t *faces, const float *vxyz,
                                 const float *vrot, const float *q_static_wxyz, const float *dL_dmeans,
                                 const float *dL_drotations_wxyz, const float *dL_dnormals,
                                 const int32_t *vert_csr_offsets, const int32_t *vert_csr_items, void *scratch,
                                 float *dL_dvxyz, float *dL_dvrot, dm4d_stream_t stream);

/* ------------------------------------------------------------------ HexPlane feature query */

/* Fused multi-scale HexPlane query of the deformation network (C/geometry/deformation.py:88-113,141-174,
 * 226-240) for M static nodes x B timestamps: feat[f, m, s*32 + c] = prod over the 6 planes of the bilinear
 * sample (align_corners, border padding).  `res` [S,4] (host) = resolution of x,y,z,t per scale;
 * `planes` = HOST array of S*6 device pointers, plane (s,p) is the reference's [1, 32, res[a1], res[a0]] parameter
 * stored either as [32][res[a1]][res[a0]] (channel_last == 0, a contiguous tensor) or as [res[a1]][res[a0]][32]
 * (channel_last != 0: the same tensor in torch.channels_last memory format -- state dicts and checkpoints are
 * unchanged, but the 32 channels of a texel are one 128-byte line instead of 32 scattered words; all planes and
 * gradient planes of a call share the layout); `aabb_host` [2,3] (host); nodes [M,3], times [B]
 * (already mapped to [-1,1]), feat [B,M,S*32] on the device.  `samples` (device, dm4d_hexplane_scratch_bytes,
 * or NULL for inference) receives the 6 plane samples of every feature; dm4d_hexplane_backward consumes it. */
int dm4d_hexplane_forward(int32_t S, int32_t M, int32_t B, const int32_t *res, const float *const *planes,
                          int32_t channel_last, const float *aabb_host, const float *nodes, const float *times,
                          float *feat, void *samples, dm4d_stream_t stream);
/* Plan helper: lower texel index of every node along x,y,z per scale, i0 [S,3,M] (device), computed with
 * the kernels' own arithmetic; the host builds the static gather lists of the backward from it. */
int dm4d_hexplane_axis_index(int32_t S, int32_t M, const int32_t *res, const float *aabb_host, const float *nodes,
                             int32_t *i0, dm4d_stream_t stream);
size_t dm4d_hexplane_scratch_bytes(int32_t S, int32_t M, int32_t B);
/* `channel_last` of dm4d_hexplane_forward / _backward is a flag word: */
#define DM4D_HEX_CHANNELS_LAST 1   /* planes (and gradient planes) are stored [H][W][32] */
#define DM4D_HEX_KEEP_SPATIAL  2   /* backward only: the SPATIAL gradient planes (xy, xz, yz) are persistent buffers of the caller
                                    * that already hold zeros outside the texels of this plan (e.g. zero-initialised once, then only
                                    * ever written by this call with the same plan): their zero fill is skipped -- 134 of the 143 MB
                                    * a step would otherwise clear.  The time planes are cleared as usual. */
#define DM4D_HEX_TIMES_01      4   /* `times` are timestamps in [0, 1]; the kernels map them to 2 t - 1 themselves
                                    * (C/geometry/dynamic_sugar.py:431), saving the caller an elementwise launch */
/* Gradients w.r.t. the planes, written (not accumulated) into the dense planes `g_planes` = HOST array of
 * S*6 device pointers (16-byte aligned, uninitialised: the call zero-fills them in one launch and hands the
 * pointers to its kernels by value).  Atomic-free and deterministic: spatial planes
 * gather per touched texel (sp_*: scale, plane, texel, CSR of items node*4+corner), time planes per touched
 * column (tp_*: scale, plane, column, CSR of items node*2+corner). */
int dm4d_hexplane_backward(int32_t S, int32_t M, int32_t B, const int32_t *res, const float *const *planes,
                           int32_t channel_last, const float *aabb_host, const float *nodes, const float *times,
                           const float *g_feat,
                           int32_t n_spatial, const int32_t *sp_scale, const int32_t *sp_plane, const int32_t *sp_texel,
                           const int32_t *sp_off, const int32_t *sp_item, int32_t n_time, const int32_t *tp_scale,
                           const int32_t *tp_plane, const int32_t *tp_col, const int32_t *tp_off, const int32_t *tp_item,
                           void *samples /* from the forward; overwritten */, float *const *g_planes, dm4d_stream_t stream);

/* ------------------------------------------------------------------ deformation MLP (fused) */

/* The MLP behind the HexPlane features of `Deformation.forward_dynamic_delta`
 * (C/geometry/deformation.py:285-305,430-436,507-512), per row of feat [P,in_dim]:
 *   h = W0 feat + b0;  x = relu(h);  y_k = x + W1_k x + b1_k;  out_k = W2_k y_k + b2_k
 * Weights are torch nn.Linear layouts: W [out,in] row-major, b [out]; width must be 64, in_dim a
 * multiple of 64 (<= 256), out_dim[k] <= 8, up to 4 heads.  All pointers are device pointers. */
typedef struct dm4d_mlp_weights {
    int32_t in_dim, width, n_heads;
    int32_t out_dim[4];
    const float *W0, *b0;                       /* [64,in_dim] [64] */
    const float *W1[4], *b1[4];                 /* [64,64] [64]     */
    const float *W2[4], *b2[4];                 /* [out,64] [out]   */
} dm4d_mlp_weights;
typedef struct dm4d_mlp_weights_grad {          /* same shapes; any pointer may be NULL (not wanted) */
    float *W0, *b0;
    float *W1[4], *b1[4];
    float *W2[4], *b2[4];
} dm4d_mlp_weights_grad;

size_t dm4d_deform_mlp_scratch_bytes(int32_t P, int32_t in_dim, int32_t n_heads);
/* out[k] [P,out_dim[k]]; h_save [P,64] and y_save [n_heads,P,64] are kept for the backward. */
int dm4d_deform_mlp_forward(int32_t P, const float *feat, const dm4d_mlp_weights *w, float *h_save, float *y_save,
                            float *const *out /* [host] n_heads device pointers */, void *scratch,
                            dm4d_stream_t stream);
/* g_out[k] may be NULL (zero).  g_feat [P,in_dim] may be NULL.  Parameter gradients are WRITTEN (not
 * accumulated); their row sums use a fixed order: deterministic, no floating-point atomics. */
int dm4d_deform_mlp_backward(int32_t P, const float *feat, const dm4d_mlp_weights *w, const float *h_save,
                             const float *y_save, const float *const *g_out /* [host] */, float *g_feat,
                             const dm4d_mlp_weights_grad *gw, void *scratch, dm4d_stream_t stream);

/* ------------------------------------------------------------------ mesh regularisers */

/* As-rigid-as-possible energy with GIVEN vertex rotations, for T timestamps in one launch:
 *   E_t = sum_i sum_{j in N(i)} w_ij || (x'_i - x'_j) - R_i (x_i - x_j) ||^2
 * = ARAPCoach.compute_arap_energy(xyz_prime, vert_rotations) of C/utils/arap_utils.py:183-224 as called per
 * timestamp by C/system/sugar_4dgen.py:374-385.  Static adjacency (device): CSR csr_offsets [V+1], neighbors [E],
 * reverse_edge [E] (index of the edge j -> i), weights [E], rest_edges [E,3] = x_i - x_j.  xyz_prime [T,V,3],
 * rotations [T,V,3,3] row-major.  forward writes vertex_energy [T,V] (E_t = its row sum); backward writes
 * g_xyz [T,V,3] and g_rotations [T,V,3,3] (either may be NULL) for the upstream g_energy [T].  Atomic-free. */
int dm4d_arap_energy_forward(int32_t T, int32_t V, const int32_t *csr_offsets, const int32_t *neighbors,
                             const int32_t *reverse_edge, const float *weights, const float *rest_edges,
                             const float *xyz_prime, const float *rotations, float *vertex_energy, dm4d_stream_t stream);
int dm4d_arap_energy_backward(int32_t T, int32_t V, const int32_t *csr_offsets, const int32_t *neighbors,
                              const int32_t *reverse_edge, const float *weights, const float *rest_edges,
                              const float *xyz_prime, const float *rotations, const float *g_energy, float *g_xyz,
                              float *g_rotations, dm4d_stream_t stream);

/* out = softmax(q k^T * scale) v per (batch, head): the self-attentions of the Zero123 UNet's transformer blocks
 * (extern/ldm_zero123/modules/attention.py:152-194) on the matrix cores.  float16 q, k, v with element (b, token, head, c) at
 * base + b * batch_stride + token * tok_stride + head * D + c (elements; the three may be views into one fused projection), out
 * [B][L][heads * D] float16; float32 statistics and accumulators, float16 probabilities.  D in {40, 64, 80, 160}, L a multiple of
 * 64, 16-byte aligned bases and strides. */
int dm4d_attention_f16(int32_t B, int32_t L, int32_t heads, int32_t D, const void *q, const void *k, const void *v, int64_t batch_stride,
                       int64_t tok_stride, void *out, float scale, dm4d_stream_t stream);

/* The image-space head of a dynamic-stage iteration (C/system/sugar_4dgen.py:148-190: comp_rgb = clamp(render, 0, 1); on the
 * reference views loss_rgb = mse(gt_rgb, comp_rgb), loss_mask = mse(gt_mask, opacity); the random views go to the Zero123
 * guidance, whose first step is a bilinear resize to 256 x 256, C/guidance/temporal_stable_zero123_guidance.py:299-310 -- at half
 * the size, the mean of each 2 x 2 block) in ONE launch each way instead of ~45 torch operators over 25 MB tensors.
 *   color [B][C >= 3][H][W], alpha [B][1][H][W] float32 (the batched renderer's outputs); ref_pos / rnd_pos [B] int32: the view's
 *   index among the reference / random views or -1; ref_images [L][H][W][3], ref_masks [L][H][W][1]; fidx_ref [n_ref] int64: the
 *   frame of each reference view.  H, W even.
 * forward: partial [B][dm4d_image_head_blocks(H, W)][2] = squared-error partial sums (rgb, mask) -- the caller adds them up (a
 *   fixed order: deterministic) and divides by n_ref H W 3 / n_ref H W; half_rgb [n_rnd][H/2][W/2][3] = the resized clamped views.
 * backward: g_rgb / g_mask = dL/d(mse_rgb), dL/d(mse_mask) (device scalars, NULL = 0), g_half = dL/d(half_rgb) or NULL; WRITES
 *   g_color [B][C][H][W] (torch.clamp's pass-through mask 0 <= x <= 1 applied; zeros in channels >= 3) and g_alpha [B][1][H][W]. */
int32_t dm4d_image_head_blocks(int32_t H, int32_t W);
int dm4d_image_head_forward(int32_t B, int32_t H, int32_t W, int32_t C, const float *color, const float *alpha, const int32_t *ref_pos,
                            const int32_t *rnd_pos, const float *ref_images, const float *ref_masks, const int64_t *fidx_ref, int32_t n_ref,
                            int32_t n_rnd, float *partial, float *half_rgb, dm4d_stream_t stream);
int dm4d_image_head_backward(int32_t B, int32_t H, int32_t W, int32_t C, const float *color, const float *alpha, const int32_t *ref_pos,
                             const int32_t *rnd_pos, const float *ref_images, const float *ref_masks, const int64_t *fidx_ref, int32_t n_ref,
                             int32_t n_rnd, const float *g_rgb, const float *g_mask, const float *g_half, float *g_color, float *g_alpha,
                             dm4d_stream_t stream);

/* The scalar arithmetic around the two heads (system/sugar_4dgen.py:296-330, sugar_static.py:246-340: `loss = lambda_rgb *
 * loss_rgb + lambda_mask * loss_mask + lambda_sds * loss_sds + ...` on 0-dim tensors, and the sum + normalisation of a head's
 * partial sums) as one launch each instead of one torch operator per multiply / add (5 us of launch for one flop, ~25 of them
 * per iteration forward + backward).
 *   dm4d_partial_sums: out[j] = sum_c matrix[c * m + j] * (sum_{i < n} partial[i * k + c]), 1 <= k, m <= 8; `matrix` is a HOST
 *     array (passed by value); one workgroup, fixed order.
 *   dm4d_weighted_sum: out[0] = ((0 + w[0] * *terms[0]) + w[1] * *terms[1]) + ... in float32, the torch expression's roundings;
 *     `terms` (an array of n <= 16 device pointers) and `weights` are HOST arrays.
 *   dm4d_weighted_sum_backward: out[i] = *g * w[i]. */
int dm4d_partial_sums(int64_t n, int32_t k, int32_t m, const float *partial, const float *matrix, float *out, dm4d_stream_t stream);
int dm4d_weighted_sum(int32_t n, const float *const *terms, const float *weights, float *out, dm4d_stream_t stream);
int dm4d_weighted_sum_backward(int32_t n, const float *g, const float *weights, float *out, dm4d_stream_t stream);

/* The image-space terms of a STATIC-stage iteration (system/sugar_static.py:110-340 with the static renderer's epilogue,
 * renderer/diff_sugar_rasterizer_normal.py:196-226) over the rendered batch color [B,6,H,W] (RGB | normal), depth, alpha [B,1,H,W]:
 * on the reference views (ref_pos >= 0) the sums of squares of mse(gt m, clamp(rgb) m) and mse(m, alpha); on the random views
 * (rnd_pos >= 0) the h / w total-variation sums of clamp(rgb), depth and the normal map normalize(normal) 0.5 alpha + 0.5, and the
 * half-size clamp(rgb) images [n_rnd,H/2,W/2,3] (the guidance's input).  partial: [B, dm4d_static_head_blocks(H, W), 8] sums per
 * workgroup {mse rgb, mse mask, tv rgb h, w, tv depth h, w, tv normal h, w} (the caller adds them and applies the normalisations of
 * F.mse_loss / threestudio's tv_loss).  backward: g_terms = upstream gradients of (mse_rgb, mse_mask, tv_rgb, tv_depth, tv_normal)
 * (5 floats on the device), g_half as half_rgb or NULL; writes dL/dcolor [B,6,H,W], dL/ddepth, dL/dalpha [B,1,H,W] (depth and
 * the normal map receive gradient only where alpha > 0.99, as the renderer detaches them elsewhere).  H, W even. */
int32_t dm4d_static_head_blocks(int32_t H, int32_t W);
int dm4d_static_head_forward(int32_t B, int32_t H, int32_t W, const float *color, const float *depth, const float *alpha, const int32_t *ref_pos,
                             const int32_t *rnd_pos, const float *ref_images, const float *ref_masks, const int64_t *fidx_ref, int32_t n_ref,
                             int32_t n_rnd, float *partial, float *half_rgb, dm4d_stream_t stream);
int dm4d_static_head_backward(int32_t B, int32_t H, int32_t W, const float *color, const float *depth, const float *alpha, const int32_t *ref_pos,
                              const int32_t *rnd_pos, const float *ref_images, const float *ref_masks, const int64_t *fidx_ref, int32_t n_ref,
                              int32_t n_rnd, const float *g_terms, const float *g_half, float *g_color, float *g_depth, float *g_alpha,
                              dm4d_stream_t stream);

/* The SuGaR geometry's per-Gaussian attributes from its parameters (geometry/sugar.py:471-570: get_xyz, get_scaling, get_rotation,
 * get_opacity, the degree-0 SH colour, the face normals; csrc/sugar_attr.hip has the expressions), one launch each way -- the static
 * stage learns every one of these parameters, and as torch operators the evaluation and its backward are ~300 launches on 50 k-element
 * tensors.  points [V,3], faces [F,3] int64, bary [G,3] (G in 1, 3, 4, 6), complex_numbers / log_scales [N,2], densities [N], sh_dc
 * [N,3], N = F G.  Outputs: the rasterizer's inputs means3D [N,3], rotations [N,4] (w,x,y,z), scales [N,3], opacities [N], colors6 [N,6]
 * (rgb | face normal).  backward: `scales` / `opacities` as the forward returned them; any dL_d* input may be NULL (= zeros), any
 * output NULL (= not wanted); dL_dpoints is zeroed and accumulated with float atomics (the order of a vertex's corner contributions is
 * the hardware's, as with the index_add of the torch composition). */
int dm4d_sugar_attributes_forward(int32_t F, int32_t G, int32_t V, const float *points, const int64_t *faces, const float *bary,
                                  const float *complex_numbers, const float *log_scales, const float *densities, const float *sh_dc,
                                  float thickness, float color_clip, float *means3D, float *rotations, float *scales, float *opacities,
                                  float *colors6, dm4d_stream_t stream);
int dm4d_sugar_attributes_backward(int32_t F, int32_t G, int32_t V, const float *points, const int64_t *faces, const float *bary,
                                   const float *complex_numbers, const float *log_scales, const float *densities, const float *sh_dc,
                                   float thickness, float color_clip, const float *scales, const float *opacities, const float *dL_dmeans3D,
                                   const float *dL_drotations, const float *dL_dscales, const float *dL_dopacities, const float *dL_dcolors6,
                                   float *dL_dpoints, float *dL_dcomplex, float *dL_dlog_scales, float *dL_ddensities, float *dL_dsh_dc,
                                   dm4d_stream_t stream);

/* R [n][3][3] (row-major) of n unit quaternions q [n][4] = (x, y, z, w): `get_timed_vertex_rotation(return_matrix=True)` of
 * C/geometry/dynamic_sugar.py:640-655 (a pypose SO3.matrix()), which the dynamic stage feeds to the ARAP term
 * (C/system/sugar_4dgen.py:304-311).  _backward_pypose: pypose's gradient with respect to the quaternion storage,
 * g_quat [n][4] = (sum_i (R e_i) x G[:, i], 0), from the forward's matrices and dL/dR.  (The Euclidean convention differentiates
 * the polynomial with torch autograd instead: ops.py::quat_xyzw_to_matrix.) */
int dm4d_quat_to_matrix_forward(int64_t n, const float *quat_xyzw, float *matrices, dm4d_stream_t stream);
int dm4d_quat_to_matrix_backward_pypose(int64_t n, const float *matrices, const float *g_matrices, float *g_quat, dm4d_stream_t stream);

/* pytorch3d.loss.mesh_normal_consistency (pytorch3d@stable, un-vendored: C/requirements.txt:46) of T deformed meshes
 * of one topology, as C/system/sugar_4dgen.py:214-226 applies it to the step's surface meshes (lambda 100,
 * C/configs/sugar_dynamic_dg.yaml:146).  `pairs` [P,4] (device, static): for every pair of faces sharing an edge,
 * (v0, v1, a, b) = the edge's vertices (v0 < v1) and the two opposite vertices, in pytorch3d's order (edges sorted,
 * incidences of an edge in face order, all i < j combinations).  terms [T,P] = 1 - cos((v1-v0)x(a-v0), -(v1-v0)x(b-v0));
 * the loss of mesh t is the mean of its terms and pytorch3d returns the mean over the meshes (host side).
 * Backward: vert_offsets [V+1] / vert_items (item = pair * 4 + role, role = 0..3 for v0, v1, a, b) list the pairs that
 * touch every vertex; g_loss [T] = dL/d(loss_t); g_xyz [T,V,3] is WRITTEN (gather, no atomics, deterministic); scratch: T P 12
 * floats, 16-byte aligned (the four vertices' gradient vectors of every pair, evaluated once per pair, then gathered). */
int dm4d_normal_consistency_forward(int32_t T, int32_t V, int32_t P, const int32_t *pairs, const float *xyz, float *terms,
                                    dm4d_stream_t stream);
int dm4d_normal_consistency_backward_scratch(int32_t T, int32_t V, int32_t P, const int32_t *pairs, const int32_t *vert_offsets,
                                             const int32_t *vert_items, const float *xyz, const float *g_loss, float *g_xyz,
                                             float *scratch, dm4d_stream_t stream);
/* The signature of rounds 1-3 (no scratch argument), kept for callers built against it: the same two kernels on a scratch the
 * LIBRARY owns (grown with hipMalloc when a larger T P is seen, one per process, calls serialised on it by stream order only --
 * callers with several streams use the _scratch form).  Round 4 had inserted `scratch` before `stream` in place. */
int dm4d_normal_consistency_backward(int32_t T, int32_t V, int32_t P, const int32_t *pairs, const int32_t *vert_offsets,
                                     const int32_t *vert_items, const float *xyz, const float *g_loss, float *g_xyz,
                                     dm4d_stream_t stream);

/* pytorch3d.loss.mesh_laplacian_smoothing(meshes, method="uniform") of T meshes of one topology (static stage lambda 1,
 * C/configs/sugar_static_refine.yaml:122, C/system/sugar_static.py:246-254; dynamic stage C/system/sugar_4dgen.py:227-230):
 * terms [T,V] = || mean of the one-ring neighbours - v_i ||; loss_t = mean_i, pytorch3d returns the mean over meshes.
 * csr_offsets [V+1] / neighbors [E]: symmetric one-ring.  unit [T,V,3] (forward output, backward input) = the unit
 * Laplacian vectors.  g_xyz [T,V,3] is WRITTEN (gather, deterministic); g_loss [T]. */
int dm4d_laplacian_smoothing_forward(int32_t T, int32_t V, const int32_t *csr_offsets, const int32_t *neighbors, const float *xyz,
                                     float *terms, float *unit, dm4d_stream_t stream);
int dm4d_laplacian_smoothing_backward(int32_t T, int32_t V, const int32_t *csr_offsets, const int32_t *neighbors, const float *unit,
                                      const float *g_loss, float *g_xyz, dm4d_stream_t stream);

/* ------------------------------------------------------------------ deformation-graph construction */

/* K nearest graph nodes of every mesh vertex in geodesic (shortest edge path) distance + skinning weights: the output
 * of DynamicSuGaRModel.build_deformation_graph(mode="geodisc") (C/geometry/dynamic_sugar.py:745-861), i.e.
 * _xyz_neighbor_node_idx [V,K] (int64) and the row-normalised _xyz_neighbor_nodes_weights [V,K] =
 * (1 - |v - node_k| / |v - node_{K+1}|)^2 (:845,859-861).  The reference runs one CPU heat-method solve per VERTEX
 * (potpourri3d, un-vendored); here the M nodes are the sources of one [M,V] relaxation over the mesh edges.
 * Static inputs (device): one-ring CSR csr_offsets [V+1] / neighbors [E] / edge_lengths [E]; verts [V,3];
 * node_xyz [M,3]; node_vertex [M] = the mesh vertex nearest to each node (:806-812).  scratch:
 * dm4d_graph_geodesic_scratch_bytes.  Synchronises the stream (convergence test). */
size_t dm4d_graph_geodesic_scratch_bytes(int32_t V, int32_t M);
int dm4d_graph_geodesic_knn(int32_t V, int32_t M, int32_t K, const int32_t *csr_offsets, const int32_t *neighbors,
                            const float *edge_lengths, const float *verts, const float *node_xyz, const int32_t *node_vertex,
                            void *scratch, int64_t *neighbor_idx, float *neighbor_weights, dm4d_stream_t stream);

/* ------------------------------------------------------------------ data-parallel gradient message */

/* The one exchange step of the path is an all-reduce of the parameter gradients.  The
 * message is one flat float32 buffer of up to DM4D_MAX_GRAD_SEGMENTS segments: segment k holds count[k] elements
 * at offset[k]; they are the whole gradient tensor grad[k] (index[k] == NULL) or its elements index[k][0..count)
 * (int64, device) -- for the HexPlane spatial planes only the texels the static graph nodes touch.  One launch
 * packs, one unpacks (message * scale -> gradients).  pack: grad[k] == NULL contributes zeros. */
#define DM4D_MAX_GRAD_SEGMENTS 64
typedef struct dm4d_grad_segments {
    int32_t n_segments;
    float *grad[DM4D_MAX_GRAD_SEGMENTS];
    const int64_t *index[DM4D_MAX_GRAD_SEGMENTS];
    int64_t count[DM4D_MAX_GRAD_SEGMENTS];
    int64_t offset[DM4D_MAX_GRAD_SEGMENTS];
} dm4d_grad_segments;
int dm4d_grad_pack(const dm4d_grad_segments *segments, float *flat, dm4d_stream_t stream);
int dm4d_grad_unpack(const dm4d_grad_segments *segments, const float *flat, float scale, dm4d_stream_t stream);

/* AdamW (torch/optim/adamw.py::_single_tensor_adamw, operation for operation; the optimiser of C/geometry/sugar.py:406-416) over the
 * elements of the message: segments->grad[k] = GRADIENT storage of segment k (NULL: zeros), param[k] = its PARAMETER storage (same
 * layout; index[] addresses both), group[k] its learning-rate group.  exp_avg / exp_avg_sq: the moments in message layout (zeros
 * before the first step); step: device scalar, steps APPLIED (advanced here unless *found_inf != 0, in which case nothing is
 * written); pending_decay (optional, [n_groups] float64 on the device): multiplied by this step's (1 - lr weight_decay) per group --
 * the whole update of the elements OUTSIDE the message (zero gradient, zero moments), which the caller applies when it needs them;
 * scratch: 4 floats.  Two launches.  grad_scale multiplies every gradient (1 / world after a sum over ranks). */
typedef struct dm4d_adamw_args {
    float beta1, beta2, eps, weight_decay;
    int32_t n_groups;
    float lr[8];
    int32_t group[DM4D_MAX_GRAD_SEGMENTS];
    float *param[DM4D_MAX_GRAD_SEGMENTS];
    float *exp_avg, *exp_avg_sq;
    double *step, *pending_decay;
    const float *found_inf;
    float *scratch;
} dm4d_adamw_args;
int dm4d_adamw_message(const dm4d_grad_segments *segments, const dm4d_adamw_args *a, float grad_scale, dm4d_stream_t stream);

/* Round 5: the same step with the hyperparameters PER GROUP and the optimiser's bookkeeping PER SEGMENT (dm4d_adamw_message stays
 * as it is for callers built against it).  Why per group: the reference's optimiser does not run one set of hyperparameters --
 * training_setup's Adam(l, lr=0, eps=1e-15) fills the group dicts of optimize_list in place (betas (0.9, 0.999), weight_decay 0) and
 * merge_optimizer's AdamW(l, betas=[0.9, 0.99], eps=1e-15) only fills what is missing, so the geometry / deformation groups run
 * beta2 = 0.999 without decay and only groups appended later get (0.9, 0.99) / 0.01 (C/geometry/sugar.py:382,406-416,
 * C/geometry/dynamic_sugar.py:231-235).
 *   skip[k] != 0: no gradient reached segment k -- torch.optim skips such a parameter entirely (nothing written, step[k] kept);
 *   step, pending_decay: [n_segments] float64 on the device (zeros / ones before the first step);
 *   grad_in_message[k] != 0: segments->grad[k] is already in MESSAGE layout (this rank's reduce-scattered slice: read at [i]) and
 *     index[k] addresses only the parameter; param_out[k] (optional): the updated values also in message layout (the all-gather's
 *     send slice) -- the sharded data-parallel step as pack, reduce-scatter, THIS, all-gather, unpack;
 *   scratch: 1 + 2 * DM4D_MAX_GRAD_SEGMENTS floats.  Two launches. */
typedef struct dm4d_adamw_step_args {
    int32_t n_groups;
    float lr[8], beta1[8], beta2[8], eps[8], weight_decay[8];
    int32_t group[DM4D_MAX_GRAD_SEGMENTS];
    float *param[DM4D_MAX_GRAD_SEGMENTS];
    float *param_out[DM4D_MAX_GRAD_SEGMENTS];
    uint8_t grad_in_message[DM4D_MAX_GRAD_SEGMENTS];
    uint8_t skip[DM4D_MAX_GRAD_SEGMENTS];
    float *exp_avg, *exp_avg_sq;
    double *step, *pending_decay;
    const float *found_inf;
    float *scratch;
} dm4d_adamw_step_args;
int dm4d_adamw_step(const dm4d_grad_segments *segments, const dm4d_adamw_step_args *a, float grad_scale, dm4d_stream_t stream);

/* ------------------------------------------------------------------ batched views (the fast path) */

/* The whole per-view hot path for B (frame, view) units of one scene in 8 launches forward /
 * 6 backward, every launch covering all views (grid.y = view), with no host synchronisation:
 * skinning -> face->Gaussians -> fused 6-channel rasterisation (RGB + normal pass of
 * C/renderer/diff_sugar_rasterizer_temporal.py:161-217 share one binning and one blend).
 * Replaces the per-view Python loop of C/renderer/gaussian_batch_renderer.py:21-76 and its
 * 2 rasterizer calls (2 host syncs) per view.  All tensors are caller-allocated device memory. */
typedef struct dm4d_views {
    int32_t B, N, F, G, V, M, K, method;      /* N = F*G Gaussians; method as dm4d_skin_vertices_forward */
    int32_t image_height, image_width;
    float tanfovx, tanfovy, scale_modifier;
    int64_t capacity;                          /* duplicates per view the binning workspace can hold */
    int64_t record_capacity;                   /* backward records per view the grad scratch can hold */
    const float *bg;                           /* [6]   */
    const float *viewmatrix, *projmatrix;      /* [B,16] each, row-vector convention */
    const float *verts;                        /* [V,3] static vertices */
    const int32_t *nbr_idx;                    /* [V,K] */
    const float *nbr_w;                        /* [V,K] */
    const float *dx, *dr, *ds, *d_opacity;     /* [B,M,3] [B,M,4] [B,M,6] [B,M] raw deformation-net outputs */
    const int32_t *faces;                      /* [F,3] */
    const float *q_static;                     /* [N,4] (w,x,y,z) */
    const float *scales, *opacities, *rgb;     /* [N,3] [N] [N,3] (shared by all views) */
    /* outputs */
    float *vxyz, *vrot;                        /* [B,V,3] [B,V,4] deformed vertices */
    float *means3D, *rotations, *colors;       /* [B,N,3] [B,N,4] [B,N,6] (colors = rgb | normal) */
    int32_t *radii;                            /* [B,N] */
    float *out_color, *out_depth, *out_alpha;  /* [B,6,H,W] [B,H,W] [B,H,W] */
    /* workspaces (dm4d_views_*_bytes) */
    void *geom, *binning, *image;
    /* Views that share a FRAME (timestamp) share its skinning and face->Gaussian transform (the reference caches
     * them per timestamp within a step, C/geometry/dynamic_sugar.py:375-386).  frame_index == NULL: every view is
     * its own frame (n_frames is ignored).  Otherwise frame_index [B] (device, values < n_frames) and every
     * per-"B" tensor above that does not depend on the camera -- dx, dr, ds, d_opacity, vxyz, vrot, means3D,
     * rotations, colors, and in dm4d_views_grads dL_dvxyz*, dL_dvrot*, dL_ddx..dL_ddo -- has n_frames leading
     * entries instead of B. */
    const int32_t *frame_index;
    int32_t n_frames;
    /* 0: `scales` is [N,3], shared by all views.  1: `scales` is per frame, [n_frames (B without frame_index), N, 3] -- the
     * reference's `d_scale: true` branch, where the deformation also stretches the Gaussians
     * (C/geometry/dynamic_sugar.py:682-704,717-720); dm4d_views_grads.dL_dscales stays per VIEW. */
    int32_t scales_per_frame;
    /* Backward records.  0 (DM4D_RECORDS_CELL): one per (Gaussian, 4x4-pixel cell), no atomics anywhere, gradients
     * bit-reproducible from run to run -- the parity tests' mode.  1 (DM4D_RECORDS_TILE): one per (Gaussian, tile), the
     * sixteen cells of a tile summed in LDS with ds_add_f32 by one workgroup: 3.4x fewer records to write and read back,
     * the last bits of the gradients depend on the order the waves were scheduled in (as with upstream's float
     * atomicAdd).  The forward image is bit-identical in both modes.  Forward and backward of a step must agree. */
    int32_t record_mode;
} dm4d_views;
#define DM4D_RECORDS_CELL 0
#define DM4D_RECORDS_TILE 1

typedef struct dm4d_views_grads {
    const float *dL_dcolor, *dL_ddepth, *dL_dalpha;     /* [B,6,H,W] [B,H,W] [B,H,W]; depth/alpha may be NULL */
    const float *dL_dvxyz_ext, *dL_dvrot_ext;           /* optional extra grads on the deformed vertices */
    const int32_t *node_csr_offsets, *node_csr_items;   /* as dm4d_skin_vertices_backward */
    const int32_t *vert_csr_offsets, *vert_csr_items;   /* as dm4d_face_gaussians_backward */
    void *grad_scratch, *skin_scratch, *face_scratch;   /* dm4d_views_{grad,skin_scratch,face_scratch}_bytes */
    /* outputs (per view; the caller reduces the static ones over B).  dL_dmeans3D, dL_drotations and dL_dcolors are optional
     * AS A GROUP: all NULL (then dL_dmeans2D may be NULL too, and dL_dopacity / dL_dscales must be) = nothing is materialised
     * per view -- the record gather and the face part of the face->Gaussian backward run as ONE kernel that keeps the three
     * gradients in registers (csrc/gather_face.hip; bit-identical vertex / node gradients, 83 MB per 8-view step less traffic
     * and one launch less on the bench scene). */
    float *dL_dmeans2D, *dL_dmeans3D, *dL_drotations;   /* [B,N,3] [B,N,3] [B,N,4] */
    float *dL_dcolors, *dL_dopacity, *dL_dscales;       /* [B,N,6] [B,N] [B,N,3]; opacity/scales may be NULL.
                                                         * dL_dopacity == NULL declares the static appearance
                                                         * (opacities, rgb) frozen, as in the reference's dynamic
                                                         * stage (C/geometry/dynamic_sugar.py:79-87): channels 0..2 of
                                                         * dL_dcolors are then written as zeros and the blend backward
                                                         * skips both reductions */
    float *dL_dvxyz, *dL_dvrot;                         /* [B,V,3] [B,V,4] */
    float *dL_ddx, *dL_ddr, *dL_dds, *dL_ddo;           /* [B,M,3] [B,M,4] [B,M,6] [B,M] */
} dm4d_views_grads;

size_t dm4d_views_geom_bytes(int32_t B, int32_t N, int32_t image_height, int32_t image_width);
size_t dm4d_views_binning_bytes(int32_t B, int64_t capacity);
size_t dm4d_views_image_bytes(int32_t B, int32_t image_height, int32_t image_width);
size_t dm4d_views_grad_bytes(int32_t B, int64_t record_capacity);
size_t dm4d_views_skin_scratch_bytes(int32_t B, int32_t V, int32_t K);
size_t dm4d_views_face_scratch_bytes(int32_t B, int32_t F);
int dm4d_views_forward(const dm4d_views *v, dm4d_stream_t stream);
int dm4d_views_backward(const dm4d_views *v, const dm4d_views_grads *g, dm4d_stream_t stream);
/* Round 5: the same backward when NO loss reads the normal image -- dL_dcolor's channels 3..5 are DECLARED zero and not read (the
 * tensor keeps its [B,6,H,W] shape), the blend backward carries 5 per-entry sums instead of 8 and no gradient flows through the
 * normals.  That is what autograd does in the reference when every normal weight is 0 (C/configs/sugar_dynamic_dg.yaml:145-157: the
 * normal pass's backward, C/renderer/diff_sugar_rasterizer_temporal.py:202-211, is never entered).  Requires the lean configuration:
 * dL_dopacity == NULL, dL_ddepth == NULL, cell records.  Results equal dm4d_views_backward fed zeros on channels 3..5. */
int dm4d_views_backward_rgb(const dm4d_views *v, const dm4d_views_grads *g, dm4d_stream_t stream);
/* num_rendered[b], num_records[b], overflowed[b] (bit 0: duplicates > capacity, bit 1: records >
 * record_capacity) of the last forward (host arrays of B, any may be NULL; synchronises the stream). */
int dm4d_views_counters(const dm4d_views *v, int64_t *num_rendered, int64_t *num_records, int32_t *overflowed,
                        dm4d_stream_t stream);

/* ------------------------------------------------------------------ B views of ONE set of Gaussians (the static stage's batch)
 * `GaussianBatchRenderer.batch_forward` (renderer/gaussian_batch_renderer.py:21-76) around `DiffGaussian.forward` of the static
 * renderer (renderer/diff_sugar_rasterizer_normal.py:88-226): every view of a batch renders the SAME Gaussians (the SuGaR geometry
 * evaluated once) from its own camera -- the reference loops over the views with one rasterizer call per pass and a host
 * synchronisation in each.  Here: one call forward, one backward, no synchronisation (capacities as for dm4d_views: counters in the
 * first 16 bytes of every view's geom workspace, dm4d_views_*_bytes for the workspace sizes).  colors = [N,6] (RGB | normal), as
 * the fused 6-channel pass of the per-view operator.  Gradients come back per VIEW (the caller sums them over B). */
typedef struct dm4d_gviews {
    int32_t B, N, image_height, image_width;
    float tanfovx, tanfovy, scale_modifier;
    int32_t record_mode;                       /* DM4D_RECORDS_CELL / DM4D_RECORDS_TILE */
    int64_t capacity, record_capacity;         /* duplicates / backward records per view */
    const float *bg;                           /* [6] */
    const float *viewmatrix, *projmatrix;      /* [B,16] each, row-vector convention */
    const float *means3D, *rotations;          /* [N,3] [N,4] (w,x,y,z) */
    const float *scales, *opacities, *colors;  /* [N,3] [N] [N,6] */
    int32_t *radii;                            /* [B,N] */
    float *out_color, *out_depth, *out_alpha;  /* [B,6,H,W] [B,H,W] [B,H,W] */
    void *geom, *binning, *image;              /* dm4d_views_{geom,binning,image}_bytes(B, ...) */
} dm4d_gviews;

typedef struct dm4d_gviews_grads {
    const float *dL_dcolor, *dL_ddepth, *dL_dalpha;     /* [B,6,H,W] [B,H,W] [B,H,W]; depth / alpha may be NULL */
    void *grad_scratch;                                 /* dm4d_views_grad_bytes(B, record_capacity) */
    float *dL_dmeans2D, *dL_dmeans3D, *dL_drotations;   /* [B,N,3] [B,N,3] [B,N,4] */
    float *dL_dscales, *dL_dopacity, *dL_dcolors;       /* [B,N,3] [B,N] [B,N,6] */
} dm4d_gviews_grads;

int dm4d_gviews_forward(const dm4d_gviews *v, dm4d_stream_t stream);
int dm4d_gviews_backward(const dm4d_gviews *v, const dm4d_gviews_grads *g, dm4d_stream_t stream);

/* ------------------------------------------------------------------ 3x3 convolution on the matrix cores (Zero123 UNet)
 * y = conv3x3(x, w) + bias (+ residual): stride 1, padding 1, float16, float32 accumulation (v_mfma_f32_32x32x16_f16).
 * x [N,H,W,C_in], y / residual [N,H,W,C_out] (NHWC = torch.channels_last storage), w [C_out,3,3,C_in] (a channels_last
 * torch weight as it lies in memory), bias [C_out] or NULL; C_in % 32 == 0, C_out % 32 == 0; 16-byte aligned device pointers.
 * `scratch` (dm4d_conv3x3_scratch_bytes, may be NULL when that returns <= 256): float32 partial sums of the split-K
 * launches of small problems.  The operator is the forward convolution (the guidance UNet runs without gradients,
 * extern/ldm_zero123/modules/diffusionmodules/openaimodel.py:214-275); the data gradient of a stride-1 convolution is the
 * same operator on the flipped, transposed filter (the VAE encoder's backward, dreammesh4d_amd/conv_mfma.py). */
size_t dm4d_conv3x3_scratch_bytes(int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout);
int dm4d_conv3x3_nhwc_f16(int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout, const void *x, const void *w, const void *bias,
                          const void *residual, void *y, void *scratch, dm4d_stream_t stream);
/* The same operator with a stride and with the zero padding of the LOW side of each axis given (the high side gets what the last
 * window needs).  stride 2, pad 1: the UNet's Downsample (openaimodel.py:123-156), output ceil(H_in / 2) x ceil(W_in / 2);
 * stride 2, pad 0: the VAE encoder's Downsample, which pads (0, 1, 0, 1) and convolves without padding
 * (diffusionmodules/model.py:85-100), output floor(H_in / 2) x floor(W_in / 2) -- the padded copy is never made.
 * stride 1 needs pad 1.  Implicit-GEMM kernel only.  The scratch size is that of the pad-1 output (never smaller). */
size_t dm4d_conv3x3_strided_scratch_bytes(int32_t N, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t stride);
int dm4d_conv3x3_strided_nhwc_f16(int32_t N, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t stride, int32_t pad, const void *x,
                                  const void *w, const void *bias, const void *residual, void *y, void *scratch, dm4d_stream_t stream);
/* Data gradient of the stride-2 / pad-0 convolution whose input carries one zero row / column behind each axis (the VAE encoder's
 * Downsample, extern/ldm_zero123/modules/diffusionmodules/model.py:85-100): dx [N,H_in,W_in,C_in] from dy [N,H_in/2,W_in/2,C_out],
 * NHWC float16, float32 accumulation; H_in, W_in even, C_in and C_out multiples of 32.  The four parity classes of the input
 * pixels are four stride-1 convolutions of dy (2x2, 2x1, 1x2, 1x1 taps) on the implicit-GEMM kernel, each writing every second
 * pixel of every second row: w_cls[2 (iy & 1) + (ix & 1)] = [C_in][KH][KW][C_out] with tap (ty, tx) = the forward filter's tap
 * (2 - 2 ty if KH == 2 else 1, 2 - 2 tx if KW == 2 else 1), transposed (conv_mfma.pack_weight_s2_dgrad). */
int dm4d_conv3x3_s2_dgrad_nhwc_f16(int32_t N, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, const void *dy, const void *const *w_cls,
                                   void *dx, dm4d_stream_t stream);
/* y = conv3x3(x, w) from exactly 128 channels to C_out <= 4 (stride 1, padding 1, float16, float32 accumulation, no bias):
 * x [N,H,W,128], w [C_out,3,3,128], y [N,H,W,C_out].  The data gradient of the VAE encoder's first convolution (image <-
 * 128 feature channels, ldm Encoder.conv_in, extern/ldm_zero123/modules/diffusionmodules/model.py:368-371): memory bound,
 * on the vector ALUs (v_dot2_f32_f16). */
int dm4d_conv3x3_c128_small_nhwc_f16(int32_t N, int32_t H, int32_t W, int32_t Cout, const void *x, const void *w, void *y,
                                     dm4d_stream_t stream);
/* y = act(x w^T + bias) (+ residual) for float16 x [M][K], w [N][K] (an nn.Linear weight, or the [C_out][1][1][C_in] filter of
 * a 1x1 convolution over NHWC pixels), bias [N] or NULL, residual [M][N] or NULL, float32 accumulation: the linear layers of
 * the Zero123 UNet's transformer blocks (extern/ldm_zero123/modules/attention.py:152-213: to_q / to_k / to_v / to_out, the
 * feed-forward, proj_in / proj_out) and the ResBlocks' 1x1 skip convolutions (openaimodel.py:247-256) on the implicit-GEMM MFMA
 * kernel of dm4d_conv3x3_nhwc_f16 with ONE tap.  The sum + bias is rounded to float16, the residual added after (rounded
 * again), as the separate add it replaces.
 *   act = 0: none, y [M][N].
 *   act = 1: GEGLU (attention.py:48-56: value, gate = proj.chunk(2); value * gelu(gate), erf form), y [M][N / 2].  The rows of
 *     w (and bias) must be INTERLEAVED in blocks of 64: rows 128 j .. 128 j + 63 = value rows 64 j .., rows 128 j + 64 .. 128 j + 127 =
 *     gate rows 64 j .. (conv_mfma.pack_geglu); N % 128 == 0, no residual.
 * K % 32 == 0, N % 8 == 0, 16-byte aligned pointers.  scratch: dm4d_linear_scratch_bytes (split-K partial sums; may be NULL
 * when that returns <= 256).  Errors: DM4D_ERR_UNSUPPORTED for shapes outside this, DM4D_ERR_INVALID for null / misaligned. */
size_t dm4d_linear_scratch_bytes(int64_t M, int32_t K, int32_t N);
int dm4d_linear_f16(int64_t M, int32_t K, int32_t N, const void *x, const void *w, const void *bias, const void *residual, void *y,
                    int32_t act, void *scratch, dm4d_stream_t stream);

/* ------------------------------------------------------------------ heat-method geodesics of the deformation graph
 * The device pieces of build_deformation_graph(mode="geodisc") with the reference's own distance, the heat method
 * (C/geometry/dynamic_sugar.py:794-861; csrc/heat.hip explains the M-instead-of-V Poisson solves):
 *
 * dm4d_cg_batched_f64: Jacobi-preconditioned conjugate gradients A X = B for S right-hand sides at once, A sparse symmetric
 * positive (semi-)definite in CSR (float64, int32 indices, diagonal included), X / B stored unknown-major [V][S] (device);
 * X holds the initial guess on entry.  diag_inv [V] = 1 / diagonal.  Stops when every column's |r| / |b| <= tol or after
 * max_iter iterations; the residual is looked at (one host sync) every check_every iterations.  All reductions are
 * two-stage and in fixed order: bit-reproducible.  Returns the number of iterations run (>= 0) or a negative error;
 * *final_rel_residual (host, optional) = the worst column's |r| / |b|. */
size_t dm4d_cg_batched_scratch_bytes(int32_t V, int32_t S);
int dm4d_cg_batched_f64(int32_t V, int32_t S, const int32_t *csr_offsets, const int32_t *csr_cols, const double *csr_vals,
                        const double *diag_inv, const double *B, double *X, void *scratch, int32_t max_iter, double tol,
                        int32_t check_every, double *final_rel_residual, dm4d_stream_t stream);
/* XT [3F][S] = -grad u / |grad u| per face and source from the heat solutions U [V][S]; G [F][3][3] (float64): grad u on
 * face f = sum_k u[faces[f][k]] G[f][k]. */
int dm4d_heat_face_directions(int32_t F, int32_t S, const int32_t *faces, const double *G, const double *U, double *XT, dm4d_stream_t stream);
/* For the S source vertices first_vertex .. first_vertex + S - 1: the K nearest of M nodes by score[m * ld + s] (smaller =
 * nearer; ties to the lower node index) and the reference's weights (1 - e_k / e_{K+1})^2 of the EUCLIDEAN distances to the node
 * positions, rows normalised (:842-861); neighbor_idx [V,K] int64 / neighbor_weights [V,K] are indexed by vertex. */
int dm4d_graph_select_knn(int32_t S, int32_t M, int32_t K, const double *score, int32_t ld, int32_t first_vertex, const float *verts,
                          const float *node_xyz, int64_t *neighbor_idx, float *neighbor_weights, dm4d_stream_t stream);

/* ------------------------------------------------------------------ deformation network of the nodes, fused
 * dm4d_hexplane_forward + dm4d_deform_mlp_forward in ONE launch (the 16 x in_dim feature tile of a workgroup goes straight
 * into the MLP's first layer), their backward in three launches instead of five (independent jobs side by side).
 * Arguments as the two operators' (`flags` = the DM4D_HEX_* word; in_dim of `w` must be S * 32; `feat` [B*M, S*32] and
 * `samples` (dm4d_hexplane_scratch_bytes) are outputs of the forward the backward reads; `scratch`:
 * dm4d_nodenet_scratch_bytes, shared by forward and backward; g_feat [B*M, S*32] is a work buffer of the backward).
 * Results are bit-identical to the two-operator path.  C/geometry/deformation.py:88-305,430-436. */
size_t dm4d_nodenet_scratch_bytes(int32_t S, int32_t M, int32_t B, int32_t n_heads);
int dm4d_nodenet_forward(int32_t S, int32_t M, int32_t B, const int32_t *res, const float *const *planes, int32_t flags,
                         const float *aabb_host, const float *nodes, const float *times, const dm4d_mlp_weights *w, float *feat,
                         void *samples, float *h_save, float *y_save, float *const *out, void *scratch, dm4d_stream_t stream);
int dm4d_nodenet_backward(int32_t S, int32_t M, int32_t B, const int32_t *res, const float *const *planes, int32_t flags,
                          const float *aabb_host, const float *nodes, const float *times, const dm4d_mlp_weights *w,
                          const float *feat, void *samples, const float *h_save, const float *y_save, const float *const *g_out,
                          int32_t n_spatial, const int32_t *sp_scale, const int32_t *sp_plane, const int32_t *sp_texel,
                          const int32_t *sp_off, const int32_t *sp_item, int32_t n_time, const int32_t *tp_scale,
                          const int32_t *tp_plane, const int32_t *tp_col, const int32_t *tp_off, const int32_t *tp_item,
                          float *g_feat, float *const *g_planes, const dm4d_mlp_weights_grad *gw, void *scratch, dm4d_stream_t stream);

/* ------------------------------------------------------------------ step object
 * One C call each way for the whole per-(frame, view) path of a dynamic-stage step on a FIXED problem:
 *   dm4d_step_forward  = dm4d_nodenet_forward (deformation network of the graph nodes at the step's timestamps)
 *                        -> dm4d_views_forward (skinning, face->Gaussian, fused RGB + normal rasterisation of every view)
 *   dm4d_step_backward = dm4d_views_backward -> dm4d_nodenet_backward (parameter gradients WRITTEN into the descriptor's buffers)
 * The object owns the structs, the pointer tables and the host arrays of the two operators (dm4d_step_create deep-copies the
 * descriptor: nothing of it has to outlive the call); all device memory stays the caller's and must stay valid and unchanged in
 * size until dm4d_step_destroy.  Per step the caller passes only what changes: timestamps, cameras, the view -> frame map, the
 * upstream gradients.  Kernels, launch order and results are exactly those of the two operators called one after the other.
 * Why it exists: the host side of a step (two autograd Functions, ~60 allocations, struct marshalling) took as long as the
 * step's ~1 ms of GPU work (dreammesh4d_amd/step.py).  Host-side loop it replaces: C/renderer/gaussian_batch_renderer.py:21-76
 * around C/geometry/dynamic_sugar.py:367-431.
 *
 * dm4d_step_desc.views: every field as for dm4d_views_forward EXCEPT viewmatrix / projmatrix / frame_index (arguments of
 *   dm4d_step_forward; n_frames is read when frame_index is given); dx / dr / ds / d_opacity must be the buffers node_out[]
 *   points at (ds / d_opacity NULL when the skinning method ignores them).
 * dm4d_step_desc.grads: scratch + outputs as for dm4d_views_backward; dL_dcolor / dL_ddepth / dL_dalpha / dL_dv*_ext are arguments
 *   of dm4d_step_backward; dL_dopacity / dL_dscales must be NULL (static appearance frozen: the dynamic stage,
 *   C/geometry/dynamic_sugar.py:79-87); dL_ddx .. dL_ddo must be the buffers node_gout[] points at.
 * node_out[k] / node_gout[k], k < w.n_heads: output of MLP head k [n_frames, M, out_dim[k]] and its upstream gradient (NULL = zero:
 *   a head the skinning method does not read).
 * hex_flags: the DM4D_HEX_* word of the forward; hex_backward_flags: OR-ed in for the backward (DM4D_HEX_KEEP_SPATIAL when g_planes
 *   are persistent buffers whose untouched texels are already zero). */
typedef struct dm4d_step dm4d_step;
typedef struct dm4d_step_desc {
    dm4d_views views;
    dm4d_views_grads grads;
    int32_t S, hex_flags, hex_backward_flags;
    const int32_t *res;                 /* host [S,4] */
    const float *aabb_host;             /* host [2,3] */
    const float *const *planes;         /* host: S*6 device pointers */
    float *const *g_planes;             /* host: S*6 device pointers (gradient planes) */
    const float *nodes;                 /* [M,3] */
    const float *times;                 /* (set by dm4d_step_forward) */
    dm4d_mlp_weights w;
    dm4d_mlp_weights_grad gw;
    float *node_out[4];
    const float *node_gout[4];
    float *feat, *h_save, *y_save, *g_feat;      /* [n_frames*M, S*32] [.,64] [n_heads,.,64] [n_frames*M, S*32] */
    void *samples, *net_scratch;                 /* dm4d_hexplane_scratch_bytes, dm4d_nodenet_scratch_bytes */
    int32_t n_spatial, n_time;                   /* gather plan of the HexPlane backward, as dm4d_nodenet_backward */
    const int32_t *sp_scale, *sp_plane, *sp_texel, *sp_off, *sp_item;
    const int32_t *tp_scale, *tp_plane, *tp_col, *tp_off, *tp_item;
} dm4d_step_desc;
int dm4d_step_create(const dm4d_step_desc *desc, dm4d_step **out);
void dm4d_step_destroy(dm4d_step *s);
int dm4d_step_forward(dm4d_step *s, const float *times01, const float *viewmatrix, const float *projmatrix,
                      const int32_t *frame_index, dm4d_stream_t stream);
int dm4d_step_backward(dm4d_step *s, const float *dL_dcolor, const float *dL_ddepth, const float *dL_dalpha,
                       const float *dL_dvxyz_ext, const float *dL_dvrot_ext, dm4d_stream_t stream);
/* dm4d_views_backward_rgb inside the step object (no depth gradient by construction). */
int dm4d_step_backward_rgb(dm4d_step *s, const float *dL_dcolor, const float *dL_dalpha, const float *dL_dvxyz_ext,
                           const float *dL_dvrot_ext, dm4d_stream_t stream);
/* the object's dm4d_views as the last forward left it (for dm4d_views_counters); NULL if `s` is not a step object */
const dm4d_views *dm4d_step_views(const dm4d_step *s);

#ifdef __cplusplus
}
#endif
#endif /* DM4D_H */
