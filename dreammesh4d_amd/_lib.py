"""ctypes binding of libdm4d_hip.so (C ABI: include/dm4d.h).

The product path has NO fallback: if the HIP library is missing or fails to load,
importing an operator raises.  (The CPU restatements under oracle/ are test
infrastructure and are never imported from here.)
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libdm4d_hip.so")
_LIB = None

OK = 0
c_f = C.POINTER(C.c_float)
c_i32 = C.POINTER(C.c_int32)
c_u32 = C.POINTER(C.c_uint32)
c_u64 = C.POINTER(C.c_uint64)
c_u8 = C.POINTER(C.c_uint8)
vp = C.c_void_p


class Dm4dError(RuntimeError):
    pass


class RasterSettings(C.Structure):
    _fields_ = [
        ("image_height", C.c_int32), ("image_width", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("scale_modifier", C.c_float), ("sh_degree", C.c_int32), ("prefiltered", C.c_int32), ("debug", C.c_int32),
        ("bg", vp), ("viewmatrix", vp), ("projmatrix", vp), ("campos", vp),
    ]


class RasterInputs(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("sh_coeffs", C.c_int32), ("n_channels", C.c_int32), ("means3D", vp), ("shs", vp), ("colors_precomp", vp),
        ("opacities", vp), ("scales", vp), ("rotations", vp), ("cov3D_precomp", vp),
    ]


class ViewsStruct(C.Structure):
    """dm4d_views (include/dm4d.h)."""
    _fields_ = [(n, C.c_int32) for n in ("B", "N", "F", "G", "V", "M", "K", "method", "image_height", "image_width")] + \
               [(n, C.c_float) for n in ("tanfovx", "tanfovy", "scale_modifier")] + [("capacity", C.c_int64), ("record_capacity", C.c_int64)] + \
               [(n, vp) for n in ("bg", "viewmatrix", "projmatrix", "verts", "nbr_idx", "nbr_w", "dx", "dr", "ds",
                                  "d_opacity", "faces", "q_static", "scales", "opacities", "rgb", "vxyz", "vrot",
                                  "means3D", "rotations", "colors", "radii", "out_color", "out_depth", "out_alpha",
                                  "geom", "binning", "image", "frame_index")] + [("n_frames", C.c_int32), ("scales_per_frame", C.c_int32), ("record_mode", C.c_int32)]


class ViewsGrads(C.Structure):
    """dm4d_views_grads (include/dm4d.h)."""
    _fields_ = [(n, vp) for n in ("dL_dcolor", "dL_ddepth", "dL_dalpha", "dL_dvxyz_ext", "dL_dvrot_ext",
                                  "node_csr_offsets", "node_csr_items", "vert_csr_offsets", "vert_csr_items",
                                  "grad_scratch", "skin_scratch", "face_scratch", "dL_dmeans2D", "dL_dmeans3D",
                                  "dL_drotations", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_dvxyz", "dL_dvrot",
                                  "dL_ddx", "dL_ddr", "dL_dds", "dL_ddo")]


class GViewsStruct(C.Structure):
    """dm4d_gviews (include/dm4d.h)."""
    _fields_ = [(n, C.c_int32) for n in ("B", "N", "image_height", "image_width")] + \
               [(n, C.c_float) for n in ("tanfovx", "tanfovy", "scale_modifier")] + [("record_mode", C.c_int32), ("capacity", C.c_int64), ("record_capacity", C.c_int64)] + \
               [(n, vp) for n in ("bg", "viewmatrix", "projmatrix", "means3D", "rotations", "scales", "opacities", "colors", "radii",
                                  "out_color", "out_depth", "out_alpha", "geom", "binning", "image")]


class GViewsGrads(C.Structure):
    """dm4d_gviews_grads (include/dm4d.h)."""
    _fields_ = [(n, vp) for n in ("dL_dcolor", "dL_ddepth", "dL_dalpha", "grad_scratch", "dL_dmeans2D", "dL_dmeans3D", "dL_drotations",
                                  "dL_dscales", "dL_dopacity", "dL_dcolors")]


class MlpWeights(C.Structure):
    """dm4d_mlp_weights (include/dm4d.h)."""
    _fields_ = [("in_dim", C.c_int32), ("width", C.c_int32), ("n_heads", C.c_int32), ("out_dim", C.c_int32 * 4),
                ("W0", vp), ("b0", vp), ("W1", vp * 4), ("b1", vp * 4), ("W2", vp * 4), ("b2", vp * 4)]


class MlpWeightsGrad(C.Structure):
    """dm4d_mlp_weights_grad (include/dm4d.h)."""
    _fields_ = [("W0", vp), ("b0", vp), ("W1", vp * 4), ("b1", vp * 4), ("W2", vp * 4), ("b2", vp * 4)]


class StepDesc(C.Structure):
    """dm4d_step_desc (include/dm4d.h)."""
    _fields_ = [("views", ViewsStruct), ("grads", ViewsGrads), ("S", C.c_int32), ("hex_flags", C.c_int32), ("hex_backward_flags", C.c_int32),
                ("res", vp), ("aabb_host", vp), ("planes", vp), ("g_planes", vp), ("nodes", vp), ("times", vp),
                ("w", MlpWeights), ("gw", MlpWeightsGrad), ("node_out", vp * 4), ("node_gout", vp * 4),
                ("feat", vp), ("h_save", vp), ("y_save", vp), ("g_feat", vp), ("samples", vp), ("net_scratch", vp),
                ("n_spatial", C.c_int32), ("n_time", C.c_int32)] + \
               [(n, vp) for n in ("sp_scale", "sp_plane", "sp_texel", "sp_off", "sp_item", "tp_scale", "tp_plane", "tp_col", "tp_off", "tp_item")]


MAX_GRAD_SEGMENTS = 64


class GradSegments(C.Structure):
    """dm4d_grad_segments (include/dm4d.h)."""
    _fields_ = [("n_segments", C.c_int32), ("grad", vp * MAX_GRAD_SEGMENTS), ("index", vp * MAX_GRAD_SEGMENTS),
                ("count", C.c_int64 * MAX_GRAD_SEGMENTS), ("offset", C.c_int64 * MAX_GRAD_SEGMENTS)]


class AdamwArgs(C.Structure):
    """dm4d_adamw_args (include/dm4d.h)."""
    _fields_ = [("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float), ("n_groups", C.c_int32),
                ("lr", C.c_float * 8), ("group", C.c_int32 * MAX_GRAD_SEGMENTS), ("param", vp * MAX_GRAD_SEGMENTS), ("exp_avg", vp),
                ("exp_avg_sq", vp), ("step", vp), ("pending_decay", vp), ("found_inf", vp), ("scratch", vp)]


class AdamwStepArgs(C.Structure):
    """dm4d_adamw_step_args (include/dm4d.h)."""
    _fields_ = [("n_groups", C.c_int32), ("lr", C.c_float * 8), ("beta1", C.c_float * 8), ("beta2", C.c_float * 8), ("eps", C.c_float * 8),
                ("weight_decay", C.c_float * 8), ("group", C.c_int32 * MAX_GRAD_SEGMENTS), ("param", vp * MAX_GRAD_SEGMENTS),
                ("param_out", vp * MAX_GRAD_SEGMENTS), ("grad_in_message", C.c_uint8 * MAX_GRAD_SEGMENTS), ("skip", C.c_uint8 * MAX_GRAD_SEGMENTS),
                ("exp_avg", vp), ("exp_avg_sq", vp), ("step", vp), ("pending_decay", vp), ("found_inf", vp), ("scratch", vp)]


ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int, C.c_size_t)

_SIGNATURES = {
    "dm4d_version": (C.c_int, []),
    "dm4d_last_error": (C.c_char_p, []),
    "dm4d_device_count": (C.c_int, []),
    "dm4d_profile_enable": (None, [C.c_uint]),
    "dm4d_profile_collect": (C.c_int64, [C.c_int, C.POINTER(C.c_double)]),
    "dm4d_device_arch": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "dm4d_raster_geom_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "dm4d_raster_binning_bytes": (C.c_size_t, [C.c_int64]),
    "dm4d_raster_image_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "dm4d_raster_grad_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "dm4d_rasterize_prepare": (C.c_int, [C.POINTER(RasterSettings), C.POINTER(RasterInputs), vp, vp, C.c_size_t, vp]),
    "dm4d_rasterize_num_rendered": (C.c_int64, [vp, vp]),
    "dm4d_rasterize_num_records": (C.c_int64, [vp, vp]),
    "dm4d_rasterize_counts": (C.c_int, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), vp]),
    "dm4d_rasterize_render": (C.c_int, [C.POINTER(RasterSettings), C.POINTER(RasterInputs), vp, vp, vp, C.c_int64,
                                        vp, vp, vp, vp, vp]),
    "dm4d_rasterize_overflowed": (C.c_int, [vp, vp]),
    "dm4d_rasterize_backward": (C.c_int, [C.POINTER(RasterSettings), C.POINTER(RasterInputs), vp, vp, vp, C.c_int64,
                                          vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dm4d_rasterize_forward": (C.c_int64, [C.POINTER(RasterSettings), C.POINTER(RasterInputs), vp, vp, vp, vp,
                                           ALLOC_FN, vp, vp]),
    "dm4d_raster_read_sorted": (C.c_int, [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int64, c_u64, c_u32, c_u32, vp]),
    "dm4d_raster_read_geom": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, c_f, c_f, c_f, c_u32, vp]),
    "dm4d_raster_read_image_state": (C.c_int, [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int64, c_u32, c_f, vp]),
    "dm4d_debug_trace": (C.c_int, [vp, C.c_uint32]),
    "dm4d_debug_sort_trace": (C.c_int, [vp]),
    "dm4d_mark_visible": (C.c_int, [C.c_int32, vp, vp, vp, vp]),
    "dm4d_groupnorm_nhwc_forward": (C.c_int, [C.c_int32] * 5 + [vp, vp, C.c_int32, vp, vp, C.c_float, C.c_int32, vp, vp, vp, C.c_int32, vp]),
    "dm4d_groupnorm_nhwc_backward": (C.c_int, [C.c_int32] * 5 + [vp, vp, C.c_int32, vp, vp, vp, C.c_int32, vp, vp, vp, C.c_int32, vp]),
    "dm4d_sds_prepare": (C.c_int, [C.c_int32] * 3 + [C.c_float] + [vp] * 17),
    "dm4d_sds_finish": (C.c_int, [C.c_int32] * 3 + [C.c_float, C.c_float] + [vp] * 18),
    "dm4d_groupnorm_nhwc_backward_add": (C.c_int, [C.c_int32] * 5 + [vp, vp, C.c_int32, vp, vp, vp, C.c_int32, vp, vp, vp, vp, C.c_int32, vp]),
    "dm4d_add_bias_nhwc": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, vp, vp, vp, vp, vp]),
    "dm4d_geglu": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, vp, vp, vp]),
    "dm4d_add_layernorm_f16": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, vp, vp, vp, vp, C.c_float, vp, vp, vp, vp]),
    "dm4d_dist2_knn3": (C.c_int, [C.c_int32, vp, vp, vp]),
    "dm4d_knn_scratch_bytes": (C.c_size_t, [C.c_int32]),
    "dm4d_dist2_knn3_ws": (C.c_int, [C.c_int32, vp, vp, vp, C.c_size_t, vp]),
    "dm4d_skin_vertices_forward": (C.c_int, [C.c_int32] * 4 + [vp] * 10),
    "dm4d_skin_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "dm4d_skin_vertices_backward": (C.c_int, [C.c_int32] * 4 + [vp] * 17),
    "dm4d_vertex_scales_forward": (C.c_int, [C.c_int32] * 5 + [vp] * 6),
    "dm4d_vertex_scales_backward": (C.c_int, [C.c_int32] * 5 + [vp] * 10),
    "dm4d_gaussian_scales_forward": (C.c_int, [C.c_int32] * 4 + [vp] * 6),
    "dm4d_gaussian_scales_backward": (C.c_int, [C.c_int32] * 4 + [vp] * 10),
    "dm4d_face_gaussians_forward": (C.c_int, [C.c_int32] * 2 + [vp] * 8),
    "dm4d_face_scratch_bytes": (C.c_size_t, [C.c_int32]),
    "dm4d_face_gaussians_backward": (C.c_int, [C.c_int32] * 3 + [vp] * 13),
    "dm4d_graph_geodesic_scratch_bytes": (C.c_size_t, [C.c_int32] * 2),
    "dm4d_graph_geodesic_knn": (C.c_int, [C.c_int32] * 3 + [vp] * 10),
    "dm4d_laplacian_smoothing_forward": (C.c_int, [C.c_int32] * 2 + [vp] * 6),
    "dm4d_laplacian_smoothing_backward": (C.c_int, [C.c_int32] * 2 + [vp] * 6),
    "dm4d_normal_consistency_forward": (C.c_int, [C.c_int32] * 3 + [vp] * 4),
    "dm4d_normal_consistency_backward": (C.c_int, [C.c_int32] * 3 + [vp] * 7),
    "dm4d_normal_consistency_backward_scratch": (C.c_int, [C.c_int32] * 3 + [vp] * 8),
    "dm4d_hexplane_forward": (C.c_int, [C.c_int32] * 3 + [vp] * 2 + [C.c_int32] + [vp] * 6),
    "dm4d_hexplane_axis_index": (C.c_int, [C.c_int32] * 2 + [vp] * 5),
    "dm4d_hexplane_scratch_bytes": (C.c_size_t, [C.c_int32] * 3),
    "dm4d_hexplane_backward": (C.c_int, [C.c_int32] * 3 + [vp] * 2 + [C.c_int32] + [vp] * 4 + [C.c_int32] + [vp] * 5 + [C.c_int32] + [vp] * 8),
    "dm4d_conv3x3_scratch_bytes": (C.c_size_t, [C.c_int32] * 5),
    "dm4d_conv3x3_nhwc_f16": (C.c_int, [C.c_int32] * 5 + [vp] * 7),
    "dm4d_conv3x3_c128_small_nhwc_f16": (C.c_int, [C.c_int32] * 4 + [vp] * 4),
    "dm4d_conv3x3_strided_scratch_bytes": (C.c_size_t, [C.c_int32] * 6),
    "dm4d_conv3x3_strided_nhwc_f16": (C.c_int, [C.c_int32] * 7 + [vp] * 7),
    "dm4d_conv3x3_s2_dgrad_nhwc_f16": (C.c_int, [C.c_int32] * 5 + [vp, C.POINTER(C.c_void_p), vp, vp]),
    "dm4d_attention_f16": (C.c_int, [C.c_int32] * 4 + [vp] * 3 + [C.c_int64, C.c_int64, vp, C.c_float, vp]),
    "dm4d_image_head_blocks": (C.c_int32, [C.c_int32, C.c_int32]),
    "dm4d_image_head_forward": (C.c_int, [C.c_int32] * 4 + [vp] * 7 + [C.c_int32, C.c_int32, vp, vp, vp]),
    "dm4d_image_head_backward": (C.c_int, [C.c_int32] * 4 + [vp] * 7 + [C.c_int32, C.c_int32] + [vp] * 6),
    "dm4d_partial_sums": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, vp, C.POINTER(C.c_float), vp, vp]),
    "dm4d_weighted_sum": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_float), vp, vp]),
    "dm4d_weighted_sum_backward": (C.c_int, [C.c_int32, vp, C.POINTER(C.c_float), vp, vp]),
    "dm4d_static_head_blocks": (C.c_int32, [C.c_int32, C.c_int32]),
    "dm4d_static_head_forward": (C.c_int, [C.c_int32] * 3 + [vp] * 8 + [C.c_int32, C.c_int32, vp, vp, vp]),
    "dm4d_static_head_backward": (C.c_int, [C.c_int32] * 3 + [vp] * 8 + [C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp]),
    "dm4d_sugar_attributes_forward": (C.c_int, [C.c_int32] * 3 + [vp] * 7 + [C.c_float, C.c_float] + [vp] * 6),
    "dm4d_sugar_attributes_backward": (C.c_int, [C.c_int32] * 3 + [vp] * 7 + [C.c_float, C.c_float] + [vp] * 13),
    "dm4d_quat_to_matrix_forward": (C.c_int, [C.c_int64, vp, vp, vp]),
    "dm4d_quat_to_matrix_backward_pypose": (C.c_int, [C.c_int64, vp, vp, vp, vp]),
    "dm4d_linear_scratch_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "dm4d_linear_f16": (C.c_int, [C.c_int64, C.c_int32, C.c_int32] + [vp] * 5 + [C.c_int32, vp, vp]),
    "dm4d_cg_batched_scratch_bytes": (C.c_size_t, [C.c_int32] * 2),
    "dm4d_cg_batched_f64": (C.c_int, [C.c_int32] * 2 + [vp] * 7 + [C.c_int32, C.c_double, C.c_int32, C.POINTER(C.c_double), vp]),
    "dm4d_heat_face_directions": (C.c_int, [C.c_int32] * 2 + [vp] * 5),
    "dm4d_graph_select_knn": (C.c_int, [C.c_int32] * 3 + [vp, C.c_int32, C.c_int32] + [vp] * 5),
    "dm4d_nodenet_scratch_bytes": (C.c_size_t, [C.c_int32] * 4),
    "dm4d_nodenet_forward": (C.c_int, [C.c_int32] * 3 + [vp] * 2 + [C.c_int32] + [vp] * 3 + [C.POINTER(MlpWeights)] + [vp] * 4 + [C.POINTER(vp)] + [vp] * 2),
    "dm4d_nodenet_backward": (C.c_int, [C.c_int32] * 3 + [vp] * 2 + [C.c_int32] + [vp] * 3 + [C.POINTER(MlpWeights)] + [vp] * 4 + [C.POINTER(vp)]
                              + [C.c_int32] + [vp] * 5 + [C.c_int32] + [vp] * 5 + [vp] * 2 + [C.POINTER(MlpWeightsGrad)] + [vp] * 2),
    "dm4d_deform_mlp_scratch_bytes": (C.c_size_t, [C.c_int32] * 3),
    "dm4d_deform_mlp_forward": (C.c_int, [C.c_int32, vp, C.POINTER(MlpWeights), vp, vp, C.POINTER(vp), vp, vp]),
    "dm4d_deform_mlp_backward": (C.c_int, [C.c_int32, vp, C.POINTER(MlpWeights), vp, vp, C.POINTER(vp), vp,
                                           C.POINTER(MlpWeightsGrad), vp, vp]),
    "dm4d_arap_energy_forward": (C.c_int, [C.c_int32, C.c_int32] + [vp] * 9),
    "dm4d_arap_energy_backward": (C.c_int, [C.c_int32, C.c_int32] + [vp] * 11),
    "dm4d_grad_pack": (C.c_int, [C.POINTER(GradSegments), vp, vp]),
    "dm4d_grad_unpack": (C.c_int, [C.POINTER(GradSegments), vp, C.c_float, vp]),
    "dm4d_adamw_message": (C.c_int, [C.POINTER(GradSegments), C.POINTER(AdamwArgs), C.c_float, vp]),
    "dm4d_adamw_step": (C.c_int, [C.POINTER(GradSegments), C.POINTER(AdamwStepArgs), C.c_float, vp]),
    "dm4d_views_geom_bytes": (C.c_size_t, [C.c_int32] * 4),
    "dm4d_views_binning_bytes": (C.c_size_t, [C.c_int32, C.c_int64]),
    "dm4d_views_image_bytes": (C.c_size_t, [C.c_int32] * 3),
    "dm4d_views_grad_bytes": (C.c_size_t, [C.c_int32, C.c_int64]),
    "dm4d_views_skin_scratch_bytes": (C.c_size_t, [C.c_int32] * 3),
    "dm4d_views_face_scratch_bytes": (C.c_size_t, [C.c_int32] * 2),
    "dm4d_views_forward": (C.c_int, [C.POINTER(ViewsStruct), vp]),
    "dm4d_gviews_forward": (C.c_int, [C.POINTER(GViewsStruct), vp]),
    "dm4d_gviews_backward": (C.c_int, [C.POINTER(GViewsStruct), C.POINTER(GViewsGrads), vp]),
    "dm4d_views_backward": (C.c_int, [C.POINTER(ViewsStruct), C.POINTER(ViewsGrads), vp]),
    "dm4d_views_backward_rgb": (C.c_int, [C.POINTER(ViewsStruct), C.POINTER(ViewsGrads), vp]),
    "dm4d_views_counters": (C.c_int, [C.POINTER(ViewsStruct), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int32), vp]),
    "dm4d_step_create": (C.c_int, [C.POINTER(StepDesc), C.POINTER(vp)]),
    "dm4d_step_destroy": (None, [vp]),
    "dm4d_step_forward": (C.c_int, [vp] * 6),
    "dm4d_step_backward": (C.c_int, [vp] * 7),
    "dm4d_step_backward_rgb": (C.c_int, [vp] * 6),
    "dm4d_step_views": (vp, [vp]),
}


def declared_symbols():
    """Every function include/dm4d.h declares (parsed from the header, not from this table)."""
    import re

    hdr = os.path.join(os.path.dirname(_HERE), "include", "dm4d.h")
    text = open(hdr).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dm4d_[a-z0-9_]+)\s*\(", text)) - {"dm4d_alloc_fn"})


def build(force: bool = False) -> str:
    """Compile libdm4d_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-s", "-C", csrc, "-j8"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return SO_PATH


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  dreammesh4d_amd has no CPU fallback.")
        L = C.CDLL(SO_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        want = abi_version()
        if L.dm4d_version() != want:
            raise ImportError(f"{SO_PATH} has ABI version {L.dm4d_version()}, include/dm4d.h declares {want}: rebuild it "
                              "(`python -c 'import __graft_entry__ as g; g.build()'`)")
        _LIB = L
    return _LIB


def abi_version() -> int:
    """DM4D_ABI_VERSION of include/dm4d.h (the header this table of signatures was written against)."""
    import re

    hdr = os.path.join(os.path.dirname(_HERE), "include", "dm4d.h")
    return int(re.search(r"#define\s+DM4D_ABI_VERSION\s+(\d+)", open(hdr).read()).group(1))


def check(rc, what=""):
    if rc is not None and rc < 0:
        msg = lib().dm4d_last_error()
        raise Dm4dError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
    return rc
