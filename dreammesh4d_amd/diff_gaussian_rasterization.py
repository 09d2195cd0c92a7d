"""Drop-in for the ``diff_gaussian_rasterization`` operator module (ashawkey fork:
4-tuple return ``(color, radii, depth, alpha)``) that DreamMesh4D imports at

* custom/threestudio-dreammesh4d/renderer/diff_sugar_rasterizer_temporal.py:8-11
* custom/threestudio-dreammesh4d/renderer/diff_sugar_rasterizer_normal.py:8-11

and calls at ...temporal.py:129-144 (settings), :169-178 (RGB pass), :202-211 (normal pass).
Same names, argument meaning and error behaviour; the compute is libdm4d_hip.so
(hand-written HIP for gfx950) through the C ABI of include/dm4d.h.  No CPU fallback.

``dreammesh4d_amd.install_compat()`` registers this module under the upstream name so the
reference's renderer files import it unmodified.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# diagnostics for bench.py: num_rendered (duplicate count D) of the most recent forward calls
LAST_NUM_RENDERED = []


def _f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _ptr(t):
    return None if t is None or t.numel() == 0 else t.data_ptr()


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


class _Call:
    """ctypes structs + the tensors that back their pointers (kept alive together)."""

    def __init__(self, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs):
        dev = means3D.device
        if dev.type != "cuda":
            raise RuntimeError("diff_gaussian_rasterization (dm4d): tensors must live on a HIP device; "
                               "there is no CPU path")
        self.dev = dev
        self.N = int(means3D.shape[0])
        self.H, self.W = int(rs.image_height), int(rs.image_width)
        keep = self.keep = {}
        keep["means3D"] = _f32(means3D, dev)
        keep["opac"] = _f32(opacities, dev)
        keep["sh"] = _f32(sh, dev) if sh.numel() else None
        keep["col"] = _f32(colors_precomp, dev) if colors_precomp.numel() else None
        keep["scales"] = _f32(scales, dev) if scales.numel() else None
        keep["rots"] = _f32(rotations, dev) if rotations.numel() else None
        keep["cov"] = _f32(cov3Ds_precomp, dev) if cov3Ds_precomp.numel() else None
        keep["bg"] = _f32(rs.bg, dev)
        keep["view"] = _f32(rs.viewmatrix, dev)
        keep["proj"] = _f32(rs.projmatrix, dev)
        keep["campos"] = _f32(rs.campos, dev)
        self.M = int(keep["sh"].shape[1]) if keep["sh"] is not None and keep["sh"].dim() == 3 else 0
        # extension of the upstream operator: colors_precomp [N,6] and a 6-vector bg blend two colour sets (RGB | normal,
        # the two passes of the reference's renderers, …_normal.py:161-195) over ONE binning and blend
        self.C = 6 if keep["col"] is not None and keep["col"].dim() == 2 and keep["col"].shape[1] == 6 else 3
        if self.C == 6 and keep["bg"].numel() != 6:
            raise ValueError("6-channel colors_precomp needs a 6-element bg")
        self.settings = _lib.RasterSettings(self.H, self.W, float(rs.tanfovx), float(rs.tanfovy),
                                            float(rs.scale_modifier), int(rs.sh_degree), int(bool(rs.prefiltered)),
                                            int(bool(rs.debug)), _ptr(keep["bg"]), _ptr(keep["view"]),
                                            _ptr(keep["proj"]), _ptr(keep["campos"]))
        self.inputs = _lib.RasterInputs(self.N, self.M, self.C, _ptr(keep["means3D"]), _ptr(keep["sh"]), _ptr(keep["col"]),
                                        _ptr(keep["opac"]), _ptr(keep["scales"]), _ptr(keep["rots"]),
                                        _ptr(keep["cov"]))


def _bytes(n, dev):
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=dev)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        L = _lib.lib()
        call = _Call(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings)
        dev, N, H, W = call.dev, call.N, call.H, call.W
        with torch.cuda.device(dev):
            st = _stream(dev)
            color = torch.empty(call.C, H, W, dtype=torch.float32, device=dev)
            depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
            alpha = torch.empty(1, H, W, dtype=torch.float32, device=dev)
            radii = torch.empty(N, dtype=torch.int32, device=dev)
            geom_bytes = L.dm4d_raster_geom_bytes(N, H, W)
            geom = _bytes(geom_bytes, dev)
            _lib.check(L.dm4d_rasterize_prepare(call.settings, call.inputs, _ptr(radii), geom.data_ptr(), geom_bytes,
                                                st), "dm4d_rasterize_prepare")
            # the one host sync the upstream operator also has (sizing the duplicate list); the same read
            # returns the number of backward records
            import ctypes as _C
            cD, cR = _C.c_int64(0), _C.c_int64(0)
            _lib.check(L.dm4d_rasterize_counts(geom.data_ptr(), _C.byref(cD), _C.byref(cR), st), "dm4d_rasterize_counts")
            D, R = int(cD.value), int(cR.value)
            binning = _bytes(L.dm4d_raster_binning_bytes(D), dev)
            image = _bytes(L.dm4d_raster_image_bytes(H, W), dev)
            _lib.check(L.dm4d_rasterize_render(call.settings, call.inputs, _ptr(radii), geom.data_ptr(),
                                               binning.data_ptr(), D, image.data_ptr(), color.data_ptr(),
                                               depth.data_ptr(), alpha.data_ptr(), st), "dm4d_rasterize_render")
        ctx.call = call
        ctx.num_rendered = int(D)
        ctx.num_records = int(R)
        LAST_NUM_RENDERED.append(int(D))
        del LAST_NUM_RENDERED[:-64]
        ctx.shapes = (means3D.shape, means2D.shape, sh.shape, colors_precomp.shape, opacities.shape, scales.shape,
                      rotations.shape, cov3Ds_precomp.shape)
        ctx.save_for_backward(radii, geom, binning, image)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        L = _lib.lib()
        call = ctx.call
        radii, geom, binning, image = ctx.saved_tensors
        dev, N, H, W, D = call.dev, call.N, call.H, call.W, ctx.num_rendered
        s_m3, s_m2, s_sh, s_col, s_op, s_sc, s_rot, s_cov = ctx.shapes
        f = dict(dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _stream(dev)
            g_color = _f32(grad_color, dev) if grad_color is not None else torch.zeros(call.C, H, W, **f)
            g_depth = _f32(grad_depth, dev) if grad_depth is not None else None
            g_alpha = _f32(grad_alpha, dev) if grad_alpha is not None else None
            d_m2 = torch.empty(N, 3, **f)
            d_m3 = torch.empty(N, 3, **f)
            d_op = torch.empty(N, **f)
            d_col = torch.empty(N, call.C, **f)
            d_sh = torch.empty(N, call.M, 3, **f) if call.M > 0 else None
            has_sr = call.keep["scales"] is not None
            d_sc = torch.empty(N, 3, **f) if has_sr else None
            d_rot = torch.empty(N, 4, **f) if has_sr else None
            d_cov = torch.empty(N, 6, **f) if not has_sr else None
            R = ctx.num_records
            grad = _bytes(L.dm4d_raster_grad_bytes(R, call.C), dev)
            _lib.check(L.dm4d_rasterize_backward(
                call.settings, call.inputs, _ptr(radii), geom.data_ptr(), binning.data_ptr(), D, image.data_ptr(),
                grad.data_ptr(), R, g_color.data_ptr(), _ptr(g_depth), _ptr(g_alpha), _ptr(d_m2), _ptr(d_m3),
                _ptr(d_op), _ptr(d_col), _ptr(d_sh), _ptr(d_sc), _ptr(d_rot), _ptr(d_cov), st),
                "dm4d_rasterize_backward")
        # ctx.call stays (it only holds the detached inputs, the workspaces are saved tensors the C backward does not
        # modify): a second backward through the operator -- retain_graph=True, two losses -- works as upstream's does

        def shaped(t, shape):
            return None if t is None or len(shape) == 0 or 0 in shape else t.reshape(shape)

        return (shaped(d_m3, s_m3), shaped(d_m2, s_m2), shaped(d_sh, s_sh),
                shaped(d_col, s_col) if call.M == 0 else None, shaped(d_op, s_op), shaped(d_sc, s_sc),
                shaped(d_rot, s_rot), shaped(d_cov, s_cov), None)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        L = _lib.lib()
        with torch.no_grad():
            rs = self.raster_settings
            dev = positions.device
            pos = _f32(positions, dev)
            view = _f32(rs.viewmatrix, dev)
            present = torch.empty(pos.shape[0], dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _lib.check(L.dm4d_mark_visible(pos.shape[0], _ptr(pos), _ptr(view), _ptr(present), _stream(dev)),
                           "dm4d_mark_visible")
        return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        empty = torch.Tensor([]).to(means3D.device)
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, raster_settings)
