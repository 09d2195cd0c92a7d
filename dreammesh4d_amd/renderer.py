"""Renderer glue of the hot path: the batched counterpart of

* ``GaussianBatchRenderer.batch_forward``  (custom/threestudio-dreammesh4d/renderer/gaussian_batch_renderer.py:9-122)
* ``DiffGaussian.forward``                 (.../renderer/diff_sugar_rasterizer_temporal.py:81-239), registered by the
  reference as ``diff-sugar-rasterizer-temporal``
* ``get_cam_info_gaussian``                (threestudio/utils/ops.py:359-413)
* ``Depth2Normal``                         (.../renderer/diff_sugar_rasterizer_temporal.py:25-54)

The reference loops over the views in Python; per view it builds the camera (two 4x4 inversions, three H2D
copies), deforms the mesh, and calls the CUDA rasterizer twice (RGB, then normals with the same geometry).  Here
one ``batch_forward`` = one ``views.render_views`` call (all views, both passes, skinning shared per timestamp)
plus the per-pixel epilogue below, written as batched tensor ops.  Output dict keys, shapes and detach rules
follow the reference so its systems (``C/system/sugar_4dgen.py``) read it unchanged.
"""
import math
import os
from typing import Dict, NamedTuple, Optional

import torch
import torch.nn.functional as F

from . import views


# ------------------------------------------------------------------------------------------------ camera
def convert_pose(c2w):
    """OpenGL camera-to-world -> the Gaussian-splatting convention: flip y and z (ops.py:359-364)."""
    flip = torch.tensor([1.0, -1.0, -1.0, 1.0], dtype=c2w.dtype, device=c2w.device)
    return c2w * flip          # right-multiplication by diag(1,-1,-1,1) scales the columns


def projection_matrix_gaussian(znear, zfar, fovx, fovy, device=None, dtype=torch.float32):
    """[...,4,4] perspective matrix of ops.py:367-388 (z_sign = +1), for scalar or batched fov (radians)."""
    fovx = torch.as_tensor(fovx, dtype=dtype, device=device)
    fovy = torch.as_tensor(fovy, dtype=dtype, device=device)
    tx, ty = torch.tan(fovx * 0.5), torch.tan(fovy * 0.5)
    P = torch.zeros(fovx.shape + (4, 4), dtype=dtype, device=fovx.device)
    # right = tx * znear, left = -right: 2 znear / (right - left) = 1 / tx; the (right + left) terms vanish
    P[..., 0, 0] = 1.0 / tx
    P[..., 1, 1] = 1.0 / ty
    P[..., 3, 2] = 1.0
    P[..., 2, 2] = zfar / (zfar - znear)
    P[..., 2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def cam_info_gaussian(c2w, fovx, fovy, znear=0.1, zfar=100.0):
    """Batched ``get_cam_info_gaussian`` (ops.py:398-413): c2w [...,4,4], fov in radians ->
    (world_view_transform, full_proj_transform, camera_center) in the rasterizer's row-vector ("transposed")
    convention."""
    c2w = convert_pose(c2w.to(torch.float32))
    wv = torch.linalg.inv(c2w).transpose(-1, -2).contiguous()
    proj = projection_matrix_gaussian(znear, zfar, fovx, fovy, device=c2w.device).transpose(-1, -2)
    full = wv @ proj
    center = torch.linalg.inv(wv)[..., 3, :3]
    return wv, full, center


def batch_cameras(batch, device):
    """(w2c, full_proj, camera_center [B,...] on `device`, fovy [B] as given) of a batch dict.  A batch whose `c2w` / `fovy` arrive
    on the HOST (the data loaders' side) has its 4 x 4 algebra done there and ONE pinned, non-blocking upload per result: on the
    device `torch.linalg.inv` checks its `info` on the host (two synchronisations) and a pageable host-to-device copy waits for
    everything queued on the stream -- six stalls per call that serialise the host with the device."""
    c2w = batch["c2w"]
    B = int(c2w.shape[0])
    fovy = torch.as_tensor(batch["fovy"], dtype=torch.float32).reshape(-1)
    if c2w.device.type == "cpu" and fovy.device.type == "cpu" and torch.device(device).type == "cuda":
        w2c, full, center = cam_info_gaussian(c2w, fovy.expand(B), fovy.expand(B), 0.1, 100.0)
        up = lambda t: torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t).to(device, non_blocking=True)
        return up(w2c), up(full), up(center), fovy
    fd = fovy.to(device).expand(B)
    w2c, full, center = cam_info_gaussian(c2w.to(device), fd, fd, 0.1, 100.0)
    return w2c, full, center, fovy


class Camera(NamedTuple):
    """``Camera`` of C/geometry/gaussian_base.py:175-184."""
    FoVx: torch.Tensor
    FoVy: torch.Tensor
    camera_center: torch.Tensor
    image_width: int
    image_height: int
    world_view_transform: torch.Tensor
    full_proj_transform: torch.Tensor
    timestamp: Optional[torch.Tensor] = None
    frame_idx: Optional[torch.Tensor] = None


def ray_directions(H, W, focal, device=None):
    """``get_ray_directions`` (ops.py:180-217) with pixel centres, principal point at the image centre."""
    i, j = torch.meshgrid(torch.arange(W, dtype=torch.float32, device=device) + 0.5,
                          torch.arange(H, dtype=torch.float32, device=device) + 0.5, indexing="xy")
    return torch.stack([(i - W / 2) / focal, -(j - H / 2) / focal, -torch.ones_like(i)], -1)


def rays(directions, c2w, normalize=True):
    """``get_rays(..., keepdim=True)`` (ops.py:274-320): directions [H,W,3], c2w [B,4,4] -> rays_o, rays_d [B,H,W,3]."""
    rays_d = (directions[None, :, :, None, :] * c2w[:, None, None, :3, :3]).sum(-1)
    rays_o = c2w[:, None, None, :3, 3].expand(rays_d.shape)
    if normalize:
        rays_d = F.normalize(rays_d, dim=-1)
    return rays_o, rays_d


# ------------------------------------------------------------------------------------------------ epilogue
def depth_to_normal(xyz):
    """``Depth2Normal`` (…temporal.py:25-54): xyz [B,3,H,W] -> -cross(d/dx, d/dy) with the 3x3 central-difference
    kernels and zero padding, as slices instead of two conv2d calls."""
    p = F.pad(xyz, (1, 1, 1, 1))
    ddx = p[:, :, 1:-1, 2:] - p[:, :, 1:-1, :-2]
    ddy = p[:, :, 2:, 1:-1] - p[:, :, :-2, 1:-1]
    return -torch.cross(ddx, ddy, dim=1)


def _where_detached(x, mask):
    """x with gradient only where `mask` (reference: ``x[~mask] = x[~mask].detach()``)."""
    return torch.where(mask, x, x.detach())


def compose_outputs(color, depth, alpha, rays_o=None, rays_d=None):
    """The reference renderer's image-space epilogue (…temporal.py:180-229) on the raw rasterizer images color [B,6,H,W] (rgb | blended
    normals), depth, alpha [B,1,H,W]: ``comp_rgb``, ``comp_normal``, ``comp_depth``, ``comp_mask`` and, with rays, ``comp_normal_from_dist``
    ([B,H,W,.]); gradients of depth / normals only where alpha > 0.99 (``x[~mask] = x[~mask].detach()``)."""
    B, _, H, W = color.shape
    mask = alpha > 0.99
    depth = _where_detached(depth, mask)                          # …temporal.py:180-181
    mask3 = mask.expand(B, 3, H, W)
    res = {}
    if rays_d is not None:
        xyz = rays_o + depth.permute(0, 2, 3, 1) * rays_d          # :187
        nd = F.normalize(depth_to_normal(xyz.permute(0, 3, 1, 2)), dim=1)
        res["comp_normal_from_dist"] = _where_detached(nd * 0.5 * alpha + 0.5, mask3).permute(0, 2, 3, 1)
    n = F.normalize(color[:, 3:], dim=1)                          # :212-217
    res["comp_normal"] = _where_detached(n * 0.5 * alpha + 0.5, mask3).permute(0, 2, 3, 1)
    res["comp_rgb"] = color[:, :3].clamp(0, 1).permute(0, 2, 3, 1)
    res["comp_depth"] = depth.permute(0, 2, 3, 1)
    res["comp_mask"] = alpha.permute(0, 2, 3, 1)
    return res


class DiffGaussianTemporal:
    """The reference's ``diff-sugar-rasterizer-temporal`` renderer over a ``sugar.DynamicSuGaR`` geometry.

    ``batch_forward(batch)`` takes the reference's batch dict (``c2w [B,4,4]``, ``fovy [B]`` radians, ``height``,
    ``width``, ``rays_o`` / ``rays_d [B,H,W,3]``, ``timestamp [B]`` and/or ``frame_indices [B]``) and returns the
    reference's output dict: ``comp_rgb``, ``comp_normal``, ``comp_normal_from_dist`` [B,H,W,3], ``comp_depth``,
    ``comp_mask`` [B,H,W,1], and the per-view lists ``viewspace_points`` (their ``.grad`` receives the screen-space
    mean gradients), ``visibility_filter``, ``radii``."""

    def __init__(self, geometry, back_ground_color=(1.0, 1.0, 1.0), training=True):
        self.geometry = geometry
        self.training = training
        self.background_tensor = torch.tensor(back_ground_color, dtype=torch.float32, device=geometry.device)
        self._renderers = {}

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def _renderer(self, H, W, tanfov):
        key = (int(H), int(W), round(float(tanfov), 9))
        if key not in self._renderers:
            g = self.geometry
            self._renderers[key] = views.ViewRenderer(g.graph, g.topo, H, W, tanfov, method=g.skinning_method)
        return self._renderers[key]

    def batch_forward(self, batch: Dict) -> Dict:
        g = self.geometry
        B = int(batch["c2w"].shape[0])
        H, W = int(batch["height"]), int(batch["width"])
        w2c, full, _, fovy = batch_cameras(batch, g.device)
        fov0 = float(fovy[0])
        if not bool((fovy == fovy[0]).all()):
            raise NotImplementedError("one fovy per batch (the shipped configurations use a fixed 20 degrees)")
        # background: white in training, inverted at evaluation (…temporal.py:96-103)
        bg = self.background_tensor if self.training else 1.0 - self.background_tensor
        ts = batch.get("timestamp")
        fi = batch.get("frame_indices")
        if ts is None and fi is None:
            raise NotImplementedError("the static stage goes through dreammesh4d_amd.diff_gaussian_rasterization")
        # node outputs once per distinct (timestamp, frame) of the batch (dynamic_sugar.py:367-405 caches them)
        dx, dr, ds, do, frame_index = g.timed_node_outputs(ts, fi)
        r = self._renderer(H, W, math.tan(0.5 * fov0))
        # per-view leaves whose .grad receives dL/d(screen-space mean), like the reference's `screenspace_points`
        vsp = [torch.zeros(g.n_gaussians, 3, device=g.device, requires_grad=True) for _ in range(B)]
        scales = g.timed_scales(ds, do) if g.d_scale else g.get_scaling      # per frame under d_scale (dynamic_sugar.py:717-720)
        out = views.render_views(r, dx, dr, ds, do, g.static_quaternions, scales, g.get_opacity.reshape(-1),
                                 g.get_points_rgb(), w2c, full, torch.cat([bg, bg]), frame_index=frame_index,
                                 means2D=torch.stack(vsp))
        g._deformed_vert_positions = out["vxyz"]                      # read by the mesh regularisers of the system
        has_rays = batch.get("rays_d") is not None
        res = compose_outputs(out["color"], out["depth"], out["alpha"], batch["rays_o"].to(g.device) if has_rays else None,
                              batch["rays_d"].to(g.device) if has_rays else None)
        res["viewspace_points"] = vsp
        res["visibility_filter"] = [out["radii"][b] > 0 for b in range(B)]
        res["radii"] = [out["radii"][b] for b in range(B)]
        return res

    __call__ = batch_forward

    def forward(self, viewpoint_camera: Camera, bg_color=None, scaling_modifier=1.0, override_color=None,
                compute_normal_from_dist=True, **kwargs) -> Dict:
        """Single view in the reference's signature (the matrices of `viewpoint_camera` are used as given)."""
        if scaling_modifier != 1.0 or override_color is not None:
            raise NotImplementedError("scaling_modifier / override_color are not used by the reference's systems")
        g = self.geometry
        H, W = int(viewpoint_camera.image_height), int(viewpoint_camera.image_width)
        bg = self.background_tensor if bg_color is None else bg_color.to(g.device)
        if not self.training:
            bg = 1.0 - bg
        ts = None if viewpoint_camera.timestamp is None else viewpoint_camera.timestamp.reshape(1)
        fi = None if viewpoint_camera.frame_idx is None else viewpoint_camera.frame_idx.reshape(1)
        dx, dr, ds, do, frame_index = g.timed_node_outputs(ts, fi)
        r = self._renderer(H, W, math.tan(0.5 * float(viewpoint_camera.FoVy)))
        vsp = torch.zeros(g.n_gaussians, 3, device=g.device, requires_grad=True)
        scales = g.timed_scales(ds, do) if g.d_scale else g.get_scaling      # per frame under d_scale (dynamic_sugar.py:717-720)
        out = views.render_views(r, dx, dr, ds, do, g.static_quaternions, scales, g.get_opacity.reshape(-1),
                                 g.get_points_rgb(), viewpoint_camera.world_view_transform[None],
                                 viewpoint_camera.full_proj_transform[None], torch.cat([bg, bg]),
                                 frame_index=frame_index, means2D=vsp[None])
        color, depth, alpha = out["color"][0], out["depth"][0], out["alpha"][0]
        mask = alpha > 0.99
        depth = _where_detached(depth, mask)
        mask3 = mask.expand(3, H, W)
        nd_raw = nd_map = None
        if compute_normal_from_dist and "rays_d" in kwargs:
            bi = kwargs.get("batch_idx", 0)
            xyz = kwargs["rays_o"][bi] + depth.permute(1, 2, 0) * kwargs["rays_d"][bi]
            nd_raw = F.normalize(depth_to_normal(xyz.permute(2, 0, 1)[None])[0], dim=0)
            nd_map = _where_detached(nd_raw * 0.5 * alpha + 0.5, mask3)
            nd_raw = _where_detached(nd_raw, mask3)
        n_raw = F.normalize(color[3:], dim=0)
        n_map = _where_detached(n_raw * 0.5 * alpha + 0.5, mask3)
        return {"render": color[:3].clamp(0, 1), "normal": n_map, "normal_from_dist": nd_map, "depth": depth,
                "mask": alpha, "viewspace_points": vsp, "visibility_filter": out["radii"][0] > 0,
                "radii": out["radii"][0], "raw_normal": _where_detached(n_raw, mask3), "raw_normal_from_dist": nd_raw}


class DiffSuGaRNormal:
    """The reference's ``diff-sugar-rasterizer-normal`` renderer (custom/threestudio-dreammesh4d/renderer/
    diff_sugar_rasterizer_normal.py:54-226) over a static ``sugar.SuGaR`` geometry: per view the RGB pass and the
    normal pass of the reference (:161-195) as ONE call of the drop-in operator with 6-channel colours (RGB | Gaussian
    normal; both passes share geometry, binning and blend), then the reference's epilogue (normal from depth, masks,
    detach rules :196-207).  Every static parameter receives its gradient through the operator's backward.
    Colours: the reference passes ``shs = get_features`` (the DC term clipped to +-color_clip) and lets the rasterizer
    evaluate max(SH_C0 sh + 0.5, 0) with a zero gradient where clamped; ``SuGaR.get_rendered_rgb`` is exactly that as
    torch ops (same float32 values, same gradient mask), which is what lets the two passes share one 6-channel call."""

    def __init__(self, geometry, back_ground_color=(1.0, 1.0, 1.0), invert_bg_prob=1.0, training=True, seed=0):
        self.geometry = geometry
        self.training = training
        self.invert_bg_prob = invert_bg_prob
        self.background_tensor = torch.tensor(back_ground_color, dtype=torch.float32, device=geometry.device)
        self._rng = torch.Generator(device="cpu").manual_seed(seed)

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def gaussians(self):
        """The geometry's per-Gaussian attributes, evaluated ONCE for all the views of a batch (the reference re-evaluates
        the properties for every view; the autograd graph is the same, shared)."""
        g = self.geometry
        if hasattr(g, "render_attributes"):
            return g.render_attributes()
        return dict(xyz=g.get_xyz, opacity=g.get_opacity, scaling=g.get_scaling, rotation=g.get_rotation, rgb=g.get_rendered_rgb(),
                    normals=g.get_gs_normals)

    def forward(self, viewpoint_camera: Camera, bg_color=None, scaling_modifier=1.0, override_color=None,
                compute_normal_from_dist=True, gaussians=None, **kwargs) -> Dict:
        from . import diff_gaussian_rasterization as dgr

        g = self.geometry
        bg = self.background_tensor if bg_color is None else bg_color.to(g.device)
        # training: the background is inverted when a uniform draw EXCEEDS invert_bg_prob (…_normal.py:93-98)
        if self.training and float(torch.rand(1, generator=self._rng)) > self.invert_bg_prob:
            bg = 1.0 - bg
        H, W = int(viewpoint_camera.image_height), int(viewpoint_camera.image_width)
        ga = self.gaussians() if gaussians is None else gaussians
        means3D = ga["xyz"]
        vsp = torch.zeros_like(means3D, requires_grad=True)
        rgb = ga["rgb"] if override_color is None else override_color
        rs = dgr.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(0.5 * float(viewpoint_camera.FoVx)),
            tanfovy=math.tan(0.5 * float(viewpoint_camera.FoVy)), bg=torch.cat([bg, bg]), scale_modifier=scaling_modifier,
            viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
            sh_degree=g.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)
        color, radii, depth, alpha = dgr.GaussianRasterizer(rs)(
            means3D=means3D, means2D=vsp, opacities=ga["opacity"], colors_precomp=torch.cat([rgb, ga["normals"]], dim=1),
            scales=ga["scaling"], rotations=ga["rotation"])
        mask = alpha > 0.99
        mask3 = mask.expand(3, H, W)
        nd_raw = nd_map = None
        if compute_normal_from_dist and "rays_d" in kwargs:
            bi = kwargs.get("batch_idx", 0)
            xyz = kwargs["rays_o"][bi] + depth.permute(1, 2, 0) * kwargs["rays_d"][bi]                  # :162-163 (before the detach)
            nd_raw = F.normalize(depth_to_normal(xyz.permute(2, 0, 1)[None])[0], dim=0)
            nd_map = _where_detached(nd_raw * 0.5 * alpha + 0.5, mask3)
            nd_raw = _where_detached(nd_raw, mask3)
        n_raw = F.normalize(color[3:], dim=0)
        n_map = _where_detached(n_raw * 0.5 * alpha + 0.5, mask3)
        return {"render": color[:3].clamp(0, 1), "normal": n_map, "normal_from_dist": nd_map, "mask": alpha,
                "depth": _where_detached(depth, mask), "viewspace_points": vsp, "visibility_filter": radii > 0, "radii": radii,
                "raw_normal": n_raw, "raw_normal_from_dist": nd_raw}

    # The views of a batch as ONE operator call (gviews.render_gaussian_views: same kernels, no host synchronisation, the images
    # bit-identical to the per-view calls) where that applies: a HIP device, one field of view for the batch, no per-view
    # background draw.  `batched = False` (or DM4D_STATIC_BATCHED=0) keeps the loop of per-view drop-in operator calls.
    batched = os.environ.get("DM4D_STATIC_BATCHED", "1") != "0"

    def _views_renderer(self, N, H, W, tanfov):
        from . import gviews

        key = (N, H, W, round(tanfov, 9))
        cache = self.__dict__.setdefault("_gviews", {})
        if key not in cache:
            cache[key] = gviews.GaussianViews(N, H, W, tanfov, self.geometry.device)
        self.views_renderer = cache[key]          # (the most recent one: overflow_flag() / poll() for the training loop)
        return cache[key]

    def _render_views(self, w2c, full, fovy, B, H, W, with_viewspace_points=True):
        from . import gviews

        g = self.geometry
        ga = self.gaussians()
        r = self._views_renderer(g.n_gaussians, H, W, math.tan(0.5 * float(fovy[0])))
        bg = self.background_tensor
        vsp = [torch.zeros(g.n_gaussians, 3, device=g.device, requires_grad=True) for _ in range(B)] if with_viewspace_points else None
        colors6 = ga["colors6"] if "colors6" in ga else torch.cat([ga["rgb"], ga["normals"]], dim=1)
        out = gviews.render_gaussian_views(r, ga["xyz"], ga["rotation"], ga["scaling"], ga["opacity"], colors6,
                                           w2c, full, torch.cat([bg, bg]), means2D=None if vsp is None else torch.stack(vsp))
        return out, vsp

    def _batched_applies(self, fovy_in):
        g = self.geometry
        return (self.batched and g.device.type == "cuda" and not (self.training and self.invert_bg_prob < 1.0)
                and bool((fovy_in == fovy_in[0]).all()))

    def render_batch_raw(self, batch: Dict):
        """The rasterizer's images of a batch WITHOUT the epilogue -- {"color" [B,6,H,W] (RGB | normal), "depth", "alpha" [B,1,H,W],
        "radii"} -- for callers that evaluate the image-space terms themselves (static_head.static_head); None where the batched
        operator does not apply (then batch_forward is the way)."""
        g = self.geometry
        B, H, W = int(batch["c2w"].shape[0]), int(batch["height"]), int(batch["width"])
        w2c, full, _, fovy_in = batch_cameras(batch, g.device)
        if not self._batched_applies(fovy_in):
            return None
        return self._render_views(w2c, full, fovy_in, B, H, W, with_viewspace_points=False)[0]

    def _batch_forward_views(self, batch, w2c, full, fovy, B, H, W) -> Dict:
        g = self.geometry
        out, vsp = self._render_views(w2c, full, fovy, B, H, W)
        color, depth, alpha = out["color"], out["depth"], out["alpha"]
        mask = alpha > 0.99
        mask3 = mask.expand(B, 3, H, W)
        res = {}
        if batch.get("rays_d") is not None:
            xyz = batch["rays_o"].to(g.device) + depth.permute(0, 2, 3, 1) * batch["rays_d"].to(g.device)      # (before the detach, :162-163)
            nd = F.normalize(depth_to_normal(xyz.permute(0, 3, 1, 2)), dim=1)
            res["comp_normal_from_dist"] = _where_detached(nd * 0.5 * alpha + 0.5, mask3).permute(0, 2, 3, 1)
        n = F.normalize(color[:, 3:], dim=1)
        res["comp_normal"] = _where_detached(n * 0.5 * alpha + 0.5, mask3).permute(0, 2, 3, 1)
        res["comp_rgb"] = color[:, :3].clamp(0, 1).permute(0, 2, 3, 1)
        res["comp_depth"] = _where_detached(depth, mask).permute(0, 2, 3, 1)
        res["comp_mask"] = alpha.permute(0, 2, 3, 1)
        res["viewspace_points"] = vsp
        res["visibility_filter"] = [out["radii"][b] > 0 for b in range(B)]
        res["radii"] = [out["radii"][b] for b in range(B)]
        return res

    def batch_forward(self, batch: Dict) -> Dict:
        """``GaussianBatchRenderer.batch_forward`` (renderer/gaussian_batch_renderer.py:9-122) for the static geometry."""
        g = self.geometry
        B = int(batch["c2w"].shape[0])
        H, W = int(batch["height"]), int(batch["width"])
        w2c, full, center, fovy_in = batch_cameras(batch, g.device)      # (fovy compared on the host when it arrives there: no synchronisation)
        fovy = fovy_in.expand(B)
        if self._batched_applies(fovy_in):
            return self._batch_forward_views(batch, w2c, full, fovy_in, B, H, W)
        outs = []
        ga = self.gaussians()
        for b in range(B):
            cam = Camera(FoVx=float(fovy[b]), FoVy=float(fovy[b]), image_width=W, image_height=H, world_view_transform=w2c[b],
                         full_proj_transform=full[b], camera_center=center[b], timestamp=None, frame_idx=None)
            kw = {"gaussians": ga}
            if batch.get("rays_d") is not None:
                kw.update(rays_o=batch["rays_o"].to(g.device), rays_d=batch["rays_d"].to(g.device), batch_idx=b)
            outs.append(self.forward(cam, self.background_tensor, **kw))
        st = lambda k: torch.stack([o[k] for o in outs]).permute(0, 2, 3, 1)
        res = {"comp_rgb": st("render"), "comp_normal": st("normal"), "comp_depth": st("depth"), "comp_mask": st("mask"),
               "viewspace_points": [o["viewspace_points"] for o in outs], "visibility_filter": [o["visibility_filter"] for o in outs],
               "radii": [o["radii"] for o in outs]}
        if outs[0]["normal_from_dist"] is not None:
            res["comp_normal_from_dist"] = st("normal_from_dist")
        return res

    __call__ = batch_forward
