"""B views of ONE set of Gaussians as one operator: the static stage's batch.

``GaussianBatchRenderer.batch_forward`` (custom/threestudio-dreammesh4d/renderer/gaussian_batch_renderer.py:21-76) loops over the
views of a batch and calls the renderer's ``forward`` for each (renderer/diff_sugar_rasterizer_normal.py:88-226: an RGB pass and a
normal pass of the rasterizer, a host synchronisation in each).  In the static stage every view of the batch renders the SAME
Gaussians -- the SuGaR geometry's properties -- so the batch is ONE call of ``dm4d_gviews_forward`` (include/dm4d.h: six launches
that each cover all views, no synchronisation) and one of ``dm4d_gviews_backward``; the per-view gradients of the shared Gaussians
are summed here.  Same kernels as the per-view drop-in operator (diff_gaussian_rasterization.py): the images are bit-identical
to B calls of it, the gradients equal up to the order of the sum over the views.
"""
import ctypes as C

import torch

from . import _lib
from .views import ViewRenderer, _f32, _p


class GaussianViews(ViewRenderer):
    """Workspaces + capacity policy (that of views.ViewRenderer: counters polled without a synchronisation, overflow_flag() for the
    optimiser, poll() enlarges) for ``render_gaussian_views``."""

    def __init__(self, n_gaussians, image_height, image_width, tanfov, device, scale_modifier=1.0, capacity_factor=6.0, record_factor=3.0,
                 deterministic=True):
        self.graph = self.topo = None
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.Dm4dError("render_gaussian_views runs on the HIP device (there is no CPU fallback in the product)")
        self.H, self.W = int(image_height), int(image_width)
        self.tanfov = float(tanfov)
        self.scale_modifier = float(scale_modifier)
        self.deterministic = bool(deterministic)
        self.N = int(n_gaussians)
        self.capacity = max(int(capacity_factor * self.N), 1 << 16)
        self.record_capacity = int(record_factor * self.capacity)
        self.calibrated = False
        self.headroom = 1.5
        self._pending = None
        self.last = None
        self._ws_pool = []
        self._scratch = {}

    def _bwd_scratch(self, B, record_capacity):
        key = (B, record_capacity)
        if key not in self._scratch:
            self._scratch = {key: dict(grad=torch.empty(_lib.lib().dm4d_views_grad_bytes(B, record_capacity), dtype=torch.uint8, device=self.device))}
        return self._scratch[key]

    def check(self):
        """Host check of the capacities against the last forward's counters (synchronises); raises after enlarging them."""
        if self.last is None:
            return None
        c = self.counters_i32().cpu()
        d, r_ = int(c[:, 0].max()), int(c[:, 2].max())
        over = bool((c[:, 1] != 0).any() or (c[:, 3] != 0).any())
        if over:
            self.capacity = max(self.capacity, int(d * 1.5) + 1024)
            self.record_capacity = max(self.record_capacity, int(r_ * 1.5) + 1024)
            raise _lib.Dm4dError(f"overflow: num_rendered {d}, records {r_}; capacities raised to {self.capacity} / {self.record_capacity}, re-run")
        self._grow(d, r_)
        return c[:, 0].tolist()


class _RenderGaussianViews(torch.autograd.Function):
    @staticmethod
    def forward(ctx, r, means3D, rotations, scales, opacities, colors, viewmats, projmats, bg6, means2D):
        L = _lib.lib()
        dev, N, H, W = r.device, r.N, r.H, r.W
        B = int(viewmats.shape[0])
        f = dict(dtype=torch.float32, device=dev)
        keep = dict(m=_f32(means3D), q=_f32(rotations), s=_f32(scales), o=_f32(opacities).reshape(-1), c=_f32(colors),
                    vm=_f32(viewmats).reshape(B, 16), pm=_f32(projmats).reshape(B, 16), bg=_f32(bg6).reshape(6))
        for k, want in (("m", (N, 3)), ("q", (N, 4)), ("s", (N, 3)), ("o", (N,)), ("c", (N, 6))):
            if tuple(keep[k].shape) != want or keep[k].device != dev:
                raise ValueError(f"render_gaussian_views: input {k} must be {want} on {dev}, got {tuple(keep[k].shape)} on {keep[k].device}")
        out = dict(radii=torch.empty(B, N, dtype=torch.int32, device=dev), color=torch.empty(B, 6, H, W, **f),
                   depth=torch.empty(B, 1, H, W, **f), alpha=torch.empty(B, 1, H, W, **f))
        ws = r._take_ws(B)
        vs = _lib.GViewsStruct(B, N, H, W, r.tanfov, r.tanfov, r.scale_modifier, 0 if r.deterministic else 1, r.capacity, r.record_capacity,
                               _p(keep["bg"]), _p(keep["vm"]), _p(keep["pm"]), _p(keep["m"]), _p(keep["q"]), _p(keep["s"]), _p(keep["o"]),
                               _p(keep["c"]), _p(out["radii"]), _p(out["color"]), _p(out["depth"]), _p(out["alpha"]), _p(ws["geom"]),
                               _p(ws["binning"]), _p(ws["image"]))
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_gviews_forward(C.byref(vs), torch.cuda.current_stream(dev).cuda_stream), "dm4d_gviews_forward")
        ctx.save_for_backward(out["color"], out["depth"], out["alpha"], out["radii"])      # (returned tensors: not as ctx attributes, views.py)
        ctx.r, ctx.vs, ctx.keep, ctx.ws = r, vs, keep, ws
        ctx.shapes = (means3D.shape, rotations.shape, scales.shape, opacities.shape, colors.shape)
        r.last = (vs, ws)
        ctx.mark_non_differentiable(out["radii"])
        ctx.set_materialize_grads(False)
        return out["color"], out["depth"], out["alpha"], out["radii"]

    @staticmethod
    def backward(ctx, g_color, g_depth, g_alpha, _g_radii):
        L = _lib.lib()
        r, vs = ctx.r, ctx.vs
        if ctx.ws is None:
            raise RuntimeError("render_gaussian_views: backward called a second time: the workspaces of this forward were recycled")
        _alive = ctx.saved_tensors
        dev, N, H, W, B = r.device, r.N, r.H, r.W, vs.B
        f = dict(dtype=torch.float32, device=dev)
        gc = _f32(g_color) if g_color is not None else torch.zeros(B, 6, H, W, **f)
        gd, ga = _f32(g_depth), _f32(g_alpha)
        o = dict(m2=torch.empty(B, N, 3, **f), m3=torch.empty(B, N, 3, **f), rot=torch.empty(B, N, 4, **f), sc=torch.empty(B, N, 3, **f),
                 op=torch.empty(B, N, **f), col=torch.empty(B, N, 6, **f))
        scr = r._bwd_scratch(B, vs.record_capacity)
        gs = _lib.GViewsGrads(_p(gc), _p(gd), _p(ga), _p(scr["grad"]), _p(o["m2"]), _p(o["m3"]), _p(o["rot"]), _p(o["sc"]), _p(o["op"]), _p(o["col"]))
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_gviews_backward(C.byref(vs), C.byref(gs), torch.cuda.current_stream(dev).cuda_stream), "dm4d_gviews_backward")
        r._give_ws(ctx.ws)
        ctx.ws = None
        r.last_grads = o
        s = ctx.shapes
        need = ctx.needs_input_grad
        red = lambda t, shape, on: t.sum(0).reshape(shape) if on else None
        return (None, red(o["m3"], s[0], need[1]), red(o["rot"], s[1], need[2]), red(o["sc"], s[2], need[3]), red(o["op"], s[3], need[4]),
                red(o["col"], s[4], need[5]), None, None, None, o["m2"] if need[9] else None)


def render_gaussian_views(renderer: GaussianViews, means3D, rotations, scales, opacities, colors6, viewmats, projmats, bg6, means2D=None):
    """B views of the Gaussians (means3D [N,3], rotations [N,4] (w,x,y,z), scales [N,3], opacities [N] or [N,1], colors6 [N,6] =
    RGB | normal) from the cameras viewmats / projmats [B,4,4] (row-vector convention, as the rasterizer settings hold them).
    Returns dict: color [B,6,H,W], depth [B,1,H,W], alpha [B,1,H,W], radii [B,N] int32.  means2D [B,N,3] (optional, zeros with
    requires_grad): the screen-space gradient carrier (``viewspace_points``) of every view."""
    args = (renderer, means3D, rotations, scales, opacities, colors6, viewmats, projmats, bg6, means2D)
    color, depth, alpha, radii = _RenderGaussianViews.apply(*args)
    if not renderer.calibrated:
        try:                                   # first call only: one synchronisation to size the capacities from the real counts
            renderer.check()
        except _lib.Dm4dError:
            color, depth, alpha, radii = _RenderGaussianViews.apply(*args)
            renderer.check()
        renderer.calibrated = True
    return {"color": color, "depth": depth, "alpha": alpha, "radii": radii}
