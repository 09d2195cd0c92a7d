"""Batched (frame, view) renderer: the fast path of the hot loop.

One ``render_views`` call = ``GaussianBatchRenderer.batch_forward`` of the reference
(custom/threestudio-dreammesh4d/renderer/gaussian_batch_renderer.py:9-122) for a whole batch:
for every view, sparse-control skinning at the view's timestamp, face->Gaussian transform, and the
RGB + normal rasterizer passes of ``DiffGaussian.forward``
(.../renderer/diff_sugar_rasterizer_temporal.py:161-217) -- as ONE C call (8 kernel launches that
each cover all views, no host sync), and one C call for the backward.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .ops import DEFAULT_GRAD_MODE, GRAD_MODES, METHODS, DeformGraph, MeshTopology

vp = C.c_void_p


ViewsStruct, ViewsGrads = _lib.ViewsStruct, _lib.ViewsGrads


def _p(t):
    return None if t is None else t.data_ptr()


def _f32(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


class ViewRenderer:
    """Static scene description + capacity policy for ``render_views``."""

    def __init__(self, graph: DeformGraph, topo: MeshTopology, image_height, image_width, tanfov, method="hybrid",
                 scale_modifier=1.0, capacity_factor=6.0, record_factor=3.0, grad_mode=None, deterministic=True):
        assert graph.device == topo.device and graph.V == topo.V
        self.graph, self.topo = graph, topo
        self.device = graph.device
        self.H, self.W = int(image_height), int(image_width)
        self.tanfov = float(tanfov)
        self.method, self.method_name = METHODS[method], method
        # gradient convention of the skinning / face->Gaussian backward: "pypose" (default, what the reference's autograd
        # returns) or "exact" (DESIGN.md "gradient convention")
        self.grad_mode = grad_mode or DEFAULT_GRAD_MODE
        self.method_flags = GRAD_MODES[self.grad_mode]
        self.scale_modifier = float(scale_modifier)
        # deterministic=True: (Gaussian, cell) backward records, no atomics, bit-reproducible gradients (the parity
        # tests).  False: (Gaussian, tile) records summed in LDS with float atomics -- the training mode: 3.4x fewer
        # records, gradients equal up to the order of float additions (include/dm4d.h, dm4d_views.record_mode)
        self.deterministic = bool(deterministic)
        # True: the backward does not materialise the per-VIEW Gaussian gradients (dL/dmeans3D, dL/drotations, dL/dcolors):
        # the record gather and the face backward run as ONE kernel (csrc/gather_face.hip), a thread per (view, Gaussian) writing
        # per-view corner records that the vertex kernel sums (round 4; the node gradients equal the two-kernel path's up to the
        # order of the float additions over a frame's views).  The training loops (bench.py, DynamicStage) switch it on; off by
        # default so that `last_grads` holds every per-view gradient for callers that look at them.
        self.fuse_face_backward = False
        # True: the caller declares that NO loss reads channels 3..5 of `color` (the normal image): their upstream gradient is not read
        # and nothing flows through the normals (dm4d_views_backward_rgb; the reference's autograd does the same when every normal
        # weight is 0, configs/sugar_dynamic_dg.yaml:145-157).  DynamicStage sets it from its loss weights; bench.py's headline step
        # keeps both passes' backward (SURVEY.md section 8d credits them).
        self.rgb_gradient_only = False
        self.N = topo.F * topo.G
        self.capacity = max(int(capacity_factor * self.N), 1 << 16)
        # backward records = (Gaussian, 4x4-pixel cell) pairs; ~2 per duplicate for mesh-bound splats
        self.record_capacity = int(record_factor * self.capacity)
        self.calibrated = False   # capacities checked against the first forward (render_views)
        self.headroom = 1.5       # capacities are kept >= headroom x the largest count seen (cameras / strain move D)
        self._pending = None      # (pinned int32 [B, 4], event): counters of an earlier forward on their way to the host
        self.last = None   # (ViewsStruct, keep-alive) of the most recent forward, for check()
        self._ws_pool = []     # recycled (geom, binning, image) workspace sets, keyed by (B, capacity)
        self._scratch = {}     # persistent backward scratch, keyed by (B, capacity)

    def _take_ws(self, B):
        L = _lib.lib()
        key = (B, self.capacity)
        for i, (k, ws) in enumerate(self._ws_pool):
            if k == key:
                del self._ws_pool[i]
                return ws
        dev, N, H, W = self.device, self.N, self.H, self.W
        return dict(key=key,
                    geom=torch.empty(L.dm4d_views_geom_bytes(B, N, H, W), dtype=torch.uint8, device=dev),
                    binning=torch.empty(L.dm4d_views_binning_bytes(B, self.capacity), dtype=torch.uint8, device=dev),
                    image=torch.empty(L.dm4d_views_image_bytes(B, H, W), dtype=torch.uint8, device=dev))

    def _give_ws(self, ws):
        if len(self._ws_pool) < 4:
            self._ws_pool.append((ws["key"], ws))

    def _bwd_scratch(self, B, record_capacity):
        L = _lib.lib()
        key = (B, record_capacity)
        if key not in self._scratch:
            g, t, dev = self.graph, self.topo, self.device
            self._scratch = {key: dict(
                grad=torch.empty(L.dm4d_views_grad_bytes(B, record_capacity), dtype=torch.uint8, device=dev),
                skin=torch.empty(L.dm4d_views_skin_scratch_bytes(B, g.V, g.K), dtype=torch.uint8, device=dev),
                face=torch.empty(L.dm4d_views_face_scratch_bytes(B, t.F), dtype=torch.uint8, device=dev))}
        return self._scratch[key]

    def counters_i32(self):
        """[B, 4] int32 device view of the last forward's counters {D, duplicate overflow, R, record overflow}
        (csrc/raster.h GeomCounter; every view's geom workspace starts with them)."""
        vs, ws = self.last
        per_view = ws["geom"].numel() // vs.B
        return ws["geom"].view(torch.int32).view(vs.B, per_view // 4)[:, :4]

    def overflow_flag(self):
        """float32 scalar on the device: 1.0 if the last forward dropped duplicates or backward records, else 0.0 -- no host
        sync.  Training loops hand it to the fused AdamW as `found_inf` (the step is then skipped on the device, the way
        a GradScaler skips a step) and call poll() at their leisure."""
        c = self.counters_i32()
        return ((c[:, 1] | c[:, 3]) != 0).any().to(torch.float32)

    def poll(self):
        """Sync-free capacity monitor: queues an async copy of the last forward's counters into pinned memory and looks at
        the copy queued by the PREVIOUS call if it has landed.  Raises Dm4dError (after enlarging the capacities for the
        following forwards) if that forward overflowed; otherwise keeps the capacities >= headroom x the counts seen."""
        done = None
        if self._pending is not None and self._pending[1].query():
            done, self._pending = self._pending[0], None
        if self._pending is None and self.last is not None:
            c = self.counters_i32()
            host = torch.empty(c.shape, dtype=torch.int32, pin_memory=True)
            host.copy_(c, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._pending = (host, ev)
        if done is None:
            return None
        d, r_ = int(done[:, 0].max()), int(done[:, 2].max())
        over = bool((done[:, 1] != 0).any() or (done[:, 3] != 0).any())
        self._grow(d, r_)
        if over:
            e = _lib.Dm4dError(f"a forward overflowed its workspaces (num_rendered {d}, records {r_}); capacities raised to "
                               f"{self.capacity} / {self.record_capacity}; the optimiser step of that iteration was skipped "
                               "on the device if overflow_flag() was given to it")
            e.overflow = True      # (callers that carry on after a skipped step tell it from other errors by this)
            raise e
        return d, r_

    def _grow(self, d, r_):
        if d * self.headroom > self.capacity:
            self.capacity = int(d * self.headroom) + 1024
        if r_ * self.headroom > self.record_capacity:
            self.record_capacity = int(r_ * self.headroom) + 1024

    def check(self):
        """Host check of the duplicate-list and record capacities (syncs).  Returns num_rendered per view; raises
        on overflow after enlarging the capacity for subsequent calls."""
        if self.last is None:
            return None
        L = _lib.lib()
        vs, _ = self.last
        B = vs.B
        nr = (C.c_int64 * B)()
        nrec = (C.c_int64 * B)()
        ov = (C.c_int32 * B)()
        with torch.cuda.device(self.device):
            _lib.check(L.dm4d_views_counters(C.byref(vs), nr, nrec, ov,
                                             torch.cuda.current_stream(self.device).cuda_stream), "dm4d_views_counters")
        nr, nrec = list(nr), list(nrec)
        self.last_num_records = nrec
        if any(o & 1 for o in ov):
            self.capacity = int(max(nr) * 1.5) + 1024
            self.record_capacity = max(self.record_capacity, int(max(nrec) * 1.5) + 1024)
            raise _lib.Dm4dError(f"duplicate list overflow: num_rendered {max(nr)} > capacity; "
                                 f"capacity raised to {self.capacity}, re-run the step")
        if any(o & 2 for o in ov):
            self.record_capacity = int(max(nrec) * 1.5) + 1024
            raise _lib.Dm4dError(f"backward record overflow: {max(nrec)} records > record_capacity; "
                                 f"raised to {self.record_capacity}, re-run the step")
        self._grow(max(nr), max(nrec))     # headroom for the forwards that follow
        return nr


class _RenderViews(torch.autograd.Function):
    @staticmethod
    def forward(ctx, r, dx, dr, ds, do, q_static, scales, opacities, rgb, viewmats, projmats, bg6, frame_index, means2D):
        L = _lib.lib()
        g, t, dev = r.graph, r.topo, r.device
        B = int(viewmats.shape[0])
        N, H, W = r.N, r.H, r.W
        f = dict(dtype=torch.float32, device=dev)
        fidx = None if frame_index is None else frame_index.detach().to(device=dev, dtype=torch.int32).contiguous()
        NF = B if fidx is None else int(dx.shape[0])      # frames: skinning + face transform run once per frame
        if fidx is not None and (tuple(fidx.shape) != (B,) or not 0 < NF <= B):
            raise ValueError(f"frame_index must be [{B}] with 1..{B} frames, got {tuple(fidx.shape)} / {NF} frames")
        keep = dict(dx=_f32(dx), dr=_f32(dr), ds=_f32(ds), do=_f32(do), qs=_f32(q_static), sc=_f32(scales),
                    op=_f32(opacities).reshape(-1), rgb=_f32(rgb), vm=_f32(viewmats).reshape(B, 16),
                    pm=_f32(projmats).reshape(B, 16), bg=_f32(bg6).reshape(6))
        for k, want in (("dx", (NF, g.M, 3)), ("dr", (NF, g.M, 4))):
            if tuple(keep[k].shape) != want:
                raise ValueError(f"{k} must be {want}, got {tuple(keep[k].shape)}")
        # scales [N,3] (shared by all views) or [NF,N,3] (per frame: the reference's d_scale branch)
        sc_per_frame = keep["sc"].dim() == 3
        if tuple(keep["sc"].shape) not in ((N, 3), (NF, N, 3)):
            raise ValueError(f"scales must be [{N},3] or [{NF},{N},3], got {tuple(keep['sc'].shape)}")
        out = dict(vxyz=torch.empty(NF, g.V, 3, **f), vrot=torch.empty(NF, g.V, 4, **f), means=torch.empty(NF, N, 3, **f),
                   rots=torch.empty(NF, N, 4, **f), colors=torch.empty(NF, N, 6, **f),
                   radii=torch.empty(B, N, dtype=torch.int32, device=dev), color=torch.empty(B, 6, H, W, **f),
                   depth=torch.empty(B, 1, H, W, **f), alpha=torch.empty(B, 1, H, W, **f))
        cap, rcap = r.capacity, r.record_capacity
        ws = r._take_ws(B)
        vs = ViewsStruct(B, N, t.F, t.G, g.V, g.M, g.K, r.method | r.method_flags, H, W, r.tanfov, r.tanfov, r.scale_modifier, cap, rcap,
                         _p(keep["bg"]), _p(keep["vm"]), _p(keep["pm"]), _p(g.verts), _p(g.nbr_idx), _p(g.nbr_w),
                         _p(keep["dx"]), _p(keep["dr"]), _p(keep["ds"]), _p(keep["do"]), _p(t.faces), _p(keep["qs"]),
                         _p(keep["sc"]), _p(keep["op"]), _p(keep["rgb"]), _p(out["vxyz"]), _p(out["vrot"]),
                         _p(out["means"]), _p(out["rots"]), _p(out["colors"]), _p(out["radii"]), _p(out["color"]),
                         _p(out["depth"]), _p(out["alpha"]), _p(ws["geom"]), _p(ws["binning"]), _p(ws["image"]), _p(fidx), NF,
                         1 if sc_per_frame else 0, 0 if r.deterministic else 1)
        keep["fidx"] = fidx
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_views_forward(C.byref(vs), torch.cuda.current_stream(dev).cuda_stream),
                       "dm4d_views_forward")
        # Tensors RETURNED from forward must not be kept as plain ctx attributes: output -> grad_fn -> ctx -> output is
        # a reference cycle through C++ that the Python GC cannot break (it leaked every step's graph: ~150 MB per
        # step).  They go through save_for_backward; only the internal buffers stay on ctx.
        returned = ("color", "depth", "alpha", "radii", "vxyz", "vrot")
        ctx.save_for_backward(*[out[k] for k in returned])
        ctx.r, ctx.vs, ctx.keep, ctx.ws = r, vs, keep, ws
        ctx.internal = {k: v for k, v in out.items() if k not in returned}
        ctx.shapes = (dx.shape, dr.shape, None if ds is None else ds.shape, None if do is None else do.shape,
                      scales.shape, opacities.shape, rgb.shape)
        ctx.need_static = any(x.requires_grad for x in (scales, opacities, rgb))
        r.last = (vs, ws)      # for check(): the counters live in ws["geom"]
        ctx.mark_non_differentiable(out["radii"])
        ctx.set_materialize_grads(False)     # outputs nobody used arrive as None, not as zero tensors (a fill launch each)
        return out["color"], out["depth"], out["alpha"], out["radii"], out["vxyz"], out["vrot"]

    @staticmethod
    def backward(ctx, g_color, g_depth, g_alpha, _g_radii, g_vxyz, g_vrot):
        L = _lib.lib()
        r, vs = ctx.r, ctx.vs
        if ctx.ws is None:
            raise RuntimeError("render_views: backward called a second time (retain_graph): the rasterizer workspaces of "
                               "this forward were recycled by the first backward; run the forward again")
        _alive = ctx.saved_tensors      # vs points into these (and into ctx.internal / ctx.keep)
        g, t, dev = r.graph, r.topo, r.device
        B, N, H, W = vs.B, r.N, r.H, r.W
        NF = vs.n_frames if ctx.keep["fidx"] is not None else B
        f = dict(dtype=torch.float32, device=dev)
        gc = _f32(g_color) if g_color is not None else torch.zeros(B, 6, H, W, **f)
        gd, ga = _f32(g_depth), _f32(g_alpha)
        gx, gr_ = _f32(g_vxyz), _f32(g_vrot)
        per_view = ctx.need_static or not r.fuse_face_backward
        want_m2 = per_view or ctx.needs_input_grad[13]
        o = dict(m2=torch.empty(B, N, 3, **f) if want_m2 else None, m3=torch.empty(B, N, 3, **f) if per_view else None,
                 rot=torch.empty(B, N, 4, **f) if per_view else None,
                 col=torch.empty(B, N, 6, **f) if per_view else None, op=torch.empty(B, N, **f) if ctx.need_static else None,
                 sc=torch.empty(B, N, 3, **f) if ctx.need_static else None, vx=torch.empty(NF, g.V, 3, **f),
                 vr=torch.empty(NF, g.V, 4, **f), dx=torch.empty(NF, g.M, 3, **f), dr=torch.empty(NF, g.M, 4, **f),
                 ds=torch.empty(NF, g.M, 6, **f) if ctx.keep["ds"] is not None else None,
                 do=torch.empty(NF, g.M, **f) if ctx.keep["do"] is not None else None)
        scr = r._bwd_scratch(B, vs.record_capacity)
        gs = ViewsGrads(_p(gc), _p(gd), _p(ga), _p(gx), _p(gr_), _p(g.csr_off), _p(g.csr_items), _p(t.csr_off),
                        _p(t.csr_items), _p(scr["grad"]), _p(scr["skin"]), _p(scr["face"]), _p(o["m2"]), _p(o["m3"]),
                        _p(o["rot"]), _p(o["col"]), _p(o["op"]), _p(o["sc"]), _p(o["vx"]), _p(o["vr"]), _p(o["dx"]),
                        _p(o["dr"]), _p(o["ds"]), _p(o["do"]))
        # renderer.rgb_gradient_only: the caller declares that no loss reads the normal image (channels 3..5 of `color`): their upstream
        # gradient is not read and the blend backward carries 5 per-entry sums instead of 8 (dm4d_views_backward_rgb) -- where the lean
        # configuration it needs applies (static appearance frozen, no depth gradient, cell records); otherwise the full call
        rgb = bool(getattr(r, "rgb_gradient_only", False)) and not ctx.need_static and gd is None and r.deterministic
        fn = L.dm4d_views_backward_rgb if rgb else L.dm4d_views_backward
        with torch.cuda.device(dev):
            _lib.check(fn(C.byref(vs), C.byref(gs), torch.cuda.current_stream(dev).cuda_stream),
                       "dm4d_views_backward_rgb" if rgb else "dm4d_views_backward")
        s = ctx.shapes
        r._give_ws(ctx.ws)   # stream-ordered reuse by the next forward is safe
        ctx.ws = ctx.internal = None
        r.last_grads = o   # per-view gradients (means2D etc.) for callers that want them
        if not ctx.need_static:
            g_sc = None
        elif len(s[4]) == 3:        # per-frame scales: a frame's gradient is the sum over its views
            fi = ctx.keep["fidx"]
            g_sc = o["sc"] if fi is None else torch.zeros(s[4], **f).index_add_(0, fi.long(), o["sc"])
        else:
            g_sc = o["sc"].sum(0).reshape(s[4])
        g_op = o["op"].sum(0).reshape(s[5]) if ctx.need_static else None
        g_rgb = o["col"][:, :, :3].sum(0).reshape(s[6]) if ctx.need_static else None
        g_m2 = o["m2"] if ctx.needs_input_grad[13] else None       # screen-space mean gradients ("viewspace_points")
        return (None, o["dx"].reshape(s[0]), o["dr"].reshape(s[1]), None if o["ds"] is None else o["ds"].reshape(s[2]),
                None if o["do"] is None else o["do"].reshape(s[3]), None, g_sc, g_op, g_rgb, None, None, None, None, g_m2)


def render_views(renderer: ViewRenderer, dx, dr, ds, d_opacity, q_static, scales, opacities, rgb, viewmats, projmats,
                 bg6, frame_index=None, means2D=None):
    """Returns dict: color [B,6,H,W] (RGB | normal), depth [B,1,H,W], alpha [B,1,H,W], radii [B,N] int32,
    vxyz [B,V,3], vrot [B,V,4].

    frame_index [B] (int): views that share a timestamp share its skinning and face->Gaussian transform (the
    reference caches them per timestamp within a step, geometry/dynamic_sugar.py:375-386): dx, dr, ds, d_opacity
    are then [n_frames, M, .] (the deformation network's output per distinct timestamp), view b renders frame
    frame_index[b], vxyz / vrot come back per frame, and the node gradients are summed over a frame's views.

    means2D [B,N,3] (optional, zeros with requires_grad): the reference's screen-space gradient carrier
    (``viewspace_points``); its gradient is dL/d(mean2D) of every view."""
    m = renderer.method
    args = (renderer, dx, dr, ds if m != 1 else None, d_opacity if m == 2 else None, q_static, scales, opacities, rgb,
            viewmats, projmats, bg6, frame_index, means2D)
    color, depth, alpha, radii, vxyz, vrot = _RenderViews.apply(*args)
    if not renderer.calibrated:
        # first call only: one host sync to size the duplicate / record capacities from the real counts
        # (both are counted exactly even when they overflow); afterwards check() is up to the caller
        try:
            renderer.check()
        except _lib.Dm4dError:
            color, depth, alpha, radii, vxyz, vrot = _RenderViews.apply(*args)
            renderer.check()
        renderer.calibrated = True
    return {"color": color, "depth": depth, "alpha": alpha, "radii": radii, "vxyz": vxyz, "vrot": vrot}
