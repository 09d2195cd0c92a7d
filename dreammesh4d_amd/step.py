"""The STEP OBJECT of the dynamic stage: ``DeformationNetwork.node_outputs`` + ``views.render_views`` of one training step as
ONE C call each way (csrc/step.hip, include/dm4d.h ``dm4d_step_*``) on buffers that are allocated once.

What a step of the reference does per (frame, view) -- the deformation query of the graph nodes at the frame's timestamp
(custom/threestudio-dreammesh4d/geometry/dynamic_sugar.py:367-431), skinning + face -> Gaussian transform (:487-613,657-743), the
RGB and the normal rasterizer pass (renderer/diff_sugar_rasterizer_temporal.py:161-217), and all of it backward -- is ~21 kernel
launches and ~1.0 ms of GPU time for 8 views of 512 x 512 over 200k Gaussians.  Through ``node_outputs`` + ``render_views`` the
host needs 0.76-0.88 ms to ENQUEUE that (two autograd Functions, ~60 tensor allocations, two ctypes structs of ~45 fields each,
22 AccumulateGrad nodes): the step is host-bound and kernel time saved does not show.  ``DynamicStep`` keeps

  * every intermediate and output in persistent tensors (no allocation per step; the outputs returned by a call are fresh
    tensor OBJECTS on the same storage: they are overwritten by the next call, like a CUDA-graph's static outputs),
  * the parameter gradients in persistent buffers installed as ``.grad`` (the contract of ``grads_in_place``: the caller drops
    its gradients with ``p.grad = None`` / ``zero_grad(set_to_none=True)`` every step; gradient ACCUMULATION over several
    backwards is not supported here -- use ``render_views`` for that),
  * the structs inside the library (``dm4d_step_create``).

It covers the dynamic stage as shipped: static appearance frozen (``static_learnable: false``, dynamic_sugar.py:79-87), the fused
64-wide deformation MLP, deterministic (Gaussian, cell) records.  Results are bit-identical to ``node_outputs`` +
``render_views`` FOR THE SAME ``renderer.fuse_face_backward`` SETTING (the same kernels in the same order; tests/test_step_gpu.py).
With ``fuse_face_backward`` on (``DynamicStage`` and ``bench.py`` switch it on) BOTH paths run csrc/gather_face.hip, whose per-view
corner records are summed by the vertex kernel in another order than the two-kernel path adds the per-view Gaussian gradients: the
images are still bit-identical, the parameter gradients equal the two-kernel path's up to the order of those additions (2e-6 of
the tensor's scale, tests/test_views_gpu.py, tests/test_step_gpu.py)."""
import ctypes as C

import torch

from . import _lib
from . import hexplane as hx


def _p(t):
    return None if t is None else t.data_ptr()


def _on(t, dev):
    """t lives on `dev` ("cuda" without an index = the current device's tensors pass)."""
    return t.device.type == dev.type and (dev.index is None or t.device.index is None or t.device.index == dev.index)


class _StepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, st, anchor, times, vm, pm, fidx):
        st._forward(times, vm, pm, fidx)
        ctx.st = st
        ctx.serial = st.serial
        ctx.set_materialize_grads(False)
        o = st.out
        # fresh tensor objects on the persistent storage (autograd attaches this call's node to them)
        return o["color"].detach(), o["depth"].detach(), o["alpha"].detach(), o["vxyz"].detach(), o["vrot"].detach()

    @staticmethod
    def backward(ctx, g_color, g_depth, g_alpha, g_vxyz, g_vrot):
        st = ctx.st
        if ctx.serial != st.serial:
            raise RuntimeError("DynamicStep: backward of a step whose buffers a later forward has already overwritten (one step "
                               "in flight at a time: call backward before the next forward, or use views.render_views)")
        st._backward(g_color, g_depth, g_alpha, g_vxyz, g_vrot)
        return None, None, None, None, None, None


class DynamicStep:
    """``step = DynamicStep(renderer, net, nodes, q_static, scales, opacities, rgb, bg6, n_views, n_frames)`` once;
    ``out = step(frame_t, viewmats, projmats, frame_index)`` per step -> dict(color [B,6,H,W], depth [B,1,H,W], alpha
    [B,1,H,W], radii [B,N], vxyz [NF,V,3], vrot [NF,V,4]); ``loss.backward()`` then leaves the gradients of the deformation
    network's parameters in ``p.grad``.

    renderer: views.ViewRenderer (capacities are taken from it; call ``calibrate()`` / let the first call do it).
    frame_t [NF] float32 timestamps in (0, 1); viewmats / projmats [B,4,4] float32; frame_index [B] int32 (or None when
    NF == B: every view its own frame)."""

    def __init__(self, renderer, net, nodes, q_static, scales, opacities, rgb, bg6, n_views, n_frames=None):
        from .deformation import DeformationNetwork

        assert isinstance(net, DeformationNetwork)
        self.r, self.net = renderer, net
        self.dev = renderer.device
        self.B = int(n_views)
        self.NF = int(n_frames) if n_frames is not None else self.B
        if any(t.requires_grad for t in (scales, opacities, rgb, q_static)):
            raise ValueError("DynamicStep renders with the static appearance FROZEN (the dynamic stage); learnable scales / opacities / "
                             "rgb go through views.render_views")
        if not renderer.deterministic:
            raise ValueError("DynamicStep uses the deterministic (Gaussian, cell) records")
        f32 = lambda t: t.detach().to(device=self.dev, dtype=torch.float32).contiguous()
        self.static = dict(qs=f32(q_static), sc=f32(scales), op=f32(opacities).reshape(-1), rgb=f32(rgb), bg=f32(bg6).reshape(6))
        if tuple(self.static["sc"].shape) != (renderer.N, 3):
            raise ValueError("DynamicStep: scales must be [N,3] (the d_scale branch goes through views.render_views)")
        self.plan = net.build_plan(nodes)
        d = net.deformation_net
        lin0 = d.feature_out[0]
        if not (lin0.out_features == 64 and lin0.in_features % 64 == 0 and lin0.in_features <= 256):
            raise ValueError("DynamicStep needs the fused 64-wide deformation MLP")
        # heads in the kernel's order: position, scales (strain), rotations, opacity (deformation.node_outputs)
        self.head_names = ["dx"] + ([] if d.no_ds else ["ds"]) + ([] if d.no_dr else ["dr"]) + ([] if d.no_do else ["do"])
        heads = [d.pos_deform] + ([] if d.no_ds else [d.scales_deform]) + ([] if d.no_dr else [d.rotations_deform]) + \
                ([] if d.no_do else [d.opacity_deform])
        if "dr" not in self.head_names:
            raise ValueError("DynamicStep needs the rotation head (no_dr = False)")
        self.mlp = [lin0.weight, lin0.bias]
        for hd in heads:
            self.mlp += [hd.feature_out[0].main_stream.weight, hd.feature_out[0].main_stream.bias, hd.feature_out[1].weight, hd.feature_out[1].bias]
        self.planes = [p for grid in d.grid.grids for p in grid]
        for p in self.mlp:
            if p.dtype != torch.float32 or not p.is_contiguous() or not _on(p, self.dev):
                raise ValueError("deformation MLP parameters must be contiguous float32 on the renderer's device")
        self.params = [p for p in self.planes + self.mlp if p.requires_grad]
        if not self.params:
            raise ValueError("DynamicStep: no trainable parameter")
        self.handle = None
        self.key = None
        self.serial = 0
        self.out = None

    # ------------------------------------------------------------------ construction of the C object
    def _ptr_state(self):
        """Everything the library holds pointers to that the caller may replace (a parameter's `.data` swap, grown capacities)."""
        return (self.r.capacity, self.r.record_capacity, bool(self.r.fuse_face_backward)) + tuple(p.data_ptr() for p in self.planes) + \
            tuple(p.data_ptr() for p in self.mlp)

    def _build(self):
        L = _lib.lib()
        r, g, t, dev = self.r, self.r.graph, self.r.topo, self.dev
        B, NF, N, H, W, M = self.B, self.NF, r.N, r.H, r.W, self.plan.M
        S = self.plan.S
        f = dict(dtype=torch.float32, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        cap, rcap = r.capacity, r.record_capacity
        n_heads = len(self.head_names)
        dims = {"dx": 3, "ds": 6, "dr": 4, "do": 1}
        P, IN = NF * M, S * 32
        o = self.out = dict(
            vxyz=torch.empty(NF, g.V, 3, **f), vrot=torch.empty(NF, g.V, 4, **f), means=torch.empty(NF, N, 3, **f),
            rots=torch.empty(NF, N, 4, **f), colors=torch.empty(NF, N, 6, **f), radii=torch.empty(B, N, dtype=torch.int32, device=dev),
            color=torch.empty(B, 6, H, W, **f), depth=torch.empty(B, 1, H, W, **f), alpha=torch.empty(B, 1, H, W, **f))
        ws = self.ws = dict(
            geom=torch.empty(L.dm4d_views_geom_bytes(B, N, H, W), **u8), binning=torch.empty(L.dm4d_views_binning_bytes(B, cap), **u8),
            image=torch.empty(L.dm4d_views_image_bytes(B, H, W), **u8), grad=torch.empty(L.dm4d_views_grad_bytes(B, rcap), **u8),
            skin=torch.empty(L.dm4d_views_skin_scratch_bytes(B, g.V, g.K), **u8), face=torch.empty(L.dm4d_views_face_scratch_bytes(B, t.F), **u8),
            feat=torch.empty(P, IN, **f), h=torch.empty(P, 64, **f), y=torch.empty(n_heads, P, 64, **f), g_feat=torch.empty(P, IN, **f),
            samples=torch.empty(L.dm4d_hexplane_scratch_bytes(S, M, NF), **u8),
            net_scratch=torch.empty(L.dm4d_nodenet_scratch_bytes(S, M, NF, n_heads), **u8))
        node = self.node = {k: torch.empty(NF, M, dims[k], **f) for k in self.head_names}
        gnode = self.gnode = {k: torch.empty(NF, M, dims[k], **f) for k in self.head_names}
        pv = (lambda *sh: None) if r.fuse_face_backward else (lambda *sh: torch.empty(*sh, **f))
        bw = self.bw = dict(m2=pv(B, N, 3), m3=pv(B, N, 3), rot=pv(B, N, 4), col=pv(B, N, 6),
                            vx=torch.empty(NF, g.V, 3, **f), vr=torch.empty(NF, g.V, 4, **f))
        # parameter gradients: persistent (installed as .grad after every backward)
        pl = [p.detach() for p in self.planes]
        if self.plan.grad_buffers is None or any(b.shape != p.shape or b.stride() != p.stride() or b.device != p.device
                                                 for b, p in zip(self.plan.grad_buffers, pl)):
            self.plan.grad_buffers = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in pl]
        self.g_planes = self.plan.grad_buffers
        self.g_mlp = [torch.empty_like(p) for p in self.mlp]
        # what the skinning method reads of the node outputs (views.render_views: ds unless LBS, d_opacity for hybrid only)
        m = r.method
        use = {"dx": True, "dr": True, "ds": m != 1, "do": m == 2}
        d = _lib.StepDesc()
        v = d.views
        v.B, v.N, v.F, v.G, v.V, v.M, v.K, v.method = B, N, t.F, t.G, g.V, g.M, g.K, r.method | r.method_flags
        v.image_height, v.image_width = H, W
        v.tanfovx = v.tanfovy = r.tanfov
        v.scale_modifier = r.scale_modifier
        v.capacity, v.record_capacity = cap, rcap
        st = self.static
        v.bg, v.verts, v.nbr_idx, v.nbr_w = _p(st["bg"]), _p(g.verts), _p(g.nbr_idx), _p(g.nbr_w)
        v.dx, v.dr = _p(node["dx"]), _p(node["dr"])
        v.ds = _p(node["ds"]) if use["ds"] and "ds" in node else None
        v.d_opacity = _p(node["do"]) if use["do"] and "do" in node else None
        v.faces, v.q_static, v.scales, v.opacities, v.rgb = _p(t.faces), _p(st["qs"]), _p(st["sc"]), _p(st["op"]), _p(st["rgb"])
        v.vxyz, v.vrot, v.means3D, v.rotations, v.colors, v.radii = (_p(o[k]) for k in ("vxyz", "vrot", "means", "rots", "colors", "radii"))
        v.out_color, v.out_depth, v.out_alpha = _p(o["color"]), _p(o["depth"]), _p(o["alpha"])
        v.geom, v.binning, v.image = _p(ws["geom"]), _p(ws["binning"]), _p(ws["image"])
        v.n_frames, v.scales_per_frame, v.record_mode = NF, 0, 0
        gr = d.grads
        gr.node_csr_offsets, gr.node_csr_items, gr.vert_csr_offsets, gr.vert_csr_items = _p(g.csr_off), _p(g.csr_items), _p(t.csr_off), _p(t.csr_items)
        gr.grad_scratch, gr.skin_scratch, gr.face_scratch = _p(ws["grad"]), _p(ws["skin"]), _p(ws["face"])
        # renderer.fuse_face_backward: the per-VIEW Gaussian gradients are not materialised (record gather + face backward as one
        # kernel, csrc/gather_face.hip); r.last_grads then only holds the vertex / node gradients
        self.fused_face = bool(r.fuse_face_backward)
        if not self.fused_face:
            gr.dL_dmeans2D, gr.dL_dmeans3D, gr.dL_drotations, gr.dL_dcolors = _p(bw["m2"]), _p(bw["m3"]), _p(bw["rot"]), _p(bw["col"])
        gr.dL_dvxyz, gr.dL_dvrot = _p(bw["vx"]), _p(bw["vr"])
        gr.dL_ddx, gr.dL_ddr = _p(gnode["dx"]), _p(gnode["dr"])
        gr.dL_dds = _p(gnode["ds"]) if v.ds else None
        gr.dL_ddo = _p(gnode["do"]) if v.d_opacity else None
        d.S = S
        d.hex_flags = hx.plane_layout(pl) | 4            # DM4D_HEX_CHANNELS_LAST?, DM4D_HEX_TIMES_01
        d.hex_backward_flags = 2                         # DM4D_HEX_KEEP_SPATIAL: persistent gradient planes
        d.res, d.aabb_host = self.plan.res_c, self.plan.aabb_c
        self._pp = hx._plane_ptr_array(pl)
        self._gp = hx._plane_ptr_array(self.g_planes)
        d.planes, d.g_planes = C.cast(self._pp, C.c_void_p), C.cast(self._gp, C.c_void_p)
        d.nodes = _p(self.plan.nodes)
        w, gw = d.w, d.gw
        w.in_dim, w.width, w.n_heads = IN, 64, n_heads
        w.W0, w.b0 = _p(self.mlp[0]), _p(self.mlp[1])
        gw.W0, gw.b0 = _p(self.g_mlp[0]), _p(self.g_mlp[1])
        for k, name in enumerate(self.head_names):
            W1, b1, W2, b2 = self.mlp[2 + 4 * k: 6 + 4 * k]
            w.out_dim[k] = int(W2.shape[0])
            w.W1[k], w.b1[k], w.W2[k], w.b2[k] = _p(W1), _p(b1), _p(W2), _p(b2)
            gw.W1[k], gw.b1[k], gw.W2[k], gw.b2[k] = (_p(x) for x in self.g_mlp[2 + 4 * k: 6 + 4 * k])
            d.node_out[k] = _p(node[name])
            d.node_gout[k] = _p(gnode[name]) if use[name] else None      # a head the method ignores: zero upstream gradient
        d.feat, d.h_save, d.y_save, d.g_feat = _p(ws["feat"]), _p(ws["h"]), _p(ws["y"]), _p(ws["g_feat"])
        d.samples, d.net_scratch = _p(ws["samples"]), _p(ws["net_scratch"])
        sp, tp = self.plan.sp, self.plan.tp
        d.n_spatial, d.n_time = self.plan.n_sp, self.plan.n_tp
        d.sp_scale, d.sp_plane, d.sp_texel, d.sp_off, d.sp_item = (_p(sp[k]) for k in ("scale", "plane", "texel", "off", "item"))
        d.tp_scale, d.tp_plane, d.tp_col, d.tp_off, d.tp_item = (_p(tp[k]) for k in ("scale", "plane", "col", "off", "item"))
        if self.handle is not None:
            L.dm4d_step_destroy(self.handle)
            self.handle = None
        h = C.c_void_p()
        _lib.check(L.dm4d_step_create(C.byref(d), C.byref(h)), "dm4d_step_create")
        self.handle = h
        self.key = self._ptr_state()
        self._views = C.cast(L.dm4d_step_views(h), C.POINTER(_lib.ViewsStruct))
        # (ViewRenderer.check() / poll() / overflow_flag() read the counters of `last`)
        self._last = (self._views.contents, ws)

    def __del__(self):
        try:
            if self.handle is not None:
                _lib.lib().dm4d_step_destroy(self.handle)
                self.handle = None
        except Exception:      # noqa: BLE001  (interpreter shutdown)
            pass

    # ------------------------------------------------------------------ the two calls
    def _forward(self, times, vm, pm, fidx):
        if self.handle is None or self.key != self._ptr_state():
            self._build()
        dev, B, NF = self.dev, self.B, self.NF
        if times.dtype != torch.float32 or not times.is_contiguous() or not _on(times, dev) or times.numel() != NF:
            raise ValueError(f"DynamicStep: timestamps must be a contiguous float32 [{NF}] tensor on {dev}")
        for name, t_ in (("viewmats", vm), ("projmats", pm)):
            if t_.dtype != torch.float32 or not t_.is_contiguous() or not _on(t_, dev) or t_.numel() != 16 * B:
                raise ValueError(f"DynamicStep: {name} must be contiguous float32 [{B},4,4] on {dev}, got {tuple(t_.shape)} {t_.dtype} "
                                 f"{t_.device} contiguous={t_.is_contiguous()}")
        if fidx is None:
            if NF != B:
                raise ValueError("DynamicStep: frame_index is required when n_frames != n_views")
        elif fidx.dtype != torch.int32 or not fidx.is_contiguous() or not _on(fidx, dev) or fidx.numel() != B:
            raise ValueError(f"DynamicStep: frame_index must be a contiguous int32 [{B}] tensor on {dev}")
        self.serial += 1
        self._keep = (times, vm, pm, fidx)        # the library reads them again in the backward
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().dm4d_step_forward(self.handle, times.data_ptr(), vm.data_ptr(), pm.data_ptr(), _p(fidx),
                                                    torch.cuda.current_stream(dev).cuda_stream), "dm4d_step_forward")
        self.r.last = self._last

    def _backward(self, g_color, g_depth, g_alpha, g_vxyz, g_vrot):
        dev = self.dev
        for p in self.params:
            if p.grad is not None:
                raise RuntimeError("DynamicStep: a parameter still holds a gradient (drop the gradients with `p.grad = None` / "
                                   "zero_grad(set_to_none=True) every step; accumulation over several backwards: views.render_views)")
        c = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        gc, gd, ga, gx, gr_ = c(g_color), c(g_depth), c(g_alpha), c(g_vxyz), c(g_vrot)
        if gc is None:
            gc = torch.zeros_like(self.out["color"])
        with torch.cuda.device(dev):
            if bool(getattr(self.r, "rgb_gradient_only", False)) and gd is None and self.r.deterministic:
                # no loss reads the normal image: channels 3..5 of the upstream gradient are not read (views.py, dm4d_views_backward_rgb)
                _lib.check(_lib.lib().dm4d_step_backward_rgb(self.handle, gc.data_ptr(), _p(ga), _p(gx), _p(gr_),
                                                             torch.cuda.current_stream(dev).cuda_stream), "dm4d_step_backward_rgb")
            else:
                _lib.check(_lib.lib().dm4d_step_backward(self.handle, gc.data_ptr(), _p(gd), _p(ga), _p(gx), _p(gr_),
                                                         torch.cuda.current_stream(dev).cuda_stream), "dm4d_step_backward")
        for p, gb in zip(self.planes, self.g_planes):
            if p.requires_grad:
                p.grad = gb
        for p, gb in zip(self.mlp, self.g_mlp):
            if p.requires_grad:
                p.grad = gb
        self.r.last_grads = self.bw

    def __call__(self, frame_t, viewmats, projmats, frame_index=None):
        color, depth, alpha, vxyz, vrot = _StepFn.apply(self, self.params[0], frame_t, viewmats, projmats, frame_index)
        r = self.r
        if not r.calibrated:
            # first call only: one host sync to size the duplicate / record capacities from the real counts
            try:
                r.check()
            except _lib.Dm4dError:
                color, depth, alpha, vxyz, vrot = _StepFn.apply(self, self.params[0], frame_t, viewmats, projmats, frame_index)
                r.check()
            if self.key != self._ptr_state():       # check() keeps headroom: the capacities may have grown without an overflow
                color, depth, alpha, vxyz, vrot = _StepFn.apply(self, self.params[0], frame_t, viewmats, projmats, frame_index)
            r.calibrated = True
        return {"color": color, "depth": depth, "alpha": alpha, "radii": self.out["radii"], "vxyz": vxyz, "vrot": vrot}

    def node_outputs(self):
        """The deformation network's raw outputs of the last forward: dict of [NF, M, .] tensors (dx, dr, ds, do as available)."""
        return dict(self.node)
