"""HexPlane + MLP deformation network of the dynamic stage.

Host-side mirror (PyTorch ops, any device) of
custom/threestudio-dreammesh4d/geometry/deformation.py: `DeformationNetwork` ->
`Deformation` -> `HexPlaneField` (:177-248), queried through `forward_dynamic_delta`
(:430-436,538-539) by `DynamicSuGaRModel._get_timed_dg_attributes`
(geometry/dynamic_sugar.py:420-431) on the ~1000 deformation-graph nodes.

Module / parameter names follow the reference so its checkpoints load with
`load_state_dict` (README.md:88: the dynamic stage starts from a static-stage ckpt and saves its own):
    timenet.{0,2}.*                                       (constructed, optimised, never used: :489-493)
    deformation_net.grid.aabb, deformation_net.grid.grids.<scale>.<plane>
    deformation_net.feature_out.0.*
    deformation_net.{pos,scales,rotations,opacity}_deform.feature_out.{0.main_stream,1}.*
Parity is pinned by tests/golden/deformation_small.npz (outputs + parameter gradients computed by the
reference file itself; generator: tests/golden/make_golden.py).

35,755,892 parameters at the shipped configuration (resolution [64,64,64,25], multires [1,2,4,8]):
143 MB of float32 gradients per iteration -- the payload of the data-parallel all-reduce.
"""
import itertools

import torch
import torch.nn as nn
import torch.nn.functional as F

PLANE_AXES = list(itertools.combinations(range(4), 2))   # (x,y) (x,z) (x,t) (y,z) (y,t) (z,t)


class HexPlaneField(nn.Module):
    def __init__(self, bounds=1.0, resolution=(64, 64, 64, 25), multires=(1, 2, 4, 8), channels=32):
        super().__init__()
        # note the order: [[+b],[−b]] -- normalisation maps x to −x/b (deformation.py:186-188)
        self.aabb = nn.Parameter(torch.tensor([[bounds] * 3, [-bounds] * 3], dtype=torch.float32), requires_grad=False)
        self.grids = nn.ModuleList()
        for mult in multires:
            reso = [int(r) * int(mult) for r in resolution[:3]] + [int(resolution[3])]
            planes = nn.ParameterList()
            for a0, a1 in PLANE_AXES:
                w = torch.empty(1, channels, reso[a1], reso[a0])
                if 3 in (a0, a1):
                    nn.init.ones_(w)                      # time planes start at 1 (:132-133)
                else:
                    nn.init.uniform_(w, a=0.1, b=0.5)
                # Same [1,C,H,W] parameter as the reference (names, shapes, state dict), held in channels_last
                # memory format: the C channels of a texel are one 128-byte line for the HIP query
                # (csrc/hexplane.hip), instead of C words H*W*4 bytes apart.
                planes.append(nn.Parameter(w.contiguous(memory_format=torch.channels_last)))
            self.grids.append(planes)
        self.feat_dim = channels * len(multires)

    def forward(self, pts, t):
        """pts [P,3], t [P,1] in [-1,1] -> features [P, channels * n_scales]."""
        lo, hi = self.aabb[0], self.aabb[1]
        x = (pts - lo) * (2.0 / (hi - lo)) - 1.0
        x4 = torch.cat([x, t], dim=-1)                    # [P,4]
        feats = []
        for planes in self.grids:
            acc = None
            for plane, (a0, a1) in zip(planes, PLANE_AXES):
                uv = x4[:, [a0, a1]].view(1, 1, -1, 2)    # grid_sample: (x=width=a0, y=height=a1)
                s = F.grid_sample(plane, uv, mode="bilinear", padding_mode="border", align_corners=True)
                s = s.reshape(plane.shape[1], -1).t()     # [P, channels]
                acc = s if acc is None else acc * s
            feats.append(acc)
        return torch.cat(feats, dim=-1)


class _LinearRes(nn.Module):
    def __init__(self, width):
        super().__init__()
        self.main_stream = nn.Linear(width, width)

    def forward(self, x):
        x = F.relu(x)
        return x + self.main_stream(x)


class _Head(nn.Module):
    """relu -> residual linear -> linear, zero-initialised (deformation.py:285-305,507-512)."""

    def __init__(self, width, out_dim):
        super().__init__()
        self.feature_out = nn.Sequential(_LinearRes(width), nn.Linear(width, out_dim))

    def zero_(self):
        for m in self.feature_out.modules():
            if isinstance(m, nn.Linear):
                nn.init.zeros_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, h):
        return self.feature_out(h)


class _Deformation(nn.Module):
    def __init__(self, width, grid, no_ds, no_dr, no_do):
        super().__init__()
        self.grid = grid
        self.feature_out = nn.Sequential(nn.Linear(grid.feat_dim, width))   # defor_depth == 1
        self.pos_deform = _Head(width, 3)
        self.scales_deform = _Head(width, 6)
        self.rotations_deform = _Head(width, 4)
        self.opacity_deform = _Head(width, 1)
        self.no_ds, self.no_dr, self.no_do = no_ds, no_dr, no_do

    def forward_dynamic_delta(self, pts, t):
        h = self.feature_out(self.grid(pts[:, :3], t[:, :1])).float()
        dx = self.pos_deform(h)
        ds = None if self.no_ds else self.scales_deform(h)
        dr = None if self.no_dr else self.rotations_deform(h)
        do = None if self.no_do else self.opacity_deform(h)
        return dx, dr, ds, do


class _DeformMLP(torch.autograd.Function):
    """feature_out + the present heads as ONE fused HIP forward and ONE backward (csrc/deform_mlp.hip,
    C ABI dm4d_deform_mlp_*), instead of ~12 GEMV-sized linears + ~70 elementwise launches."""

    @staticmethod
    def forward(ctx, feat, n_heads, *params):
        import ctypes as C

        from . import _lib

        L = _lib.lib()
        dev = feat.device
        P, IN = int(feat.shape[0]), int(feat.shape[1])
        f = feat.detach().to(torch.float32).contiguous()
        ps = [p.detach() for p in params]
        for p in ps:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("deformation MLP parameters must be contiguous float32")
        W0, b0 = ps[0], ps[1]
        heads = [ps[2 + 4 * k: 6 + 4 * k] for k in range(n_heads)]      # (W1, b1, W2, b2) per head
        w = _lib.MlpWeights()
        w.in_dim, w.width, w.n_heads = IN, int(W0.shape[0]), n_heads
        w.W0, w.b0 = W0.data_ptr(), b0.data_ptr()
        outs = []
        for k, (W1, b1, W2, b2) in enumerate(heads):
            w.out_dim[k] = int(W2.shape[0])
            w.W1[k], w.b1[k], w.W2[k], w.b2[k] = W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr()
            outs.append(torch.empty(P, int(W2.shape[0]), dtype=torch.float32, device=dev))
        h = torch.empty(P, 64, dtype=torch.float32, device=dev)
        y = torch.empty(n_heads, P, 64, dtype=torch.float32, device=dev)
        scratch = torch.empty(L.dm4d_deform_mlp_scratch_bytes(P, IN, n_heads), dtype=torch.uint8, device=dev)
        optr = (C.c_void_p * 4)(*[o.data_ptr() for o in outs])
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_deform_mlp_forward(P, f.data_ptr(), C.byref(w), h.data_ptr(), y.data_ptr(), optr,
                                                 scratch.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                       "dm4d_deform_mlp_forward")
        ctx.w, ctx.keep, ctx.n_heads = w, (f, ps, h, y, scratch), n_heads
        ctx.set_materialize_grads(False)     # an unused head arrives as None (the C call takes NULL), not as a zero tensor
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g_outs):
        import ctypes as C

        from . import _lib

        L = _lib.lib()
        f, ps, h, y, scratch = ctx.keep
        dev = f.device
        P = int(f.shape[0])
        gs = [None if g is None else g.detach().to(torch.float32).contiguous() for g in g_outs]
        gptr = (C.c_void_p * 4)(*[None if g is None else g.data_ptr() for g in gs])
        g_feat = torch.empty_like(f)
        grads = [torch.empty_like(p) for p in ps]
        gw = _lib.MlpWeightsGrad()
        gw.W0, gw.b0 = grads[0].data_ptr(), grads[1].data_ptr()
        for k in range(ctx.n_heads):
            gw.W1[k], gw.b1[k], gw.W2[k], gw.b2[k] = (grads[2 + 4 * k + j].data_ptr() for j in range(4))
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_deform_mlp_backward(P, f.data_ptr(), C.byref(ctx.w), h.data_ptr(), y.data_ptr(), gptr,
                                                  g_feat.data_ptr(), C.byref(gw), scratch.data_ptr(),
                                                  torch.cuda.current_stream(dev).cuda_stream), "dm4d_deform_mlp_backward")
        return (g_feat, None, *grads)


class _NodeNetwork(torch.autograd.Function):
    """HexPlane query + MLP of the graph nodes as ONE operator (csrc/nodenet.hip, C ABI dm4d_nodenet_*): 1 launch forward,
    3 backward, the 2 t - 1 of the timestamps inside -- against 1 + 1 + 1 forward and 2 + 3 backward for
    ``hexplane._HexPlaneFeatures`` + ``_DeformMLP`` + the torch kernel in front.  Bit-identical results (the same kernel
    bodies).  Inputs: (plan, timestamps in [0,1] [B], grads_in_place, n_heads, n_planes, *planes, *mlp parameters)."""

    @staticmethod
    def forward(ctx, plan, timestamps, in_place, n_heads, n_planes, *params):
        import ctypes as C

        from . import _lib, hexplane as hx

        L = _lib.lib()
        planes, mlp = params[:n_planes], params[n_planes:]
        pl = [p.detach() for p in planes]
        ps = [p.detach() for p in mlp]
        for p in ps:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("deformation MLP parameters must be contiguous float32")
        dev = timestamps.device
        B, M, S = int(timestamps.shape[0]), plan.M, plan.S
        P, IN = B * M, S * 32
        t = timestamps.detach().to(torch.float32).contiguous()
        flags = hx.plane_layout(pl) | 4          # DM4D_HEX_CHANNELS_LAST?, DM4D_HEX_TIMES_01
        W0, b0 = ps[0], ps[1]
        heads = [ps[2 + 4 * k: 6 + 4 * k] for k in range(n_heads)]
        w = _lib.MlpWeights()
        w.in_dim, w.width, w.n_heads = IN, int(W0.shape[0]), n_heads
        w.W0, w.b0 = W0.data_ptr(), b0.data_ptr()
        outs = []
        for k, (W1, b1, W2, b2) in enumerate(heads):
            w.out_dim[k] = int(W2.shape[0])
            w.W1[k], w.b1[k], w.W2[k], w.b2[k] = W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr()
            outs.append(torch.empty(P, int(W2.shape[0]), dtype=torch.float32, device=dev))
        f = dict(dtype=torch.float32, device=dev)
        feat, h, y = torch.empty(P, IN, **f), torch.empty(P, 64, **f), torch.empty(n_heads, P, 64, **f)
        samples = torch.empty(L.dm4d_hexplane_scratch_bytes(S, M, B), dtype=torch.uint8, device=dev)
        scratch = torch.empty(L.dm4d_nodenet_scratch_bytes(S, M, B, n_heads), dtype=torch.uint8, device=dev)
        optr = (C.c_void_p * 4)(*[o.data_ptr() for o in outs])
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_nodenet_forward(S, M, B, plan.res_c, hx._plane_ptr_array(pl), flags, plan.aabb_c, plan.nodes.data_ptr(),
                                              t.data_ptr(), C.byref(w), feat.data_ptr(), samples.data_ptr(), h.data_ptr(), y.data_ptr(),
                                              optr, scratch.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "dm4d_nodenet_forward")
        ctx.plan, ctx.w, ctx.flags, ctx.n_heads = plan, w, flags, n_heads
        ctx.keep = (t, pl, ps, feat, samples, h, y, scratch)
        # (in-place gradient planes only when EVERY plane is a trainable leaf: a frozen plane must not receive a `.grad` an
        # optimiser would then apply; DM4D_HEX_KEEP_SPATIAL also assumes that nothing else accumulates into these buffers --
        # a plane TV / weight regulariser needs grads_in_place = False)
        ctx.plane_params = list(planes) if in_place and all(p.is_leaf and p.requires_grad for p in planes) else None
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g_outs):
        import ctypes as C

        from . import _lib, hexplane as hx

        L = _lib.lib()
        plan, n_heads = ctx.plan, ctx.n_heads
        t, pl, ps, feat, samples, h, y, scratch = ctx.keep
        dev = t.device
        B, M, S = int(t.shape[0]), plan.M, plan.S
        gs = [None if g is None else g.detach().to(torch.float32).contiguous() for g in g_outs]
        gptr = (C.c_void_p * 4)(*[None if g is None else g.data_ptr() for g in gs])
        g_feat = torch.empty_like(feat)
        grads = [torch.empty_like(p) for p in ps]
        gw = _lib.MlpWeightsGrad()
        gw.W0, gw.b0 = grads[0].data_ptr(), grads[1].data_ptr()
        for k in range(n_heads):
            gw.W1[k], gw.b1[k], gw.W2[k], gw.b2[k] = (grads[2 + 4 * k + j].data_ptr() for j in range(4))
        # persistent gradient planes installed as `.grad` (hexplane._HexPlaneFeatures.backward explains why and when)
        in_place = ctx.plane_params is not None and all(p.grad is None for p in ctx.plane_params)
        flags = ctx.flags
        if in_place:
            if plan.grad_buffers is None or any(b.shape != p.shape or b.stride() != p.stride() or b.device != p.device
                                                for b, p in zip(plan.grad_buffers, pl)):
                plan.grad_buffers = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in pl]
            pg = plan.grad_buffers
            flags |= 2        # DM4D_HEX_KEEP_SPATIAL
        else:
            pg = [torch.empty_like(p, memory_format=torch.preserve_format) for p in pl]
        sp, tp = plan.sp, plan.tp
        _p = lambda x: x.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_nodenet_backward(
                S, M, B, plan.res_c, hx._plane_ptr_array(pl), flags, plan.aabb_c, _p(plan.nodes), _p(t), C.byref(ctx.w), _p(feat), _p(samples),
                _p(h), _p(y), gptr, plan.n_sp, _p(sp["scale"]), _p(sp["plane"]), _p(sp["texel"]), _p(sp["off"]), _p(sp["item"]),
                plan.n_tp, _p(tp["scale"]), _p(tp["plane"]), _p(tp["col"]), _p(tp["off"]), _p(tp["item"]), _p(g_feat),
                hx._plane_ptr_array(pg), C.byref(gw), _p(scratch), torch.cuda.current_stream(dev).cuda_stream), "dm4d_nodenet_backward")
        ctx.keep = None
        if in_place:
            for p, gbuf in zip(ctx.plane_params, pg):
                p.grad = gbuf
            return (None,) * 5 + (None,) * len(pg) + tuple(grads)
        return (None,) * 5 + tuple(pg) + tuple(grads)


class DeformationNetwork(nn.Module):
    def __init__(self, net_width=64, bounds=1.0, resolution=(64, 64, 64, 25), multires=(1, 2, 4, 8),
                 no_ds=False, no_dr=False, no_do=True, timebase_pe=4, posebase_pe=10, scale_rotation_pe=2,
                 opacity_pe=2, timenet_width=64, timenet_output=32):
        super().__init__()
        self.timenet = nn.Sequential(nn.Linear(2 * timebase_pe + 1, timenet_width), nn.ReLU(),
                                     nn.Linear(timenet_width, timenet_output))
        self.deformation_net = _Deformation(net_width, HexPlaneField(bounds, resolution, multires), no_ds, no_dr, no_do)
        for name, n in (("time_poc", timebase_pe), ("pos_poc", posebase_pe), ("rotation_scaling_poc", scale_rotation_pe),
                        ("opacity_poc", opacity_pe)):
            self.register_buffer(name, torch.tensor([2.0 ** i for i in range(n)]))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight, gain=1)       # biases keep PyTorch's default (:557-565)
        for head in (self.deformation_net.pos_deform, self.deformation_net.scales_deform,
                     self.deformation_net.rotations_deform, self.deformation_net.opacity_deform):
            head.zero_()

    def forward_dynamic_delta(self, point, times_sel):
        """(pts [P,3], t [P,1] = 2*timestamp-1) -> (dx [P,3], dr [P,4] | None, ds [P,6] | None, do [P,1] | None)."""
        return self.deformation_net.forward_dynamic_delta(point, times_sel)

    def node_outputs(self, nodes, timestamps):
        """All B timestamps in one batched query: nodes [M,3], timestamps [B] in (0,1) ->
        dx [B,M,3], dr [B,M,4], ds [B,M,6] | None, do [B,M] | None  (dynamic_sugar.py:420-431, ts*2-1).

        On a HIP device the 24 grid_sample calls are ONE fused kernel (csrc/hexplane.hip) with an
        atomic-free gather backward; the plan (static gather lists) is built once per node set."""
        B, M = int(timestamps.shape[0]), int(nodes.shape[0])
        if nodes.is_cuda:
            from . import hexplane as hx

            self.build_plan(nodes)
            d = self.deformation_net
            # the module walk (ModuleList indexing, attribute lookups) costs ~0.1 ms per call: done once, the parameter OBJECTS
            # are stable (an optimiser updates them in place; `.data` swaps keep the object)
            cache = self.__dict__.get("_node_param_cache")
            if cache is None:
                lin0 = d.feature_out[0]
                heads = [d.pos_deform] + ([] if d.no_ds else [d.scales_deform]) + ([] if d.no_dr else [d.rotations_deform]) + \
                        ([] if d.no_do else [d.opacity_deform])
                fused_mlp = lin0.out_features == 64 and lin0.in_features % 64 == 0 and lin0.in_features <= 256
                params = [lin0.weight, lin0.bias]
                for hd in heads:
                    params += [hd.feature_out[0].main_stream.weight, hd.feature_out[0].main_stream.bias,
                               hd.feature_out[1].weight, hd.feature_out[1].bias]
                cache = self.__dict__["_node_param_cache"] = (heads, fused_mlp, params, [p for grid in d.grid.grids for p in grid])
            heads, fused_mlp, params, planes = cache
            in_place = getattr(self, "grads_in_place", False)
            if fused_mlp and getattr(self, "fuse_node_network", True) and B <= 16:
                # query + MLP as one operator (csrc/nodenet.hip), the 2 t - 1 of dynamic_sugar.py:431 inside
                outs = list(_NodeNetwork.apply(self._hex_plan, timestamps, in_place, len(heads), len(planes), *planes, *params))
                feat = None
            else:
                # 2 t - 1 in one launch: addcmul(-1, t, 2) rounds exactly like (t * 2) - 1 (2 t is exact)
                c = getattr(self, "_affine_consts", None)
                if c is None or c[0].device != timestamps.device:
                    c = self._affine_consts = (torch.tensor(-1.0, device=timestamps.device), torch.tensor(2.0, device=timestamps.device))
                feat = hx.hexplane_features(d.grid, self._hex_plan, torch.addcmul(c[0], timestamps.float(), c[1]), grads_in_place=in_place)
                if fused_mlp:
                    outs = list(_DeformMLP.apply(feat.view(B * M, -1), len(heads), *params))
            if fused_mlp:
                dx = outs.pop(0)
                ds = None if d.no_ds else outs.pop(0)
                dr = None if d.no_dr else outs.pop(0)
                do = None if d.no_do else outs.pop(0)
            else:   # other widths: the same layers as torch ops on the device
                h = d.feature_out(feat.view(B * M, -1)).float()
                dx = d.pos_deform(h)
                ds = None if d.no_ds else d.scales_deform(h)
                dr = None if d.no_dr else d.rotations_deform(h)
                do = None if d.no_do else d.opacity_deform(h)
        else:
            pts = nodes.unsqueeze(0).expand(B, M, 3).reshape(-1, 3)
            t = (timestamps.view(B, 1, 1).expand(B, M, 1).reshape(-1, 1)) * 2.0 - 1.0
            dx, dr, ds, do = self.forward_dynamic_delta(pts, t)
        r = lambda x, k: None if x is None else x.view(B, M, k)
        do = None if do is None else do.view(B, M)
        return r(dx, 3), r(dr, 4), r(ds, 6), do

    def build_plan(self, nodes):
        """The static gather lists of the fused HexPlane backward for this node set (built once; the nodes never move).
        Training loops call it in their constructor so that everything derived from the plan -- the structured-sparse
        gradient message, the sharded optimiser's state -- exists before the first step."""
        from . import hexplane as hx

        key = (nodes.data_ptr(), int(nodes.shape[0]), nodes.device)
        if getattr(self, "_hex_plan_key", None) != key:
            self._hex_plan = hx.HexPlan(self.deformation_net.grid, nodes)
            self._hex_plan_key = key
        return self._hex_plan

    def get_mlp_parameters(self):
        return [p for n, p in self.named_parameters() if "grid" not in n]

    def get_grid_parameters(self):
        return list(self.deformation_net.grid.parameters())
