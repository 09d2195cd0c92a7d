"""Autograd front-end of the fused HexPlane kernels (csrc/hexplane.hip, C ABI dm4d_hexplane_*).

`hexplane_features(field, nodes, times)` == `HexPlaneField.forward` of the M static graph nodes at B
timestamps (custom/threestudio-dreammesh4d/geometry/deformation.py:226-240 on the query built at
geometry/dynamic_sugar.py:420-431), in 1 launch forward and 3 launches backward instead of 24 + 24.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

PLANE_AXES = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
SPATIAL = (0, 1, 3)
TIME = (2, 4, 5)


def _p(t):
    return None if t is None else t.data_ptr()


class HexPlan:
    """Static gather lists of the backward for one node set (the nodes never move)."""

    def __init__(self, field, nodes):
        L = _lib.lib()
        dev = nodes.device
        self.S = len(field.grids)
        self.M = int(nodes.shape[0])
        res = []
        for planes in field.grids:       # plane (x,y) is [1,C,res_y,res_x]; (x,t) gives res_t
            rx, ry = planes[0].shape[3], planes[0].shape[2]
            rz, rt = planes[1].shape[2], planes[2].shape[2]
            res.append([rx, ry, rz, rt])
        self.res = np.asarray(res, np.int32)
        self.res_c = self.res.ctypes.data_as(C.c_void_p)
        self.aabb = field.aabb.detach().cpu().numpy().astype(np.float32).reshape(6).copy()
        self.aabb_c = self.aabb.ctypes.data_as(C.c_void_p)
        self.nodes = nodes.detach().to(torch.float32).contiguous()
        i0 = torch.empty(self.S, 3, self.M, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_hexplane_axis_index(self.S, self.M, self.res_c, self.aabb_c, _p(self.nodes), _p(i0),
                                                  torch.cuda.current_stream(dev).cuda_stream), "dm4d_hexplane_axis_index")
        i0 = i0.cpu().numpy().astype(np.int64)
        sp = dict(scale=[], plane=[], texel=[], off=[0], item=[])
        tp = dict(scale=[], plane=[], col=[], off=[0], item=[])
        node = np.arange(self.M, dtype=np.int64)
        for s in range(self.S):
            for p in SPATIAL:
                a0, a1 = PLANE_AXES[p]
                W, Hh = int(self.res[s][a0]), int(self.res[s][a1])
                x0, y0 = i0[s, a0], i0[s, a1]
                x1, y1 = np.minimum(x0 + 1, W - 1), np.minimum(y0 + 1, Hh - 1)
                tex = np.stack([y0 * W + x0, y0 * W + x1, y1 * W + x0, y1 * W + x1], 1).reshape(-1)   # corner = 2*row+col
                items = (node[:, None] * 4 + np.arange(4)[None, :]).reshape(-1)
                order = np.argsort(tex, kind="stable")
                tex_s, items_s = tex[order], items[order]
                uniq, start = np.unique(tex_s, return_index=True)
                sp["scale"] += [s] * len(uniq)
                sp["plane"] += [p] * len(uniq)
                sp["texel"] += uniq.tolist()
                base = len(sp["item"])
                sp["item"] += items_s.tolist()
                sp["off"] += (base + np.append(start[1:], len(tex_s))).tolist()
            for p in TIME:
                a0, _ = PLANE_AXES[p]
                W = int(self.res[s][a0])
                x0 = i0[s, a0]
                x1 = np.minimum(x0 + 1, W - 1)
                col = np.stack([x0, x1], 1).reshape(-1)
                items = (node[:, None] * 2 + np.arange(2)[None, :]).reshape(-1)
                order = np.argsort(col, kind="stable")
                col_s, items_s = col[order], items[order]
                uniq, start = np.unique(col_s, return_index=True)
                tp["scale"] += [s] * len(uniq)
                tp["plane"] += [p] * len(uniq)
                tp["col"] += uniq.tolist()
                base = len(tp["item"])
                tp["item"] += items_s.tolist()
                tp["off"] += (base + np.append(start[1:], len(col_s))).tolist()
        T = lambda a: torch.tensor(np.asarray(a, np.int32), device=dev)
        self.sp = {k: T(v) for k, v in sp.items()}
        self.tp = {k: T(v) for k, v in tp.items()}
        self.n_sp, self.n_tp = len(sp["texel"]), len(tp["col"])
        self.grad_buffers = None     # persistent dense gradient planes (hexplane_features(..., grads_in_place=True))


def plane_layout(planes):
    """0 if every plane is a contiguous [1,32,H,W] tensor, 1 if every plane is in torch.channels_last memory format
    (storage [H][W][32], what `HexPlaneField` allocates: one 128-byte line per texel); anything else is an error."""
    if all(p.dtype == torch.float32 and p.is_contiguous(memory_format=torch.channels_last) and p.shape[1] > 1 for p in planes):
        return 1
    if all(p.dtype == torch.float32 and p.is_contiguous() for p in planes):
        return 0
    raise ValueError("HexPlane planes must be float32 and all contiguous or all channels_last")


def _plane_ptr_array(planes):
    arr = (C.c_void_p * len(planes))(*[p.data_ptr() for p in planes])
    return arr


class _HexPlaneFeatures(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, times, in_place, *planes):
        L = _lib.lib()
        dev = times.device
        B = int(times.shape[0])
        pl = [p.detach() for p in planes]
        cl = plane_layout(pl)
        t = times.detach().to(torch.float32).contiguous()
        feat = torch.empty(B, plan.M, plan.S * 32, dtype=torch.float32, device=dev)
        need_bwd = any(p.requires_grad for p in planes)
        samples = torch.empty(L.dm4d_hexplane_scratch_bytes(plan.S, plan.M, B), dtype=torch.uint8, device=dev) if need_bwd else None
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_hexplane_forward(plan.S, plan.M, B, plan.res_c, _plane_ptr_array(pl), cl, plan.aabb_c,
                                               _p(plan.nodes), _p(t), _p(feat), _p(samples),
                                               torch.cuda.current_stream(dev).cuda_stream), "dm4d_hexplane_forward")
        ctx.plan, ctx.t, ctx.planes, ctx.samples, ctx.cl = plan, t, pl, samples, cl
        # grads_in_place: the Parameters themselves (leaves, not outputs: no reference cycle), see backward()
        # (only when EVERY plane is a trainable leaf: a frozen plane must never receive a `.grad`; the persistent buffers are
        # overwritten at the touched texels only, so no other gradient source may accumulate into them in this mode)
        ctx.params = list(planes) if in_place and all(p.is_leaf and p.requires_grad for p in planes) else None
        return feat

    @staticmethod
    def backward(ctx, g_feat):
        L = _lib.lib()
        plan, t, pl = ctx.plan, ctx.t, ctx.planes
        dev = t.device
        B = int(t.shape[0])
        g = g_feat.detach().to(torch.float32).contiguous()
        # In-place mode (a training loop that drops its gradients with zero_grad(set_to_none=True) every step): the dense
        # gradient planes are PERSISTENT buffers of the plan, installed as `.grad` of the plane Parameters right here.
        # The nodes are static, so the spatial planes receive gradient at the same texels every step and each of those is
        # overwritten by the gather: the 134 MB zero fill of the spatial planes disappears (the C call clears only the small
        # time planes, DM4D_HEX_KEEP_SPATIAL).  Handing the buffers to autograd instead would make AccumulateGrad CLONE them
        # (it only steals a gradient nobody else references).  Falls back to the plain path whenever a `.grad` is already
        # present (gradient accumulation over several backwards).
        in_place = ctx.params is not None and all(p.grad is None for p in ctx.params)
        flags = ctx.cl
        if in_place:
            if plan.grad_buffers is None or any(b.shape != p.shape or b.stride() != p.stride() or b.device != p.device
                                                for b, p in zip(plan.grad_buffers, pl)):
                plan.grad_buffers = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in pl]
            grads = plan.grad_buffers
            flags |= 2        # DM4D_HEX_KEEP_SPATIAL
        else:
            # dense, in the planes' own memory format; the C call zero-fills them (one launch) before the gathers
            grads = [torch.empty_like(p, memory_format=torch.preserve_format) for p in pl]
        gptr = _plane_ptr_array(grads)
        scratch, ctx.samples = ctx.samples, None      # the forward's plane samples; the backward works in place
        sp, tp = plan.sp, plan.tp
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_hexplane_backward(
                plan.S, plan.M, B, plan.res_c, _plane_ptr_array(pl), flags, plan.aabb_c, _p(plan.nodes), _p(t), _p(g),
                plan.n_sp, _p(sp["scale"]), _p(sp["plane"]), _p(sp["texel"]), _p(sp["off"]), _p(sp["item"]),
                plan.n_tp, _p(tp["scale"]), _p(tp["plane"]), _p(tp["col"]), _p(tp["off"]), _p(tp["item"]),
                _p(scratch), gptr, torch.cuda.current_stream(dev).cuda_stream), "dm4d_hexplane_backward")
        if in_place:
            for p, gbuf in zip(ctx.params, grads):
                p.grad = gbuf
            return (None, None, None) + (None,) * len(grads)
        return (None, None, None) + tuple(grads)


def hexplane_features(field, plan: HexPlan, times_pm1, grads_in_place=False):
    """field: HexPlaneField; times_pm1 [B] already in [-1, 1] -> features [B, M, 32 * n_scales].

    grads_in_place: the backward installs persistent gradient buffers as `.grad` of the planes instead of returning
    fresh tensors (see _HexPlaneFeatures.backward); for loops that call zero_grad(set_to_none=True) every step."""
    planes = [p for grid in field.grids for p in grid]
    return _HexPlaneFeatures.apply(plan, times_pm1, bool(grads_in_place), *planes)
