"""Autograd front-ends of the HIP skinning / face->Gaussian kernels (C ABI: include/dm4d.h).

Host-side mirror of the reference's geometry math for one timestamp:

* ``skin_vertices``   <->  DynamicSuGaRModel._get_timed_dg_attributes +
                           _get_timed_vertex_attributes_from_dg
                           (custom/threestudio-dreammesh4d/geometry/dynamic_sugar.py:408-465,487-613)
* ``face_gaussians``  <->  get_timed_gs_attributes / _get_gs_xyz_from_vertex / fuse_rotations /
                           get_timed_gs_normals (.../dynamic_sugar.py:657-676,726-743,877-889,330-364)

No CPU fallback: tensors must live on a HIP device.
"""
import numpy as np
import torch

from . import _lib

METHODS = {"lbs": 0, "dqs": 1, "hybrid": 2}
GRAD_MODES = {"exact": 0, "pypose": 0x100}      # DM4D_GRAD_PYPOSE (include/dm4d.h): the reference's pypose autograd convention
DEFAULT_GRAD_MODE = "pypose"                    # what the reference trains with (DESIGN.md "gradient convention")


def _p(t):
    return None if t is None else t.data_ptr()


def _st(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _f32(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


def _csr(keys: np.ndarray, n_keys: int):
    """items sorted by key (stable) + offsets; the static adjacency the backward gathers over."""
    keys = np.asarray(keys).reshape(-1)
    order = np.argsort(keys, kind="stable").astype(np.int32)
    off = np.zeros(n_keys + 1, np.int64)
    np.cumsum(np.bincount(keys, minlength=n_keys), out=off[1:])
    return off.astype(np.int32), order


class DeformGraph:
    """Static skinning tables: vertices, K nearest graph nodes per vertex and their weights
    (dynamic_sugar.py:745-861 builds them once, on the CPU)."""

    def __init__(self, verts, nbr_idx, nbr_w, n_nodes, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("dreammesh4d_amd.ops: HIP device required (no CPU path)")
        self.device = dev
        idx = np.asarray(nbr_idx.cpu() if torch.is_tensor(nbr_idx) else nbr_idx).astype(np.int64)
        self.V, self.K = idx.shape
        self.M = int(n_nodes)
        if idx.size and (idx.min() < 0 or idx.max() >= self.M):
            raise ValueError("neighbour index out of range")
        self.verts = torch.as_tensor(np.asarray(verts.cpu() if torch.is_tensor(verts) else verts), dtype=torch.float32).to(dev).contiguous()
        self.nbr_idx = torch.as_tensor(idx.astype(np.int32)).to(dev).contiguous()
        self.nbr_w = torch.as_tensor(np.asarray(nbr_w.cpu() if torch.is_tensor(nbr_w) else nbr_w), dtype=torch.float32).to(dev).contiguous()
        off, items = _csr(idx, self.M)
        self.csr_off = torch.as_tensor(off).to(dev)
        self.csr_items = torch.as_tensor(items).to(dev)


class MeshTopology:
    """Static faces + vertex->corner adjacency for the face->Gaussian backward."""

    def __init__(self, faces, n_verts, n_per_face, device):
        dev = torch.device(device)
        f = np.asarray(faces.cpu() if torch.is_tensor(faces) else faces).astype(np.int64)
        self.F, self.V, self.G = int(f.shape[0]), int(n_verts), int(n_per_face)
        if self.G not in (1, 3, 4, 6):
            raise ValueError("n_gaussians_per_surface_triangle must be 1, 3, 4 or 6")
        self.device = dev
        self.faces = torch.as_tensor(f.astype(np.int32)).to(dev).contiguous()
        off, items = _csr(f, self.V)
        self.csr_off = torch.as_tensor(off).to(dev)
        self.csr_items = torch.as_tensor(items).to(dev)


class _SkinVertices(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, method, dx, dr, ds, do):
        L = _lib.lib()
        g = graph
        flags, method = method & 0x100, method & 0xff
        dev = g.device
        dx_, dr_, ds_, do_ = _f32(dx), _f32(dr), _f32(ds), _f32(do)
        xyz = torch.empty(g.V, 3, dtype=torch.float32, device=dev)
        rot = torch.empty(g.V, 4, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_skin_vertices_forward(method, g.V, g.M, g.K, _p(g.verts), _p(g.nbr_idx), _p(g.nbr_w),
                                                    _p(dx_), _p(dr_), _p(ds_), _p(do_), _p(xyz), _p(rot), _st(dev)),
                       "dm4d_skin_vertices_forward")
        ctx.graph, ctx.method = g, method | flags
        ctx.save_for_backward(dx_, dr_, *( [ds_] if ds_ is not None else []), *([do_] if do_ is not None else []))
        ctx.has = (ds_ is not None, do_ is not None)
        ctx.shapes = (dx.shape, dr.shape, None if ds is None else ds.shape, None if do is None else do.shape)
        return xyz, rot

    @staticmethod
    def backward(ctx, g_xyz, g_rot):
        L = _lib.lib()
        g, method = ctx.graph, ctx.method
        dev = g.device
        saved = list(ctx.saved_tensors)
        dx_, dr_ = saved[0], saved[1]
        ds_ = saved[2] if ctx.has[0] else None
        do_ = saved[2 + int(ctx.has[0])] if ctx.has[1] else None
        f = dict(dtype=torch.float32, device=dev)
        o_dx, o_dr = torch.empty(g.M, 3, **f), torch.empty(g.M, 4, **f)
        o_ds = torch.empty(g.M, 6, **f) if ds_ is not None else None
        o_do = torch.empty(g.M, **f) if do_ is not None else None
        scratch = torch.empty(L.dm4d_skin_scratch_bytes(g.V, g.K), dtype=torch.uint8, device=dev)
        gx, gr = _f32(g_xyz), _f32(g_rot)
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_skin_vertices_backward(method, g.V, g.M, g.K, _p(g.verts), _p(g.nbr_idx), _p(g.nbr_w),
                                                     _p(dx_), _p(dr_), _p(ds_), _p(do_), _p(gx), _p(gr),
                                                     _p(g.csr_off), _p(g.csr_items), _p(scratch), _p(o_dx), _p(o_dr),
                                                     _p(o_ds), _p(o_do), _st(dev)), "dm4d_skin_vertices_backward")
        s = ctx.shapes
        return (None, None, o_dx.reshape(s[0]), o_dr.reshape(s[1]), None if o_ds is None else o_ds.reshape(s[2]),
                None if o_do is None else o_do.reshape(s[3]))


def skin_vertices(graph: DeformGraph, dx, dr, ds=None, d_opacity=None, method="hybrid", grad_mode=None):
    """Raw deformation-network outputs for one timestamp -> (vertex xyz [V,3], vertex rotation [V,4] xyzw).
    grad_mode: "pypose" (default: the reference's autograd convention) or "exact" (Euclidean gradient of the forward)."""
    m = METHODS[method]
    gm = GRAD_MODES[grad_mode or DEFAULT_GRAD_MODE]
    if m != 1 and ds is None:
        raise ValueError("lbs / hybrid skinning needs the strain head output")
    if m == 2 and d_opacity is None:
        raise ValueError("hybrid skinning needs the opacity head output")
    return _SkinVertices.apply(graph, m | gm, dx, dr, ds if m != 1 else None, d_opacity if m == 2 else None)


class _FaceGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, topo, vxyz, vrot, q_static, want_normals, flags):
        L = _lib.lib()
        t = topo
        dev = t.device
        vx, vr, qs = _f32(vxyz), _f32(vrot), _f32(q_static)
        N = t.F * t.G
        f = dict(dtype=torch.float32, device=dev)
        means, rots = torch.empty(N, 3, **f), torch.empty(N, 4, **f)
        normals = torch.empty(N, 3, **f) if want_normals else None
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_face_gaussians_forward(t.F, t.G, _p(t.faces), _p(vx), _p(vr), _p(qs), _p(means), _p(rots),
                                                     _p(normals), _st(dev)), "dm4d_face_gaussians_forward")
        ctx.topo, ctx.flags = t, flags
        ctx.save_for_backward(vx, vr, qs)
        if normals is None:
            normals = torch.empty(0, **f)
        return means, rots, normals

    @staticmethod
    def backward(ctx, g_means, g_rots, g_normals):
        L = _lib.lib()
        t = ctx.topo
        dev = t.device
        vx, vr, qs = ctx.saved_tensors
        f = dict(dtype=torch.float32, device=dev)
        o_x, o_r = torch.empty(t.V, 3, **f), torch.empty(t.V, 4, **f)
        scratch = torch.empty(L.dm4d_face_scratch_bytes(t.F), dtype=torch.uint8, device=dev)
        gm, gr = _f32(g_means), _f32(g_rots)
        gn = _f32(g_normals) if g_normals is not None and g_normals.numel() else None
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_face_gaussians_backward(t.F, t.G | ctx.flags, t.V, _p(t.faces), _p(vx), _p(vr), _p(qs), _p(gm), _p(gr),
                                                      _p(gn), _p(t.csr_off), _p(t.csr_items), _p(scratch), _p(o_x),
                                                      _p(o_r), _st(dev)), "dm4d_face_gaussians_backward")
        return None, o_x, o_r, None, None, None


def face_gaussians(topo: MeshTopology, vxyz, vrot, q_static_wxyz, want_normals=True, grad_mode=None):
    """Deformed vertices -> (means [N,3], rotations [N,4] wxyz, normals [N,3] or empty)."""
    return _FaceGaussians.apply(topo, vxyz, vrot, q_static_wxyz, bool(want_normals), GRAD_MODES[grad_mode or DEFAULT_GRAD_MODE])


class _MatrixPypose(torch.autograd.Function):
    """``SO3.matrix()`` of a unit quaternion with pypose's backward (SO3_Act on the basis vectors): the gradient with
    respect to the quaternion storage is (sum_i R e_i x G[:, i], 0)."""

    @staticmethod
    def forward(ctx, q):
        if q.is_cuda:          # one launch (csrc/meshreg.hip) instead of ~45 elementwise ones; the same arithmetic
            qc = _f32(q)
            R = torch.empty(qc.shape[:-1] + (3, 3), dtype=torch.float32, device=q.device)
            with torch.cuda.device(q.device):
                _lib.check(_lib.lib().dm4d_quat_to_matrix_forward(qc.numel() // 4, _p(qc), _p(R), _st(q.device)), "dm4d_quat_to_matrix_forward")
        else:
            x, y, z, w = q.unbind(-1)
            R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                             2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                             2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(*q.shape[:-1], 3, 3)
        ctx.save_for_backward(R)
        return R

    @staticmethod
    def backward(ctx, G):
        (R,) = ctx.saved_tensors
        if R.is_cuda:
            Gc = _f32(G)
            gq = torch.empty(R.shape[:-2] + (4,), dtype=torch.float32, device=R.device)
            with torch.cuda.device(R.device):
                _lib.check(_lib.lib().dm4d_quat_to_matrix_backward_pypose(R.numel() // 9, _p(R), _p(Gc), _p(gq), _st(R.device)),
                           "dm4d_quat_to_matrix_backward_pypose")
            return gq
        t = torch.linalg.cross(R.transpose(-1, -2), G.transpose(-1, -2), dim=-1).sum(dim=-2)     # sum over the columns i
        return torch.cat([t, torch.zeros_like(t[..., :1])], dim=-1)


def quat_xyzw_to_matrix(q, grad_mode=None):
    """[..., 4] (x, y, z, w) unit quaternions -> [..., 3, 3] (``get_timed_vertex_rotation(return_matrix=True)``,
    dynamic_sugar.py:640-655: a pypose ``.matrix()``), in the chosen gradient convention."""
    if (grad_mode or DEFAULT_GRAD_MODE) == "pypose":
        return _MatrixPypose.apply(q)
    x, y, z, w = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(*q.shape[:-1], 3, 3)


# ----------------------------------------------------------------------------- d_scale branch (dynamic_sugar.py:592-612,682-704)
# `d_scale: true` also stretches the Gaussians by the blended strain of the graph nodes.  The shipped configuration sets it
# to false, so this branch is not on the measured path: the two small blends below are plain torch operators on device
# tensors (autograd provides their backward); the per-frame scales they produce go through the HIP rasterizer like any
# other input (views.render_views accepts [n_frames, N, 3] scales and returns their gradient).
def strain_to_matrix(ds):
    """I + sym(ds) [...,3,3] from the strain head's 6-vector (diag = ds[0:3], (01) = ds[3], (02) = ds[4], (12) = ds[5];
    strain_tensor_to_matrix, dynamic_sugar.py:29-39)."""
    one = torch.ones_like(ds[..., 0])
    return torch.stack([one + ds[..., 0], ds[..., 3], ds[..., 4], ds[..., 3], one + ds[..., 1], ds[..., 5],
                        ds[..., 4], ds[..., 5], one + ds[..., 2]], dim=-1).reshape(ds.shape[:-1] + (3, 3))


class _VertexScales(torch.autograd.Function):
    """csrc/dscale.hip: dm4d_vertex_scales_forward / _backward (atomic-free gather over the node -> (vertex, k) adjacency)."""

    @staticmethod
    def forward(ctx, graph, m, ds, d_opacity):
        L, g, dev = _lib.lib(), graph, graph.device
        ds_, do_ = _f32(ds), (None if d_opacity is None else _f32(d_opacity))
        NF = int(ds_.shape[0])
        out = torch.empty(NF, g.V, 3, 3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_vertex_scales_forward(m, NF, g.V, g.M, g.K, _p(g.nbr_idx), _p(g.nbr_w), _p(ds_), _p(do_), _p(out), _st(dev)),
                       "dm4d_vertex_scales_forward")
        ctx.graph, ctx.m = g, m
        ctx.save_for_backward(ds_, do_)
        return out

    @staticmethod
    def backward(ctx, g_out):
        L, g, dev = _lib.lib(), ctx.graph, ctx.graph.device
        ds_, do_ = ctx.saved_tensors
        NF = int(ds_.shape[0])
        go = _f32(g_out)
        o_ds = torch.empty_like(ds_)
        o_do = torch.empty_like(do_) if do_ is not None else None
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_vertex_scales_backward(ctx.m, NF, g.V, g.M, g.K, _p(g.nbr_idx), _p(g.nbr_w), _p(ds_), _p(do_), _p(g.csr_off),
                                                     _p(g.csr_items), _p(go), _p(o_ds), _p(o_do), _st(dev)), "dm4d_vertex_scales_backward")
        return None, None, o_ds, o_do


def vertex_scale_matrices(graph: DeformGraph, ds, d_opacity=None, method="hybrid"):
    """Per-vertex scale matrices [n_frames, V, 3, 3] of the `d_scale` branch (dynamic_sugar.py:593-611): lbs: the weighted
    sum of the neighbour nodes' strain matrices; hybrid: weighted by the nodes' opacities as well, plus (1 - lbs weight) I
    with the lbs weight clamp(sum_k w_k o_k + 0.4, max = 1) of the position blend (:572-578).  ds [n_frames, M, 6] raw strain
    head outputs, d_opacity [n_frames, M] raw opacity logits.  HIP kernels (csrc/dscale.hip) on the graph's device."""
    if method not in ("lbs", "hybrid"):
        raise ValueError("d_scale needs skinning_method lbs or hybrid (the reference defines no vertex scale for dqs)")
    if method == "hybrid" and d_opacity is None:
        raise ValueError("hybrid skinning needs the opacity head output")
    if not ds.is_cuda:
        raise _lib.Dm4dError("vertex_scale_matrices: the operator runs on the HIP device (no CPU fallback in the product)")
    if tuple(ds.shape[1:]) != (graph.M, 6) or (method == "hybrid" and tuple(d_opacity.shape) != (ds.shape[0], graph.M)):
        raise ValueError(f"vertex_scale_matrices: ds must be [n_frames, {graph.M}, 6] and d_opacity [n_frames, {graph.M}]")
    return _VertexScales.apply(graph, METHODS[method], ds, d_opacity if method == "hybrid" else None)


class _GaussianScales(torch.autograd.Function):
    """csrc/dscale.hip: dm4d_gaussian_scales_forward / _backward (gather over the vertex -> (face, corner) adjacency)."""

    @staticmethod
    def forward(ctx, topo, bary, vertex_scales, scaling):
        L, t, dev = _lib.lib(), topo, topo.device
        sv, sc = _f32(vertex_scales), _f32(scaling).reshape(t.F * t.G, 3)
        NF = int(sv.shape[0])
        out = torch.empty(NF, t.F * t.G, 3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_gaussian_scales_forward(NF, t.F, t.G, t.V, _p(t.faces), _p(bary), _p(sv), _p(sc), _p(out), _st(dev)),
                       "dm4d_gaussian_scales_forward")
        ctx.topo, ctx.bary, ctx.sc_shape = t, bary, scaling.shape
        ctx.save_for_backward(sv, sc)
        return out

    @staticmethod
    def backward(ctx, g_out):
        L, t, dev = _lib.lib(), ctx.topo, ctx.topo.device
        sv, sc = ctx.saved_tensors
        NF = int(sv.shape[0])
        go = _f32(g_out)
        o_sv = torch.empty_like(sv) if ctx.needs_input_grad[2] else None
        o_sc = torch.empty_like(sc) if ctx.needs_input_grad[3] else None
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_gaussian_scales_backward(NF, t.F, t.G, t.V, _p(t.faces), _p(ctx.bary), _p(sv), _p(sc), _p(t.csr_off), _p(t.csr_items),
                                                       _p(go), _p(o_sv), _p(o_sc), _st(dev)), "dm4d_gaussian_scales_backward")
        return None, None, o_sv, None if o_sc is None else o_sc.reshape(ctx.sc_shape)


def gaussian_scales(topo: MeshTopology, vertex_scales, scaling):
    """Scales [n_frames, N, 3] of the bound Gaussians under `d_scale` (dynamic_sugar.py:697-704): the barycentric blend of
    the three corner vertices' scale matrices applied to the static scaling vector (thickness, s1, s2).  HIP kernels
    (csrc/dscale.hip)."""
    from . import geometry as geo

    if not vertex_scales.is_cuda:
        raise _lib.Dm4dError("gaussian_scales: the operator runs on the HIP device (no CPU fallback in the product)")
    if tuple(vertex_scales.shape[1:]) != (topo.V, 3, 3) or scaling.numel() != topo.F * topo.G * 3:
        raise ValueError(f"gaussian_scales: vertex_scales must be [n_frames, {topo.V}, 3, 3] and scaling [{topo.F * topo.G}, 3]")
    bary = topo.__dict__.get("_bary_dev")
    if bary is None:
        bary = topo.__dict__["_bary_dev"] = geo.bary_coords(topo.G, topo.device, torch.float32)[..., 0].contiguous()       # [G, 3]
    return _GaussianScales.apply(topo, bary, vertex_scales, scaling)
