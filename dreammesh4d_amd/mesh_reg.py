"""Mesh regularisers of the dynamic stage: host mirror of ``ARAPCoach``
(custom/threestudio-dreammesh4d/utils/arap_utils.py:17-224) for the way the system uses it --
``compute_arap_energy(xyz_prime, vert_rotations)`` with the skinned vertex rotations, once per key frame and per
inter-frame timestamp (system/sugar_4dgen.py:304-311,331-385) -- on one HIP launch for all timestamps
(csrc/meshreg.hip, C ABI ``dm4d_arap_energy_*``).

The static part (one-ring neighbours, the reference's cotangent weights, rest edges) is computed once on the
host with the reference's arithmetic, quirk included (dense branch of ``produce_cot_weights_nfmt``: the weight of
the directed edge (f_a, f_b) of a face is assigned 0.5 * cot(angle at f_a) / 4, then W + W^T).  The SVD branch
(rotations fitted from the deformation) is not on the dynamic stage's path and is not mirrored.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _one_ring(faces, n_verts):
    nb = [set() for _ in range(n_verts)]
    for f in np.asarray(faces):
        for j in range(3):
            nb[int(f[j])].add(int(f[(j + 1) % 3]))
            nb[int(f[j])].add(int(f[(j + 2) % 3]))
    return [sorted(s) for s in nb]


def _cot_weight_matrix(verts, faces):
    faces_t = torch.as_tensor(np.asarray(faces), dtype=torch.long)
    fv = verts[faces_t]
    v0, v1, v2 = fv[:, 0], fv[:, 1], fv[:, 2]
    A, B, Cc = (v1 - v2).norm(dim=1), (v0 - v2).norm(dim=1), (v0 - v1).norm(dim=1)
    s = 0.5 * (A + B + Cc)
    area = (s * (s - A) * (s - B) * (s - Cc)).clamp_(min=1e-12).sqrt()      # Heron (arap_utils.py:118-121)
    A2, B2, C2 = A * A, B * B, Cc * Cc
    cot = torch.stack([(B2 + C2 - A2) / area, (A2 + C2 - B2) / area, (A2 + B2 - C2) / area], dim=1) / 4.0
    i, j = faces_t[:, [0, 1, 2]].flatten(), faces_t[:, [1, 2, 0]].flatten()
    return i, j, 0.5 * cot.flatten()


class _ArapEnergy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coach, xyz, rot):
        L = _lib.lib()
        dev = coach.device
        x = xyz.detach().to(torch.float32).contiguous()
        r = rot.detach().to(torch.float32).contiguous()
        T, V = int(x.shape[0]), coach.n_verts
        ev = torch.empty(T, V, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_arap_energy_forward(T, V, coach._off.data_ptr(), coach._nbr.data_ptr(), coach._rev.data_ptr(),
                                                  coach._w.data_ptr(), coach._e.data_ptr(), x.data_ptr(), r.data_ptr(),
                                                  ev.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                       "dm4d_arap_energy_forward")
        ctx.coach = coach
        ctx.save_for_backward(x, r)
        return ev.sum(dim=1)

    @staticmethod
    def backward(ctx, g_energy):
        L = _lib.lib()
        coach = ctx.coach
        x, r = ctx.saved_tensors
        dev = coach.device
        T, V = int(x.shape[0]), coach.n_verts
        g = g_energy.detach().to(torch.float32).contiguous()
        gx = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        gr = torch.empty_like(r) if ctx.needs_input_grad[2] else None
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_arap_energy_backward(T, V, coach._off.data_ptr(), coach._nbr.data_ptr(), coach._rev.data_ptr(),
                                                   coach._w.data_ptr(), coach._e.data_ptr(), x.data_ptr(), r.data_ptr(),
                                                   g.data_ptr(), None if gx is None else gx.data_ptr(),
                                                   None if gr is None else gr.data_ptr(),
                                                   torch.cuda.current_stream(dev).cuda_stream), "dm4d_arap_energy_backward")
        return None, gx, gr


class ARAPCoach:
    """``ARAPCoach(verts, faces, device)`` of the reference, mesh (faces given) variant."""

    def __init__(self, verts, faces, device):
        self.device = torch.device(device)
        verts_c = torch.as_tensor(verts, dtype=torch.float32).detach().cpu()
        faces_n = np.asarray(torch.as_tensor(faces).cpu() if torch.is_tensor(faces) else faces)
        self.verts = verts_c.to(self.device)
        self.faces = faces_n
        self.n_verts, self.n_faces = int(verts_c.shape[0]), int(len(faces_n))
        nb = _one_ring(faces_n, self.n_verts)
        self.one_ring_neighbors = {i: n for i, n in enumerate(nb)}
        self.max_n_neighbors = max((len(n) for n in nb), default=0)
        off = np.zeros(self.n_verts + 1, np.int64)
        off[1:] = np.cumsum([len(n) for n in nb])
        nbr = np.concatenate([np.asarray(n, np.int64) for n in nb]) if len(nb) else np.zeros(0, np.int64)
        src = np.repeat(np.arange(self.n_verts), np.diff(off))
        # weights: directed assignment (later faces win), then symmetrised -- as W[i, j] = ...; W = W + W.T
        i, j, wd = _cot_weight_matrix(verts_c, faces_n)
        directed = {}
        for a, b, v in zip(i.tolist(), j.tolist(), wd.tolist()):
            directed[(a, b)] = v
        w = np.asarray([np.float32(np.float32(directed.get((a, b), 0.0)) + np.float32(directed.get((b, a), 0.0)))
                        for a, b in zip(src.tolist(), nbr.tolist())], np.float32)
        pos = {(a, b): k for k, (a, b) in enumerate(zip(src.tolist(), nbr.tolist()))}
        rev = np.asarray([pos[(b, a)] for a, b in zip(src.tolist(), nbr.tolist())], np.int64)
        e = (verts_c[src] - verts_c[nbr]).numpy()
        T = lambda a, dt: torch.as_tensor(a, dtype=dt, device=self.device).contiguous()
        self._off, self._nbr, self._rev = T(off, torch.int32), T(nbr, torch.int32), T(rev, torch.int32)
        self._w, self._e = T(w, torch.float32), T(e, torch.float32)
        self.edge_weights, self.edge_sources, self.edge_targets = w, src, nbr

    def compute_arap_energy(self, xyz_prime, vert_rotations):
        """xyz_prime [V,3] + vert_rotations [V,3,3] -> scalar (the reference's call), or batched
        [T,V,3] + [T,V,3,3] -> [T] (all timestamps of an iteration in one launch)."""
        if vert_rotations is None:
            raise NotImplementedError("the SVD branch (rotations fitted to the deformation) is not on the dynamic stage's path")
        if not xyz_prime.is_cuda:
            raise _lib.Dm4dError("ARAP energy runs on the HIP device (no CPU fallback in the product)")
        single = xyz_prime.dim() == 2
        x = xyz_prime[None] if single else xyz_prime
        r = vert_rotations[None] if single else vert_rotations
        E = _ArapEnergy.apply(self, x, r.reshape(x.shape[0], self.n_verts, 3, 3))
        return E[0] if single else E


class _NormalConsistency(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nc, xyz):
        L = _lib.lib()
        dev = nc.device
        x = xyz.detach().to(torch.float32).contiguous()
        T = int(x.shape[0])
        terms = torch.empty(T, nc.n_pairs, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_normal_consistency_forward(T, nc.n_verts, nc.n_pairs, nc._pairs.data_ptr(), x.data_ptr(),
                                                         terms.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                       "dm4d_normal_consistency_forward")
        ctx.nc = nc
        ctx.save_for_backward(x)
        return terms.sum(dim=1) / float(max(nc.n_pairs, 1))

    @staticmethod
    def backward(ctx, g_loss):
        L = _lib.lib()
        nc = ctx.nc
        (x,) = ctx.saved_tensors
        dev = nc.device
        T = int(x.shape[0])
        g = g_loss.detach().to(torch.float32).contiguous()
        gx = torch.empty_like(x)
        roles = torch.empty(T, max(nc.n_pairs, 1), 12, dtype=torch.float32, device=dev)      # the pairs' per-vertex gradient vectors
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_normal_consistency_backward_scratch(T, nc.n_verts, nc.n_pairs, nc._pairs.data_ptr(), nc._off.data_ptr(),
                                                          nc._items.data_ptr(), x.data_ptr(), g.data_ptr(), gx.data_ptr(),
                                                          roles.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                       "dm4d_normal_consistency_backward_scratch")
        return None, gx


class MeshNormalConsistency:
    """``pytorch3d.loss.mesh_normal_consistency(Meshes(verts=[T x V x 3], faces=[same F x 3] * T))`` for the T deformed
    surface meshes of an iteration (system/sugar_4dgen.py:214-226: ``get_timed_surface_mesh`` of the batch's
    timestamps, lambda_normal_consistency = 100) on one HIP launch (csrc/meshreg.hip).  The face pairs that share an
    edge are enumerated once, in pytorch3d's order; pytorch3d itself is not vendored by the reference nor installed
    here, so parity rests on its published algorithm (oracle/mesh_reg.py, closed-form cases) -- unpinned."""

    def __init__(self, faces, n_verts, device):
        self.device = torch.device(device)
        f = np.asarray(torch.as_tensor(faces).cpu() if torch.is_tensor(faces) else faces, np.int64)
        self.n_verts = int(n_verts)
        e = np.stack([f[:, [1, 2]], f[:, [2, 0]], f[:, [0, 1]]], 1).reshape(-1, 2)   # edge opposite to corner k of a face
        opp = f.reshape(-1)
        e = np.sort(e, axis=1)
        key = e[:, 0] * (int(f.max()) + 1 if f.size else 1) + e[:, 1]
        order = np.argsort(key, kind="stable")
        key_s, e_s, opp_s = key[order], e[order], opp[order]
        bounds = np.flatnonzero(np.concatenate([[True], key_s[1:] != key_s[:-1], [True]])) if len(key_s) else np.zeros(1, np.int64)
        rows = []
        for s0, s1 in zip(bounds[:-1], bounds[1:]):
            for i in range(s0, s1):
                for j in range(i + 1, s1):
                    rows.append((e_s[s0, 0], e_s[s0, 1], opp_s[i], opp_s[j]))
        pairs = np.asarray(rows, np.int64).reshape(-1, 4)
        self.n_pairs = int(len(pairs))
        # vertex -> (pair, role) items for the gather backward
        vert = pairs.reshape(-1)
        item = np.arange(vert.size, dtype=np.int64)                                    # pair * 4 + role
        order = np.argsort(vert, kind="stable")
        off = np.zeros(self.n_verts + 1, np.int64)
        np.add.at(off, vert + 1, 1)
        off = np.cumsum(off)
        T_ = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int32, device=self.device)
        self._pairs = T_(pairs if self.n_pairs else np.zeros((1, 4), np.int64))
        self._off, self._items = T_(off), T_(item[order] if vert.size else np.zeros(1, np.int64))

    def __call__(self, verts):
        """verts [T,V,3] (or [V,3]) on the HIP device -> scalar loss (mean over the meshes, as pytorch3d returns)."""
        if not verts.is_cuda:
            raise _lib.Dm4dError("mesh normal consistency runs on the HIP device (no CPU fallback in the product)")
        x = verts[None] if verts.dim() == 2 else verts
        if self.n_pairs == 0:
            return x.sum() * 0.0
        return _NormalConsistency.apply(self, x).mean()


class _LaplacianSmoothing(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ls, xyz):
        L = _lib.lib()
        dev = ls.device
        x = xyz.detach().to(torch.float32).contiguous()
        T = int(x.shape[0])
        terms = torch.empty(T, ls.n_verts, dtype=torch.float32, device=dev)
        unit = torch.empty(T, ls.n_verts, 3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_laplacian_smoothing_forward(T, ls.n_verts, ls._off.data_ptr(), ls._nbr.data_ptr(), x.data_ptr(),
                                                          terms.data_ptr(), unit.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                       "dm4d_laplacian_smoothing_forward")
        ctx.ls = ls
        ctx.save_for_backward(unit)
        return terms.sum(dim=1) / float(max(ls.n_verts, 1))

    @staticmethod
    def backward(ctx, g_loss):
        L = _lib.lib()
        ls = ctx.ls
        (unit,) = ctx.saved_tensors
        dev = ls.device
        T = int(unit.shape[0])
        g = g_loss.detach().to(torch.float32).contiguous()
        gx = torch.empty_like(unit)
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_laplacian_smoothing_backward(T, ls.n_verts, ls._off.data_ptr(), ls._nbr.data_ptr(), unit.data_ptr(),
                                                           g.data_ptr(), gx.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                       "dm4d_laplacian_smoothing_backward")
        return None, gx


class MeshLaplacianSmoothing:
    """``pytorch3d.loss.mesh_laplacian_smoothing(Meshes(...), method="uniform")`` for T meshes of one topology
    (static stage: system/sugar_static.py:246-254, lambda 1; dynamic stage: system/sugar_4dgen.py:227-230, lambda 0 as
    shipped) on csrc/meshreg.hip.  pytorch3d is not vendored / installed: parity rests on its published algorithm
    (oracle/mesh_reg.py::laplacian_smoothing, closed-form cases) -- unpinned."""

    def __init__(self, faces, n_verts, device):
        from .graph_build import mesh_edge_csr

        self.device = torch.device(device)
        self.n_verts = int(n_verts)
        f = np.asarray(torch.as_tensor(faces).cpu() if torch.is_tensor(faces) else faces, np.int64)
        off, nbr, _ = mesh_edge_csr(np.zeros((self.n_verts, 3)), f)
        T_ = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int32, device=self.device)
        self._off, self._nbr = T_(off), T_(nbr if len(nbr) else np.zeros(1, np.int64))

    def __call__(self, verts):
        if not verts.is_cuda:
            raise _lib.Dm4dError("mesh Laplacian smoothing runs on the HIP device (no CPU fallback in the product)")
        x = verts[None] if verts.dim() == 2 else verts
        return _LaplacianSmoothing.apply(self, x).mean()
