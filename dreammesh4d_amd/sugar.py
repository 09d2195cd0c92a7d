"""Mesh-bound Gaussian geometry of the dynamic stage: the host-side mirror of ``DynamicSuGaRModel``
(custom/threestudio-dreammesh4d/geometry/dynamic_sugar.py:42-899, registered as ``dynamic-sugar``) on top of the
static ``SuGaRModel`` state (geometry/sugar.py:33-978, ``sugar``).

Attribute and parameter names follow the reference (``_points``, ``_surface_mesh_faces``, ``_scales``,
``_quaternions``, ``all_densities``, ``_sh_coordinates_dc``, ``surface_mesh_thickness``, ``_deformation``,
``_deform_graph_node_xyz``, ``_xyz_neighbor_node_idx``, ``_xyz_neighbor_nodes_weights``), so a reference
checkpoint's geometry entries load by name.  The method surface is the one the reference's renderer and systems
use: ``get_xyz_verts``, ``get_faces``, ``get_scaling``, ``get_opacity``,
``get_points_rgb()``, ``get_timed_gs_all_single_time``, ``get_timed_gs_normals``, ``get_timed_vertex_xyz``,
``get_timed_vertex_rotation``, ``get_timed_face_normals``, ``optimizer`` / ``update_learning_rate`` ...

Everything time-dependent runs on the HIP kernels behind ``ops`` / ``views`` / ``deformation``; the static SuGaR
properties are the torch ops of ``geometry.py`` (evaluated once: the static parameters are frozen in the dynamic
stage, dynamic_sugar.py:77-87).  What is NOT mirrored: building the deformation graph from geodesic distances
(open3d / potpourri3d, :745-861; next) -- the graph (node positions, K neighbours and
weights per vertex) is an input here -- and the reference's ``discrete`` dynamic mode (per-frame tables).
"""
import math
import os

import torch
import torch.nn as nn

from . import geometry as geo
from . import ops
from .deformation import DeformationNetwork
from .schedule import C


def _thickness(module):
    """`surface_mesh_thickness` as a Python number, read from the device ONCE per value (float(parameter) is a host
    synchronisation: it stalled every evaluation of get_scaling, i.e. every training iteration)."""
    p = module.surface_mesh_thickness
    key = (p.data_ptr(), p._version)
    c = module.__dict__.get("_thickness_cache")
    if c is None or c[0] != key:
        c = module.__dict__["_thickness_cache"] = (key, float(p))
    return c[1]


class _SugarAttributes(torch.autograd.Function):
    """(points, complex numbers, log scales, densities, sh_dc) -> (means, rotations, scales, opacities, colors6 = rgb | face normal):
    ``dm4d_sugar_attributes_forward`` / ``_backward`` (csrc/sugar_attr.hip), the static SuGaR properties as one launch each way."""

    @staticmethod
    def forward(ctx, points, cx, log_scales, densities, sh_dc, faces, bary, thickness, clip):
        from . import _lib

        L = _lib.lib()
        dev = points.device
        Fn, G, V = int(faces.shape[0]), int(bary.shape[0]), int(points.shape[0])
        N = Fn * G
        f = dict(dtype=torch.float32, device=dev)
        out = (torch.empty(N, 3, **f), torch.empty(N, 4, **f), torch.empty(N, 3, **f), torch.empty(N, **f), torch.empty(N, 6, **f))
        b = bary.detach().reshape(G, 3).to(torch.float32).contiguous()
        args = tuple(t.detach() for t in (points, cx, log_scales, densities, sh_dc))
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_sugar_attributes_forward(Fn, G, V, args[0].data_ptr(), faces.data_ptr(), b.data_ptr(), args[1].data_ptr(),
                                                       args[2].data_ptr(), args[3].data_ptr(), args[4].data_ptr(), float(thickness), float(clip),
                                                       *[o.data_ptr() for o in out], torch.cuda.current_stream(dev).cuda_stream),
                       "dm4d_sugar_attributes_forward")
        ctx.save_for_backward(*args, faces, b, out[2], out[3])
        ctx.consts = (Fn, G, V, float(thickness), float(clip))
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, g_m, g_q, g_s, g_o, g_c):
        from . import _lib

        L = _lib.lib()
        points, cx, ls, den, sh, faces, b, scales, opac = ctx.saved_tensors
        Fn, G, V, thickness, clip = ctx.consts
        dev = points.device
        need = ctx.needs_input_grad
        p = lambda t: 0 if t is None else t.data_ptr()
        c = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        g_m, g_q, g_s, g_o, g_c = c(g_m), c(g_q), c(g_s), c(g_o), c(g_c)
        out = [torch.empty_like(t) if n else None for t, n in zip((points, cx, ls, den, sh), need[:5])]
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_sugar_attributes_backward(Fn, G, V, points.data_ptr(), faces.data_ptr(), b.data_ptr(), cx.data_ptr(), ls.data_ptr(),
                                                        den.data_ptr(), sh.data_ptr(), thickness, clip, scales.data_ptr(), opac.data_ptr(),
                                                        p(g_m), p(g_q), p(g_s), p(g_o), p(g_c), *[p(o) for o in out],
                                                        torch.cuda.current_stream(dev).cuda_stream), "dm4d_sugar_attributes_backward")
        return (*out, None, None, None, None)


def RGB2SH(rgb):
    return (rgb - 0.5) / 0.28209479177387814


class DynamicSuGaR(nn.Module):
    def __init__(self, verts, faces, node_xyz, nbr_idx, nbr_w, n_gaussians_per_surface_triangle=6,
                 skinning_method="hybrid", spatial_extent=3.8, vertex_colors=None, log_scales=None, complex_numbers=None,
                 densities=None, sh_dc=None, deformation_kwargs=None, deformation_lr=0.00032, grid_lr=0.0032,
                 d_scale=False, init_gs_opacity=0.5, init_gs_scales_s=1.7, learn_opacities=True, device="cuda"):
        """Defaults of the optional static state follow ``SuGaRModel.Config`` (sugar.py:35-72: init_gs_opacity 0.5,
        init_gs_scales_s 1.7; 0.9999 when the opacities are not learnt, :100-107); a dynamic-stage run normally loads the
        static stage's checkpoint over them (``weights``)."""
        super().__init__()
        dev = torch.device(device)
        T = lambda a, dt=torch.float32: torch.as_tensor(a, dtype=dt, device=dev)
        verts, faces = T(verts), T(faces, torch.long)
        G = int(n_gaussians_per_surface_triangle)
        F_, V = int(faces.shape[0]), int(verts.shape[0])
        N = F_ * G
        self.cfg_n_gaussians_per_surface_triangle = G
        self.skinning_method = skinning_method
        self.register_buffer("_surface_mesh_faces", faces)
        self.surface_mesh_thickness = nn.Parameter(T(spatial_extent / 1_000_000), requires_grad=False)
        self._points = nn.Parameter(verts, requires_grad=False)
        # static Gaussian state (frozen in the dynamic stage); defaults follow sugar.py:201-233,300-327
        if sh_dc is None:
            if vertex_colors is None:
                vertex_colors = torch.full((V, 3), 0.5, device=dev)
            bary = geo.bary_coords(G, dev)                                   # [G,3,1]
            colors = (T(vertex_colors)[faces][:, None] * bary[None]).sum(-2).reshape(-1, 3)
            sh_dc = RGB2SH(colors).unsqueeze(1)
        self._sh_coordinates_dc = nn.Parameter(T(sh_dc).reshape(N, 1, 3), requires_grad=False)
        self._sh_coordinates_rest = nn.Parameter(torch.zeros(N, 0, 3, device=dev), requires_grad=False)
        if densities is None:
            o0 = init_gs_opacity if learn_opacities else 0.9999              # sugar.py:100-107
            densities = torch.full((N, 1), math.log(o0 / (1.0 - o0)), device=dev)
        self.all_densities = nn.Parameter(T(densities).reshape(N, 1), requires_grad=False)
        if log_scales is None:
            fv = verts[faces]
            s = (fv - fv[:, [1, 2, 0]]).norm(dim=-1).min(dim=-1)[0] * geo.circle_radius(G, init_gs_scales_s)
            log_scales = torch.log(s.clamp_min(1e-7)).reshape(F_, 1, 1).expand(-1, G, 2).reshape(-1, 2)
        self._scales = nn.Parameter(T(log_scales).reshape(N, 2).clone(), requires_grad=False)
        if complex_numbers is None:
            complex_numbers = torch.zeros(N, 2, device=dev)
            complex_numbers[:, 0] = 1.0
        self._quaternions = nn.Parameter(T(complex_numbers).reshape(N, 2).clone(), requires_grad=False)
        # deformation graph (input; see module docstring)
        self._deform_graph_node_xyz = T(node_xyz)
        self._xyz_neighbor_node_idx = T(nbr_idx, torch.long)
        self._xyz_neighbor_nodes_weights = T(nbr_w)
        M = int(self._deform_graph_node_xyz.shape[0])
        self.graph = ops.DeformGraph(verts, self._xyz_neighbor_node_idx, self._xyz_neighbor_nodes_weights, M, dev)
        self.graph.verts = self._points.data           # one storage: load_state_dict copies stage-2 vertices in place, skinning sees them
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module.invalidate_static())
        self.topo = ops.MeshTopology(faces, V, G, dev)
        # deformation network: heads as in dynamic_sugar.py:141-147
        self.d_scale = bool(d_scale)
        if self.d_scale and skinning_method not in ("lbs", "hybrid"):
            raise ValueError("d_scale: true needs skinning_method lbs or hybrid (the reference defines no vertex scale for dqs, "
                             "dynamic_sugar.py:593-611)")
        kw = dict(no_dr=False, no_ds=not (d_scale or skinning_method in ("hybrid", "lbs")), no_do=skinning_method != "hybrid")
        kw.update(deformation_kwargs or {})
        self._deformation = DeformationNetwork(**kw).to(dev)
        self.training_setup_dynamic(deformation_lr, grid_lr)
        self.active_sh_degree = 0
        self._static_cache = None
        self._deformed_vert_positions = {}
        self.global_step = 0

    # ------------------------------------------------------------------ sizes / device
    @property
    def device(self):
        return self._points.device

    @property
    def n_verts(self):
        return int(self._points.shape[0])

    @property
    def n_faces(self):
        return int(self._surface_mesh_faces.shape[0])

    @property
    def n_gaussians(self):
        return self.n_faces * self.cfg_n_gaussians_per_surface_triangle

    # ------------------------------------------------------------------ static SuGaR properties (sugar.py)
    def set_static_learnable(self, flag=True):
        """`static_learnable: true` (dynamic_sugar.py:47,79-87): the static appearance tensors keep requires_grad, the
        rasterizer then runs its FULL backward (dL/dopacity, dL/d rgb, dL/dscales reduced and recorded as well).  As in the
        reference they are NOT added to the optimiser (training_setup_dynamic, :167-235): they only receive `.grad`."""
        self.static_learnable = bool(flag)
        for p in (self._scales, self.all_densities, self._sh_coordinates_dc):
            p.requires_grad_(self.static_learnable)
        self.invalidate_static()

    def _static(self):
        G = self.cfg_n_gaussians_per_surface_triangle
        if getattr(self, "static_learnable", False):       # differentiable, recomputed (three small elementwise ops)
            with torch.no_grad():
                q = geo.quaternions(self._points, self._surface_mesh_faces, self._quaternions, G)
                xyz = geo.points(self._points, self._surface_mesh_faces, geo.bary_coords(G, self.device))
            return dict(q=q, xyz=xyz, scaling=geo.scaling(self._scales, _thickness(self)),
                        opacity=geo.strengths(self.all_densities), rgb=geo.points_rgb(self._sh_coordinates_dc))
        if self._static_cache is None:
            with torch.no_grad():
                self._static_cache = dict(
                    q=geo.quaternions(self._points, self._surface_mesh_faces, self._quaternions, G),
                    scaling=geo.scaling(self._scales, _thickness(self)),
                    opacity=geo.strengths(self.all_densities),
                    rgb=geo.points_rgb(self._sh_coordinates_dc),
                    xyz=geo.points(self._points, self._surface_mesh_faces, geo.bary_coords(G, self.device)))
        return self._static_cache

    def invalidate_static(self):
        """Call after changing a static parameter (they are frozen during the dynamic stage)."""
        self._static_cache = None

    @property
    def get_xyz_verts(self):
        return self._points

    @property
    def get_faces(self):
        return self._surface_mesh_faces

    @property
    def get_xyz(self):
        return self._static()["xyz"]

    @property
    def get_scaling(self):
        return self._static()["scaling"]

    @property
    def get_rotation(self):
        return self._static()["q"]

    @property
    def static_quaternions(self):
        return self._static()["q"]

    @property
    def get_opacity(self):
        return self._static()["opacity"].reshape(-1, 1)

    @property
    def get_features(self):
        return torch.cat([self._sh_coordinates_dc, self._sh_coordinates_rest], dim=1)

    def get_points_rgb(self):
        return self._static()["rgb"]

    # ------------------------------------------------------------------ optimiser (dynamic_sugar.py:167-279, sugar.py:406-416)
    def training_setup_dynamic(self, deformation_lr=0.00032, grid_lr=0.0032):
        self._lr = {"deformation": deformation_lr, "grid": grid_lr}
        self.optimize_list = [
            {"params": self._deformation.get_mlp_parameters(), "lr": deformation_lr, "name": "deformation"},
            {"params": self._deformation.get_grid_parameters(), "lr": grid_lr, "name": "grid"}]
        self.optimize_params = [d["name"] for d in self.optimize_list]
        self.optimizer = torch.optim.Adam(self.optimize_list, lr=0.0, eps=1e-15)

    def merge_optimizer(self, net_optimizer):
        """sugar.py:406-416.  The reference appends the other modules' groups to ``optimize_list`` and builds
        ``AdamW(l, lr=0.0, betas=[0.9, 0.99], eps=1e-15)`` -- but ``training_setup``'s ``Adam(l, lr=0.0, eps=1e-15)`` (:382) has already
        filled the dicts of ``optimize_list`` IN PLACE with Adam's defaults, and AdamW only fills what is missing: the geometry groups
        effectively run betas (0.9, 0.999) and weight_decay 0, only the appended groups get (0.9, 0.99) / 0.01.  That effective
        behaviour is kept: betas / eps / weight_decay / amsgrad / maximize are carried over from the Adam-filled dicts; the
        implementation switches Adam left there (fused / foreach / capturable / differentiable: None / False, which would pin the
        slow path) are not."""
        keep = ("lr", "name", "betas", "eps", "weight_decay", "amsgrad", "maximize")
        groups = [{"params": g["params"], **{k: g[k] for k in keep if k in g}} for g in self.optimize_list] + \
            ([{"params": g["params"], "lr": g["lr"]} for g in net_optimizer.param_groups] if net_optimizer is not None else [])
        # (on a HIP device the fused implementation: one multi-tensor launch, and it honours `found_inf` -- the training loop skips
        # the step on the device when the batched renderer overflowed a capacity, static_stage.StaticStage.iteration)
        fused = {"fused": True} if all(p.is_cuda for g in groups for p in g["params"]) and len(groups) > 0 else {}
        self.optimizer = torch.optim.AdamW(groups, lr=0.0, betas=(0.9, 0.99), eps=1e-15, **fused)
        return self.optimizer

    def update_learning_rate(self, iteration):
        for g in self.optimizer.param_groups:
            if g.get("name") in self._lr:
                g["lr"] = C(self._lr[g["name"]], 0, iteration, interpolation="exp")

    def update_step(self, epoch, global_step, on_load_weights=False):
        self._deformed_vert_positions = {}
        self.global_step = global_step

    # ------------------------------------------------------------------ time-dependent quantities
    def _times(self, timestamp, frame_idx):
        if timestamp is None:
            raise NotImplementedError("frame_idx without timestamp (the reference's `discrete` mode) is not mirrored")
        return torch.as_tensor(timestamp, dtype=torch.float32, device=self.device).reshape(-1)

    def timed_node_outputs(self, timestamp, frame_idx=None):
        """Raw deformation-network outputs once per DISTINCT timestamp of the batch (the reference caches them per
        (t, f) within a step, dynamic_sugar.py:367-405) + the view -> frame map: dx [F,M,3], dr [F,M,4],
        ds [F,M,6] | None, do [F,M] | None, frame_index [B] int32."""
        t = self._times(timestamp, frame_idx)
        uniq, inv = torch.unique(t, sorted=True, return_inverse=True)
        dx, dr, ds, do = self._deformation.node_outputs(self._deform_graph_node_xyz, uniq)
        return dx, dr, ds, do, inv.to(torch.int32)

    def get_timed_vertex_attributes(self, timestamp=None, frame_idx=None):
        """{"xyz": [N_t,V,3], "rotation": [N_t,V,4] (x,y,z,w)} (dynamic_sugar.py:468-613)."""
        t = self._times(timestamp, frame_idx)
        dx, dr, ds, do = self._deformation.node_outputs(self._deform_graph_node_xyz, t)
        xyz, rot = [], []
        for i in range(int(t.shape[0])):       # convenience accessor; the training path is views.render_views
            x, r = ops.skin_vertices(self.graph, dx[i], dr[i], None if ds is None else ds[i],
                                     None if do is None else do[i], self.skinning_method)
            xyz.append(x)
            rot.append(r)
        out = {"xyz": torch.stack(xyz), "rotation": torch.stack(rot)}
        if self.d_scale:
            out["scale"] = ops.vertex_scale_matrices(self.graph, ds, do, self.skinning_method)      # [N_t,V,3,3] (:593-611)
        return out

    def timed_scales(self, ds, do):
        """Gaussian scales [n_frames, N, 3] under `d_scale` (dynamic_sugar.py:697-704) from the raw strain / opacity head
        outputs of the step's frames; None without `d_scale` (the static scaling is used)."""
        if not self.d_scale:
            return None
        return ops.gaussian_scales(self.topo, ops.vertex_scale_matrices(self.graph, ds, do, self.skinning_method), self.get_scaling)

    def get_timed_vertex_xyz(self, timestamp=None, frame_idx=None):
        return self.get_timed_vertex_attributes(timestamp, frame_idx)["xyz"]

    def get_timed_vertex_rotation(self, timestamp=None, frame_idx=None, return_matrix=False):
        q = self.get_timed_vertex_attributes(timestamp, frame_idx)["rotation"]
        return ops.quat_xyzw_to_matrix(q) if return_matrix else q        # pypose .matrix() (and its gradient convention)

    def get_timed_gs_attributes(self, timestamp=None, frame_idx=None):
        """Per timestamp: {"xyz" [N_t,N,3], "rotation" [N_t,N,4] (w,x,y,z), "normals" [N_t,N,3]} (:657-706, :330-364)."""
        va = self.get_timed_vertex_attributes(timestamp, frame_idx)
        res = [ops.face_gaussians(self.topo, x, r, self.static_quaternions) for x, r in zip(va["xyz"], va["rotation"])]
        out = {"xyz": torch.stack([m for m, _, _ in res]), "rotation": torch.stack([q for _, q, _ in res]),
               "normals": torch.stack([n for _, _, n in res]), "vertex_xyz": va["xyz"]}
        if self.d_scale:
            out["scale"] = ops.gaussian_scales(self.topo, va["scale"], self.get_scaling)
        return out

    def get_timed_gs_all_single_time(self, timestamp=None, frame_idx=None):
        """(means3D, scales, rotations, opacity, colors_precomp) of ONE timestamp (dynamic_sugar.py:708-724)."""
        t = None if timestamp is None else torch.as_tensor(timestamp).reshape(1)
        a = self.get_timed_gs_attributes(t, frame_idx)
        scales = a["scale"][0] if self.d_scale else self.get_scaling                   # (:717-720)
        return a["xyz"][0], scales, a["rotation"][0], self.get_opacity, self.get_points_rgb()

    def get_timed_gs_normals(self, timestamp=None, frame_idx=None):
        return self.get_timed_gs_attributes(timestamp, frame_idx)["normals"]

    def get_timed_face_normals(self, timestamp=None, frame_idx=None):
        G = self.cfg_n_gaussians_per_surface_triangle
        return self.get_timed_gs_normals(timestamp, frame_idx)[:, ::G]

    def get_timed_surface_mesh(self, timestamp=None, frame_idx=None):
        """(deformed vertices [N_t,V,3], faces [F,3]) -- the reference returns a pytorch3d ``Meshes`` of them."""
        return self.get_timed_vertex_xyz(timestamp, frame_idx), self._surface_mesh_faces


class SuGaR(nn.Module):
    """Static mesh-bound Gaussians: the host-side mirror of ``SuGaRModel`` bound to a surface mesh
    (custom/threestudio-dreammesh4d/geometry/sugar.py:33-978, registered as ``sugar``), the geometry of the static
    (refinement) stage -- configs/sugar_static_refine.yaml.  Same parameter names as the reference (``_points``,
    ``_surface_mesh_faces``, ``surface_mesh_thickness``, ``_scales``, ``_quaternions``, ``all_densities``,
    ``_sh_coordinates_dc`` / ``_rest``), so its checkpoints load by name (wire_formats.load_geometry).  Unlike the
    dynamic stage every property is differentiable here: they are the torch ops of geometry.py evaluated per call
    (positions, scales, opacities, colours are LEARNT in this stage, sugar.py:329-382), and they feed the HIP
    rasterizer through the drop-in operator.  The mesh-extraction / texture-baking half of SuGaRModel is out of scope."""

    def __init__(self, verts, faces, n_gaussians_per_surface_triangle=6, spatial_extent=3.8, vertex_colors=None,
                 learn_positions=True, learn_opacities=True, learn_scales=True, freeze_gaussians=False,
                 position_lr=0.00048, feature_lr=0.001, opacity_lr=0.02, scaling_lr=0.005, rotation_lr=0.001,
                 spatial_lr_scale=10.0, init_gs_opacity=0.9, init_gs_scales_s=1.3, color_clip=2.0, device="cuda"):
        """The keyword defaults are the values configs/sugar_static_refine.yaml ships (learning rates :49-58, init_gs_opacity
        0.9, init_gs_scales_s 1.3 :64-66); ``SuGaRModel.Config``'s own defaults (sugar.py:35-72: 0.5, 1.7, lr 0.001 / 0.01 /
        0.05 / 0.005 / 0.005) are what threestudio_host.SuGaRModel passes when a cfg leaves them out."""
        super().__init__()
        dev = torch.device(device)
        T = lambda a, dt=torch.float32: torch.as_tensor(a, dtype=dt, device=dev)
        verts, faces = T(verts), T(faces, torch.long)
        G = int(n_gaussians_per_surface_triangle)
        F_, V = int(faces.shape[0]), int(verts.shape[0])
        N = F_ * G
        self.cfg_n_gaussians_per_surface_triangle = G
        self.active_sh_degree = 0
        self.register_buffer("_surface_mesh_faces", faces)
        self.register_buffer("_bary", geo.bary_coords(G, dev))
        self.surface_mesh_thickness = nn.Parameter(T(spatial_extent / 1_000_000), requires_grad=False)
        self._points = nn.Parameter(verts.clone(), requires_grad=learn_positions)
        if vertex_colors is None:
            vertex_colors = torch.full((V, 3), 0.5, device=dev)
        colors = (T(vertex_colors)[faces][:, None] * self._bary[None]).sum(-2).reshape(-1, 3)    # sugar.py:209-224
        self._sh_coordinates_dc = nn.Parameter(RGB2SH(colors).unsqueeze(1), requires_grad=not freeze_gaussians)
        self._sh_coordinates_rest = nn.Parameter(torch.zeros(N, 0, 3, device=dev), requires_grad=not freeze_gaussians)
        o0 = init_gs_opacity if learn_opacities else 0.9999                                       # sugar.py:100-107
        self.all_densities = nn.Parameter(torch.full((N, 1), math.log(o0 / (1.0 - o0)), device=dev), requires_grad=learn_opacities)
        self._color_clip_cfg = color_clip
        self.color_clip = C(color_clip, 0, 0)                                                     # sugar.py:384
        fv = verts[faces]
        s0 = (fv - fv[:, [1, 2, 0]]).norm(dim=-1).min(dim=-1)[0] * geo.circle_radius(G, init_gs_scales_s)   # sugar.py:262-268
        self._scales = nn.Parameter(torch.log(s0.clamp_min(1e-7)).reshape(F_, 1, 1).expand(-1, G, 2).reshape(-1, 2).clone(),
                                    requires_grad=learn_scales)
        cx = torch.zeros(N, 2, device=dev)
        cx[:, 0] = 1.0
        self._quaternions = nn.Parameter(cx, requires_grad=learn_scales)       # (the reference ties it to learn_scales, :170)
        self._lr = dict(points=position_lr, f_dc=feature_lr, f_rest=feature_lr, all_densities=opacity_lr, scales=scaling_lr,
                        quaternions=rotation_lr)
        self.spatial_lr_scale = spatial_lr_scale
        self.training_setup()

    @property
    def device(self):
        return self._points.device

    @property
    def n_verts(self):
        return int(self._points.shape[0])

    @property
    def n_gaussians(self):
        return int(self._surface_mesh_faces.shape[0]) * self.cfg_n_gaussians_per_surface_triangle

    # ---- the property surface the renderer and the systems use (sugar.py:471-570)
    @property
    def get_xyz_verts(self):
        return self._points

    @property
    def get_faces(self):
        return self._surface_mesh_faces

    @property
    def get_xyz(self):
        return geo.points(self._points, self._surface_mesh_faces, self._bary)

    @property
    def get_scaling(self):
        return geo.scaling(self._scales, _thickness(self))

    @property
    def get_rotation(self):
        return geo.quaternions(self._points, self._surface_mesh_faces, self._quaternions, self.cfg_n_gaussians_per_surface_triangle)

    @property
    def get_opacity(self):
        return geo.strengths(self.all_densities).reshape(-1, 1)

    @property
    def get_features(self):
        """``sh_coordinates`` (sugar.py:457-462): the DC term clipped to +-color_clip."""
        return torch.cat([self._sh_coordinates_dc.clip(-self.color_clip, self.color_clip), self._sh_coordinates_rest], dim=1)

    def get_points_rgb(self):
        return geo.points_rgb(self._sh_coordinates_dc)

    def get_rendered_rgb(self):
        """The colours the reference's static renderer blends: it hands ``shs = get_features`` to the rasterizer
        (diff_sugar_rasterizer_normal.py:151-154), whose degree-0 evaluation is max(SH_C0 sh + 0.5, 0) with a zero
        gradient where clamped -- the same values and the same gradient mask, as torch ops, so that the RGB pass and the
        normal pass can share ONE 6-channel call."""
        return geo.points_rgb(self._sh_coordinates_dc.clip(-self.color_clip, self.color_clip)).clamp_min(0.0)

    @property
    def get_face_normals(self):
        return geo.face_normals(self._points, self._surface_mesh_faces)

    @property
    def get_gs_normals(self):
        return self.get_face_normals.repeat_interleave(self.cfg_n_gaussians_per_surface_triangle, dim=0)

    _ATTR_KEYS = ("xyz", "opacity", "scaling", "rotation", "rgb", "normals")

    def _attributes_fn(self, thickness):
        """(points, complex numbers, log scales, densities, sh_dc, color clip) -> the six attribute tensors: the properties' torch
        operators as a function of the parameters (the clip value as a 0-dim tensor: it follows a schedule)."""
        G, faces, bary = self.cfg_n_gaussians_per_surface_triangle, self._surface_mesh_faces, self._bary

        def fn(points, cx, log_scales, densities, sh_dc, clip):
            fv = geo.face_verts(points, faces)
            fnrm = geo.face_normals(None, None, fv=fv)
            rgb = geo.points_rgb(sh_dc.clamp(-clip, clip)).clamp_min(0.0)              # (tensor bounds: the same gradient mask as the number form)
            return (geo.points(None, None, bary, fv=fv), geo.strengths(densities).reshape(-1, 1), geo.scaling(log_scales, thickness),
                    geo.quaternions(None, faces, cx, G, fv=fv, normals=fnrm), rgb, fnrm.repeat_interleave(G, dim=0))
        return fn

    def render_attributes(self):
        """What a renderer call reads -- get_xyz, get_opacity, get_scaling, get_rotation, get_rendered_rgb(), get_gs_normals --
        evaluated together: the same values, with the face vertices gathered and the face normals computed ONCE for the three
        properties that need them (each property alone gathers its own).  (Replaying these ~100 operators and the ~200 of their
        backward from hipGraphs -- torch.cuda.make_graphed_callables -- was measured: host time per iteration 6.1 -> 4.6 ms, but the
        iteration with the Zero123 step, which is bound by the device, went from 13.9 to 14.4 ms: removed.)"""
        params = (self._points, self._quaternions, self._scales, self.all_densities, self._sh_coordinates_dc)
        if self.fused_attributes and self._points.is_cuda and self.cfg_n_gaussians_per_surface_triangle in (1, 3, 4, 6) and \
                all(p.dtype == torch.float32 and p.is_contiguous() for p in params):
            # one launch each way (csrc/sugar_attr.hip; DM4D_FUSED_ATTRIBUTES=0 / `fused_attributes = False`: the torch operators)
            m, q, sc, op, c6 = _SugarAttributes.apply(*params, self._surface_mesh_faces, self._bary, _thickness(self), float(self.color_clip))
            return dict(xyz=m, opacity=op.reshape(-1, 1), scaling=sc, rotation=q, rgb=c6[:, :3], normals=c6[:, 3:], colors6=c6)
        clip = torch.as_tensor(float(self.color_clip), device=self.device)
        return dict(zip(self._ATTR_KEYS, self._attributes_fn(_thickness(self))(*params, clip)))

    fused_attributes = os.environ.get("DM4D_FUSED_ATTRIBUTES", "1") != "0"

    # ---- optimiser (sugar.py:329-416)
    def training_setup(self):
        ls = self._lr
        cand = [("points", self._points, C(ls["points"], 0, 0) * self.spatial_lr_scale), ("f_dc", self._sh_coordinates_dc, C(ls["f_dc"], 0, 0)),
                ("f_rest", self._sh_coordinates_rest, C(ls["f_rest"], 0, 0) / 20.0), ("all_densities", self.all_densities, C(ls["all_densities"], 0, 0)),
                ("scales", self._scales, C(ls["scales"], 0, 0)), ("quaternions", self._quaternions, C(ls["quaternions"], 0, 0))]
        self.optimize_list = [{"params": [p], "lr": lr, "name": n} for n, p, lr in cand if p.requires_grad]
        self.optimize_params = [d["name"] for d in self.optimize_list]
        self.optimizer = torch.optim.Adam(self.optimize_list, lr=0.0, eps=1e-15)

    def update_learning_rate(self, iteration):
        """Only the position and feature rates follow a schedule (sugar.py:386-403)."""
        for gq in self.optimizer.param_groups:
            n = gq.get("name")
            if n == "points":
                gq["lr"] = C(self._lr["points"], 0, iteration, interpolation="exp") * self.spatial_lr_scale
            elif n == "f_dc":
                gq["lr"] = C(self._lr["f_dc"], 0, iteration, interpolation="exp")
            elif n == "f_rest":
                gq["lr"] = C(self._lr["f_rest"], 0, iteration, interpolation="exp") / 20.0
        self.color_clip = C(self._color_clip_cfg, 0, iteration)                                   # sugar.py:404

    def merge_optimizer(self, net_optimizer):
        """sugar.py:406-416 with the reference's EFFECTIVE per-group hyperparameters (see DynamicSuGaR.merge_optimizer: training_setup's
        Adam fills the dicts of optimize_list in place -- betas (0.9, 0.999), weight_decay 0 -- and AdamW only fills what is missing)."""
        keep = ("lr", "name", "betas", "eps", "weight_decay", "amsgrad", "maximize")
        groups = [{"params": g["params"], **{k: g[k] for k in keep if k in g}} for g in self.optimize_list] + \
            ([{"params": g["params"], "lr": g["lr"]} for g in net_optimizer.param_groups] if net_optimizer is not None else [])
        # (on a HIP device the fused implementation: one multi-tensor launch, and it honours `found_inf` -- the training loop skips
        # the step on the device when the batched renderer overflowed a capacity, static_stage.StaticStage.iteration)
        fused = {"fused": True} if all(p.is_cuda for g in groups for p in g["params"]) and len(groups) > 0 else {}
        self.optimizer = torch.optim.AdamW(groups, lr=0.0, betas=(0.9, 0.99), eps=1e-15, **fused)
        return self.optimizer


    def update_step(self, epoch, global_step, on_load_weights=False):
        self.global_step = global_step
