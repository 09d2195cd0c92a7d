"""One dynamic-stage training iteration on the fast path.

Host-side restatement of the loop body of `SuGaR4DGen.training_step`
(custom/threestudio-dreammesh4d/system/sugar_4dgen.py:397-429: a zero123 substep + a reference substep
per iteration), its batch (custom/threestudio-dreammesh4d/data/temporal_image.py:291-324: 4 of L frames,
one reference view + `random_camera.batch_size` random views per frame), its loss weights
(configs/sugar_dynamic_dg.yaml:135-158) and its optimiser (geometry/dynamic_sugar.py:167-235,
geometry/sugar.py:406-416: groups `deformation` 3.2e-4 / `grid` 3.2e-3 in an AdamW constructed with betas [0.9, 0.99],
eps 1e-15 -- whose two groups EFFECTIVELY run betas (0.9, 0.999) and weight_decay 0, because training_setup_dynamic's
Adam(l, lr=0, eps=1e-15) filled their dicts in place first and AdamW only fills what is missing:
distributed.REFERENCE_GEOMETRY_GROUP; `optimizer_hyper=` overrides).

What runs per iteration on each GPU:
  HexPlane+MLP at the step's 4 timestamps (1 fused launch) -> render_views for the 8 (frame, view) units
  (skinning, face->Gaussian, fused RGB+normal raster) -> SDS loss on the random views (Zero123, fp16,
  no grad through the UNet) + rgb / mask MSE on the reference views -> backward -> ONE gradient
  all-reduce -> AdamW.
Of the reference step's mesh regularisers ("next" rows) the normal consistency of the
step's deformed meshes (system/sugar_4dgen.py:214-226, lambda 100) is part of this loop when a
`mesh_reg.MeshNormalConsistency` is passed, and the two ARAP terms when a `mesh_reg.ARAPCoach` is passed: the
key-frame energy of the step's frames (ref substep, :304-311) and, every `inter_frame_reg` iterations, the energy at
`num_inter_frames` timestamps densely sampled in a random window (:331-385; their own deformation query + skinning,
no rendering), both from `milestone_arap_reg` on, with the skinned vertex rotations (:369-385).
"""
import math

import os

import torch
import torch.nn.functional as F

from . import distributed as D
from . import synthetic as syn
from .schedule import C
from .loss_sum import weighted_sum
from .views import render_views

LAMBDA = {"sds_zero123": 0.1, "rgb": 5000.0, "mask": [200, 500.0, 5000.0, 1000],      # sugar_dynamic_dg.yaml:135-158
          "normal_consistency": 100.0, "arap_reg_key_frame": 10.0, "arap_reg_inter_frame": 10.0}


# Loss terms of system/sugar_4dgen.py:181-207,236-300 whose weight is 0 in the shipped configuration (sugar_dynamic_dg.yaml:135-158): computed by
# DynamicStage._optional_terms -- the reference's torch expressions on the reference's images (renderer.compose_outputs) -- when a
# configuration gives them a weight.  (ref) = reference substep only, (both) = once per substep, each over its own views.
OPTIONAL_TERMS = ("depth", "depth_rel", "normal",                                                    # (ref), need ref_depths / ref_normals
                  "normal_smooth", "rgb_tv", "depth_tv", "normal_tv", "normal_depth_consistency",      # (both)
                  "ref_xyz", "laplacian_smoothing",                                                   # (ref)
                  "obj_centric")                                                                      # (both)
# the terms of system/sugar_4dgen.py that read comp_normal (:201-207 normal / normal_smooth, :236-275 normal_tv, normal_depth_consistency)
NORMAL_TERMS = ("normal", "normal_smooth", "normal_tv", "normal_depth_consistency", "3d_normal_smooth")


def quat_xyzw_to_matrix(q, grad_mode=None):
    from .ops import quat_xyzw_to_matrix as f

    return f(q, grad_mode)



def upload_packed(arrays, device):
    """{name: numpy array} -> {name: tensor on `device`}.  On a HIP device the arrays travel as ONE pinned, non-blocking copy (8-byte
    aligned segments of one byte buffer, viewed with their dtypes): a dozen index lists of a few bytes each were a dozen copy
    kernels in the stream (~5 us each at the head of every iteration) and, unpinned, a dozen staging copies on the host."""
    import numpy as np

    device = torch.device(device)
    if device.type != "cuda":
        return {k: torch.as_tensor(np.ascontiguousarray(v)).to(device) for k, v in arrays.items()}
    offs, total = {}, 0
    for k, v in arrays.items():
        offs[k] = total
        total += (v.nbytes + 7) // 8 * 8
    host = torch.empty(max(total, 8), dtype=torch.uint8, pin_memory=True)      # (a fresh block of the caching host allocator: it is kept until the copy has run)
    hv = host.numpy()
    for k, v in arrays.items():
        hv[offs[k]:offs[k] + v.nbytes] = np.ascontiguousarray(v).view(np.uint8).reshape(-1)
    dev = host.to(device, non_blocking=True)
    out = {}
    for k, v in arrays.items():
        out[k] = dev[offs[k]:offs[k] + v.nbytes].view(getattr(torch, str(v.dtype))).view(v.shape)
    return out


class DynamicStage:
    def __init__(self, renderer, net, nodes, static, timestamps, ref_images, ref_masks, ref_camera, guidance=None,
                 frames_per_step=4, random_views_per_frame=1, deformation_lr=0.00032, grid_lr=0.0032, seed=0,
                 normal_consistency=None, arap=None, milestone_arap_reg=100, inter_frame_reg=0, num_inter_frames=10,
                 length_inter_frames=0.1, sharded_optimizer=None, lambdas=None, optimizer_hyper=None, ref_depths=None, ref_normals=None,
                 laplacian_smoothing=None):
        self.r, self.net, self.nodes, self.static = renderer, net, nodes, static
        # loss weights: `system.loss` of the configuration (from_cfg), defaulting to the shipped sugar_dynamic_dg.yaml values
        self.lam = dict(LAMBDA)
        self.lam.update(lambdas or {})
        self.timestamps = timestamps                     # [L] in (0,1)
        self._timestamps_host = None                     # (float32 copy on the host, sample_batch)
        # [L,H,W,3], [L,H,W,1]; the reference's gt image is composited on white outside the mask
        # (data/temporal_image.py:201-202) and the dynamic step compares it UNMASKED (system/sugar_4dgen.py:164-167)
        # (float32 on the stage's device whatever the data module delivered -- float64 from numpy, uint8-derived ...: the fused
        # image head reads them through raw float32 pointers)
        ref_masks = ref_masks.to(device=nodes.device, dtype=torch.float32)
        ref_images = ref_images.to(device=nodes.device, dtype=torch.float32)
        self.ref_masks = ref_masks.contiguous()
        self.ref_images = (ref_images * ref_masks + (1.0 - ref_masks)).contiguous()
        self.ref_camera = ref_camera
        # optional inputs of optional terms: the data module's `ref_depth` [L,H,W,1] / `ref_normal` [L,H,W,3] (lambda_depth, lambda_depth_rel,
        # lambda_normal: system/sugar_4dgen.py:181-213), a mesh_reg.MeshLaplacianSmoothing (lambda_laplacian_smoothing, :227-230)
        self.ref_depths = None if ref_depths is None else ref_depths.to(device=nodes.device, dtype=torch.float32).contiguous()
        self.ref_normals = None if ref_normals is None else ref_normals.to(device=nodes.device, dtype=torch.float32).contiguous()
        self.laplacian_smoothing = laplacian_smoothing
        for k, need in (("depth", self.ref_depths), ("depth_rel", self.ref_depths), ("normal", self.ref_normals), ("laplacian_smoothing", laplacian_smoothing)):
            if self._weight_is_set(k) and need is None:
                raise ValueError(f"lambda_{k} is set but the stage was not given what the term reads "
                                 f"({'ref_depths' if 'depth' in k else 'ref_normals' if k == 'normal' else 'laplacian_smoothing'}=...)")
        self.guidance = guidance
        self.normal_consistency = normal_consistency     # mesh_reg.MeshNormalConsistency of the surface mesh, or None
        self.arap = arap                                  # mesh_reg.ARAPCoach of the surface mesh, or None
        self.milestone_arap_reg, self.inter_frame_reg = milestone_arap_reg, inter_frame_reg      # yaml:126-127 (inter_frame_reg: 0 as shipped)
        self.num_inter_frames, self.length_inter_frames = num_inter_frames, length_inter_frames  # yaml:51-52
        self.frames_per_step, self.rv = frames_per_step, random_views_per_frame
        self.dev = nodes.device
        self.gen = torch.Generator(device="cpu").manual_seed(seed + 977 * D.rank())   # per-rank seed (launch.py:166)
        hyper = dict(D.REFERENCE_GEOMETRY_GROUP)         # what the reference's two groups effectively run (module docstring)
        hyper.update(optimizer_hyper or {})
        self.opt = torch.optim.AdamW([
            {"params": net.get_mlp_parameters(), "lr": C(deformation_lr, 0, 0, interpolation="exp"), "name": "deformation", **hyper},
            {"params": net.get_grid_parameters(), "lr": C(grid_lr, 0, 0, interpolation="exp"), "name": "grid", **hyper}],
            lr=0.0, betas=(0.9, 0.99), eps=1e-15, **({"fused": True} if self.dev.type == "cuda" else {}))   # one multi-tensor launch
        self.sched = {"deformation": deformation_lr, "grid": grid_lr}
        # The exchange: the structured-sparse message (touched texels + time planes + MLP: 13.5 MB instead of 143 MB at the
        # shipped size, distributed.GradAllReducer) FROM THE FIRST STEP: the HexPlane gather plan of the (static) node set
        # is built here, not by the first query, so the reducer -- and the sharded optimiser's moments -- are never rebuilt
        # mid-run.  sharded_optimizer: reduce-scatter -> AdamW on this rank's slice of the message -> all-gather
        # (distributed.ShardedAdamW) instead of all-reduce + replicated AdamW.
        if self.dev.type == "cuda":
            plan = net.build_plan(nodes)
            self.reducer = D.GradAllReducer(net.parameters(), touched=D.touched_from_plan(net.deformation_net.grid, plan))
            net.grads_in_place = True     # persistent HexPlane gradient planes: this loop drops its gradients every step
        else:
            self.reducer = D.GradAllReducer(net.parameters())
        # (default: on for a HIP device, whatever the world size -- the message-space step is two launches over the 3.4 M elements
        # that can receive gradient (csrc/gradpack.hip: dm4d_adamw_step) against 0.2 ms of dense AdamW over 35.76 M; with world > 1:
        # pack -> reduce-scatter -> the same kernel on this rank's 1 / world slice -> all-gather -> unpack, SURVEY.md section 8e.
        # Round 4 fell back to all-reduce + torch's dense fused AdamW when world > 1.  Off on CPU tensors; DM4D_MESSAGE_ADAMW=0.)
        if sharded_optimizer is None:
            sharded_optimizer = self.dev.type == "cuda" and os.environ.get("DM4D_MESSAGE_ADAMW", "1") != "0"
        self.sharded_optimizer = bool(sharded_optimizer)
        self.sharded = D.ShardedAdamW(self.opt.param_groups, self.reducer, betas=(0.9, 0.99), eps=1e-15) if self.sharded_optimizer else None
        if self.sharded is not None and hasattr(net, "register_state_dict_pre_hook"):
            # the sharded optimiser defers the weight decay of the texels no node touches: anything that reads the parameters as a
            # whole through net.state_dict() (an exporter, a checkpoint callback) sees materialised values, not only state_for_checkpoint()
            net.register_state_dict_pre_hook(lambda *_a, **_k: self.sharded.materialize())
        self.overflow_skipped = 0        # iterations whose optimiser step was skipped on the device (reported by poll)
        self.global_step = 0
        self.fused_image_head = True     # clamp + reference-view MSEs + the SDS views' resize as one operator (image_head.py); False: the torch composition
        self.poll_every = 8              # iterations between sync-free looks at the rasterizer's capacity counters
        self.bg6 = torch.ones(6, device=self.dev)      # training background is white (diff_sugar_rasterizer_temporal.py:96-101)
        self.use_step_object = True      # node network + render_views through dm4d_step_* (step.py) where it applies
        # nothing in this loop reads the per-view Gaussian gradients: record gather + face backward as one kernel (views.ViewRenderer)
        if hasattr(renderer, "fuse_face_backward") and all(not t.requires_grad for t in (static["scales"], static["opacities"], static["rgb"])):
            renderer.fuse_face_backward = True
        # ... and nothing in it reads the normal image unless one of the normal terms has a weight: then the backward of the normal
        # pass is skipped, as autograd skips it in the reference (system/sugar_4dgen.py:201-207,236-275 only touch comp_normal under
        # lambda_normal*, lambda_normal_depth_consistency; all 0 in sugar_dynamic_dg.yaml:145-157)
        if hasattr(renderer, "rgb_gradient_only"):
            renderer.rgb_gradient_only = not any(self._weight_is_set(k) for k in NORMAL_TERMS) and os.environ.get("DM4D_RGB_GRADIENT_ONLY", "1") != "0"
        self._step_objects = {}

    def _weight_is_set(self, name):
        """A loss weight of the configuration that is not identically zero: a number, or a C() schedule normalised the way schedule.C
        reads it -- [v0, v1, end] (start 0), [start, v0, v1, end], or piecewise [s0, v0, v1, s1, v2, s2, ...] -- of which EVERY value
        entry counts (a weight that starts at 0 and is switched on later is set); a shape C() would not accept counts as set."""
        v = self.lam.get(name, 0)
        if v is None:
            return False
        if isinstance(v, (list, tuple)):
            v = list(v)
            if len(v) == 3:
                v = [0] + v
            if len(v) == 4:
                vals = v[1:3]
            elif len(v) >= 6:
                vals = [v[1], v[2]] + v[4::2]           # v0, v1, then the value of every further (value, step) pair
            else:
                return True
            try:
                return any(float(x) != 0.0 for x in vals)
            except (TypeError, ValueError):
                return True
        return float(v) != 0.0

    def sample_batch(self):
        """4 frames of L without replacement (this rank's share) + cameras: the fixed reference camera and
        random views (elev U[-10,80], azim U[-180,180], dist 3.8, fovy 20 deg; sugar_dynamic_dg.yaml:28-31)."""
        L = int(self.timestamps.shape[0])
        if D.world() > 1:
            frames = D.shard_frames(L, D.rank(), D.world(), self.frames_per_step, self.global_step)
        else:
            frames = torch.randperm(L, generator=self.gen)[:self.frames_per_step].tolist()
        cams, unit_frame, is_ref, elev, azim = [], [], [], [], []
        H, W = self.r.H, self.r.W
        for fi, _ in enumerate(frames):
            cams.append(self.ref_camera)
            unit_frame.append(fi)
            is_ref.append(True)
            elev.append(5.0)
            azim.append(0.0)
            for _ in range(self.rv):
                e = -10.0 + 90.0 * float(torch.rand(1, generator=self.gen))
                a = -180.0 + 360.0 * float(torch.rand(1, generator=self.gen))
                cams.append(syn.make_camera(H, W, elev_deg=e, azim_deg=a))
                unit_frame.append(fi)
                is_ref.append(False)
                elev.append(e)
                azim.append(a)
        # everything the step indexes with is decided here, on the host: index lists instead of boolean masks, one upload
        # per array -- `x[mask]` / `mask.any()` on device tensors cost a host synchronisation each (six per iteration)
        import numpy as np

        ref_idx = [i for i, r in enumerate(is_ref) if r]
        rnd_idx = [i for i, r in enumerate(is_ref) if not r]
        ref_pos, rnd_pos = [-1] * len(is_ref), [-1] * len(is_ref)          # the view's index among the reference / random views (image_head)
        for k, i in enumerate(ref_idx):
            ref_pos[i] = k
        for k, i in enumerate(rnd_idx):
            rnd_pos[i] = k
        fidx = [frames[u] for u in unit_frame]
        if self._timestamps_host is None or self._timestamps_host[1] is not self.timestamps:
            self._timestamps_host = (self.timestamps.detach().to("cpu", torch.float32).numpy(), self.timestamps)      # (once: a host sync)
        arrays = {"vm": np.stack([c.viewmatrix for c in cams]).astype(np.float32), "pm": np.stack([c.projmatrix for c in cams]).astype(np.float32),
                  "c2w": np.stack([c.c2w for c in cams]).astype(np.float32),          # (threestudio convention: the rays of normal_depth_consistency)
                  "frames_t": self._timestamps_host[0][frames], "unit_frame": np.asarray(unit_frame, np.int32),
                  "ref_idx": np.asarray(ref_idx, np.int64), "rnd_idx": np.asarray(rnd_idx, np.int64),
                  "ref_pos": np.asarray(ref_pos, np.int32), "rnd_pos": np.asarray(rnd_pos, np.int32),
                  "frames_t_idx": np.asarray(frames, np.int64), "fidx_ref": np.asarray([fidx[i] for i in ref_idx], np.int64),
                  "fidx_rnd": np.asarray([fidx[i] for i in rnd_idx], np.int64)}
        return {"frames": frames, "n_ref": len(ref_idx), "n_rnd": len(rnd_idx), **upload_packed(arrays, self.dev),
                "elev_rnd": torch.tensor([elev[i] for i in rnd_idx], dtype=torch.float32),        # (host tensors: see iteration())
                "azim_rnd": torch.tensor([azim[i] for i in rnd_idx], dtype=torch.float32),
                # (a host copy for the guidance: with all of its conditioning inputs on the host the conditioning graph does not wait
                #  for the caller's stream, zero123._forward_one_graph)
                "fidx_rnd_host": torch.from_numpy(arrays["fidx_rnd"].copy())}

    def inter_frame_arap(self):
        """ARAP energy at `num_inter_frames` timestamps of a random window of length `length_inter_frames`
        (training_substep_inter_frames, sugar_4dgen.py:331-385): deformation query + skinning only."""
        from . import ops

        start = float(torch.rand(1, generator=self.gen)) * (1.0 - self.length_inter_frames)
        ts = torch.linspace(start, start + self.length_inter_frames, self.num_inter_frames, device=self.dev)
        dx, dr, ds, do = self.net.node_outputs(self.nodes, ts)
        xyz, rot = [], []
        for i in range(self.num_inter_frames):
            x, q = ops.skin_vertices(self.r.graph, dx[i], dr[i], None if ds is None else ds[i], None if do is None else do[i],
                                     self.r.method_name, grad_mode=self.r.grad_mode)
            xyz.append(x)
            rot.append(q)
        return self.arap.compute_arap_energy(torch.stack(xyz), quat_xyzw_to_matrix(torch.stack(rot), self.r.grad_mode)).sum()

    def _step_object(self, B, NF):
        """The step object for this batch shape (step.DynamicStep), or None where it does not apply: CPU tensors, a learnable
        static appearance, per-frame scales, a second gradient source on the network's parameters within the iteration (the
        inter-frame ARAP term: the step object installs its persistent buffers as `.grad`, it does not accumulate)."""
        if not self.use_step_object or self.dev.type != "cuda" or self.inter_frame_reg > 0 or self._weight_is_set("ref_xyz"):
            return None            # (ref_xyz queries the network a second time, at t = 0: a second gradient source like the inter-frame ARAP term)
        key = (B, NF)
        if key not in self._step_objects:
            from .step import DynamicStep

            st = self.static
            try:
                self._step_objects[key] = DynamicStep(self.r, self.net, self.nodes, st["q_static"], st["scales"], st["opacities"], st["rgb"],
                                                      self.bg6, n_views=B, n_frames=NF)
            except ValueError as e:      # a configuration outside its domain: the two-operator host path (same kernels)
                self._step_objects[key] = None
                self.step_object_reason = str(e)
        return self._step_objects[key]

    def update_learning_rate(self, it):
        for g in self.opt.param_groups:
            g["lr"] = C(self.sched[g["name"]], 0, it, interpolation="exp")      # spatial_lr_scale = 1 (yaml:81)

    def iteration(self):
        st, it = self.static, self.global_step
        self.update_learning_rate(it)
        if self.guidance is not None:
            if hasattr(self.guidance, "cfg"):      # the cfg-constructed plugin schedules its own timestep range / grad clip
                self.guidance.update_step(0, it)
            else:                                   # yaml:115-116
                self.guidance.update_step(0, it, min_step_percent=C(0.02, 0, it), max_step_percent=C(0.5, 0, it))
        b = self.sample_batch()
        if self.guidance is not None and b["n_rnd"] and getattr(self.guidance, "host_frame_indices", False):
            # the guidance's conditioning (a function of the cameras, the frame indices and the timesteps it draws) ahead of the render
            self.guidance.prefetch(b["elev_rnd"], b["azim_rnd"], frame_indices=b["fidx_rnd_host"])
        frames_t = b["frames_t"]                 # = timestamps[frames], gathered on the host and uploaded with the cameras
        self.opt.zero_grad(set_to_none=True)
        u = b["unit_frame"]
        # views of the same frame share its skinning / face transform (reference: cached per timestamp within a step)
        step = self._step_object(int(b["vm"].shape[0]), len(b["frames"]))
        if step is not None:
            # node network + render_views as ONE C call each way on persistent buffers (step.py: the same kernels; bit-identical for the same fuse_face_backward setting)
            out = step(frames_t.to(torch.float32).contiguous(), b["vm"].contiguous(), b["pm"].contiguous(), u)
        else:
            dx, dr, ds, do = self.net.node_outputs(self.nodes, frames_t)
            out = render_views(self.r, dx, dr, ds, do, st["q_static"], st["scales"], st["opacities"], st["rgb"], b["vm"], b["pm"],
                               self.bg6, frame_index=u)
        # (the operator folds the guidance's resize to 256 x 256 in: only at the shipped 512 x 512, where that is a 2 x 2 mean)
        fused_head = self.fused_image_head and out["color"].is_cuda and self.r.H == 512 and self.r.W == 512
        if fused_head:
            # clamp, the two reference-view MSEs and the random views at half the size (what the guidance's first step, a bilinear
            # resize to 256 x 256, makes of them) as ONE operator each way (image_head.py, csrc/imagehead.hip)
            from .image_head import image_head

            mse_rgb, mse_mask, half = image_head(out["color"], out["alpha"], b["ref_pos"], b["rnd_pos"], self.ref_images, self.ref_masks,
                                                 b["fidx_ref"], b["n_ref"], b["n_rnd"] if self.guidance is not None else 0)
            rgb = mask = None
        else:
            rgb = out["color"][:, :3].clamp(0, 1).permute(0, 2, 3, 1)          # "render": clamp(0,1) (…temporal.py:229)
            mask = out["alpha"].permute(0, 2, 3, 1)
        # (index_select, not `x[idx]`: the same rows, but a gather whose backward is ONE index_add launch -- advanced indexing
        # differentiates through a sort-based index_put: six launches per use, three uses per iteration.  The loss starts as the
        # number 0: `rgb.sum() * 0.0` was a full-resolution reduction forward and a full-resolution fill backward for nothing.)
        # `loss = 0.0 + lambda_a * a + lambda_b * b + ...` is evaluated by ONE launch each way at the end (loss_sum.weighted_sum: the
        # same float32 expression left to right; the torch operators -- two per term forward, two or three backward -- for CPU tensors)
        pairs = []
        terms = {}
        if b["n_ref"]:
            if fused_head:
                terms["rgb"], terms["mask"] = mse_rgb, mse_mask
            else:
                ref = b["ref_idx"]
                terms["rgb"] = F.mse_loss(self.ref_images.index_select(0, b["fidx_ref"]), rgb.index_select(0, ref))     # unmasked: colour outside the silhouette is penalised
                terms["mask"] = F.mse_loss(mask.index_select(0, ref), self.ref_masks.index_select(0, b["fidx_ref"]))
            pairs += [(C(self.lam["rgb"], 0, it), terms["rgb"]), (C(self.lam["mask"], 0, it), terms["mask"])]
        if self.guidance is not None and b["n_rnd"]:
            # elevation / azimuth stay on the HOST (they only feed the four-number camera embedding of get_cond: a dozen
            # elementwise launches on 4-element device tensors otherwise)
            g = self.guidance(half if fused_head else rgb.index_select(0, b["rnd_idx"]), b["elev_rnd"], b["azim_rnd"],
                              torch.full_like(b["elev_rnd"], 3.8),
                              frame_indices=b["fidx_rnd_host"] if getattr(self.guidance, "host_frame_indices", False) else b["fidx_rnd"])
            terms["sds"] = g["loss_sds"]
            pairs.append((C(self.lam["sds_zero123"], 0, it), g["loss_sds"]))
        if self.normal_consistency is not None:
            # mesh_normal_consistency(get_timed_surface_mesh(batch timestamps)): the step's deformed meshes, one per frame
            terms["normal_consistency"] = self.normal_consistency(out["vxyz"])
            pairs.append((C(self.lam["normal_consistency"], 0, it), terms["normal_consistency"]))
        if self.arap is not None and it >= self.milestone_arap_reg:
            terms["arap_reg_key_frame"] = self.arap.compute_arap_energy(out["vxyz"], quat_xyzw_to_matrix(out["vrot"], self.r.grad_mode)).sum()
            pairs.append((C(self.lam["arap_reg_key_frame"], 0, it), terms["arap_reg_key_frame"]))
            if self.inter_frame_reg > 0 and it % self.inter_frame_reg == 0:
                terms["arap_reg_inter_frame"] = self.inter_frame_arap()
                pairs.append((C(self.lam["arap_reg_inter_frame"], 0, it), terms["arap_reg_inter_frame"]))
        if any(self._weight_is_set(k) for k in OPTIONAL_TERMS):
            for k, v in self._optional_terms(out, b, it).items():
                terms[k] = v if k not in terms else terms[k] + v
                pairs.append((C(self.lam[k.split("/")[0]], 0, it), v))
        loss = weighted_sum(pairs)
        if not torch.is_tensor(loss):                   # (no term applied: nothing to differentiate, but the step's contract is a backward)
            loss = out["color"].sum() * 0.0
        loss.backward()
        # A forward that overflowed its duplicate / record capacity rendered a wrong image: the optimiser step is skipped ON
        # THE DEVICE (found_inf, as a GradScaler would; no host sync), on EVERY rank (MAX over the ranks of the flag: the
        # replicas stay identical) and for BOTH optimisers; poll() notices a step or two later and enlarges the capacities.
        flag = None
        if self.dev.type == "cuda":
            flag = self.r.overflow_flag()
            if D.world() > 1:
                D.all_reduce_max(flag)
        if self.sharded_optimizer:
            for gs, go in zip(self.sharded.param_groups, self.opt.param_groups):
                gs.update({k: go[k] for k in ("lr", "betas", "eps", "weight_decay") if k in go})
            self.sharded.step(found_inf=flag)         # the exchange (reduce-scatter / all-gather) is inside
        else:
            self.reducer()                  # the one exchange step (no-op for a single process)
            if flag is not None:
                self.opt.found_inf, self.opt.grad_scale = flag, None
            self.opt.step()
        self.global_step += 1
        terms = {k: v.detach() for k, v in terms.items()}
        if self.dev.type == "cuda" and self.global_step % self.poll_every == 0:
            # poll() raises after it has ALREADY enlarged the capacities, and only on the rank that overflowed: the step was
            # skipped on the device, so the loop just carries on (all ranks on the same control path) and reports it
            from ._lib import Dm4dError

            try:
                self.r.poll()
            except Dm4dError as e:
                if not getattr(e, "overflow", False):      # anything else poll() may raise is a real error, not a skipped step
                    raise
                self.overflow_skipped += 1
                self.last_overflow_message = str(e)
                terms["overflow_skipped"] = torch.tensor(float(self.overflow_skipped))      # numeric like every other term
        return {"loss": loss.detach(), **terms}

    def _optional_terms(self, out, b, it):
        """The terms of OPTIONAL_TERMS that have a weight, as the reference writes them (system/sugar_4dgen.py:181-300), on the reference
        renderer's images of this step's views (renderer.compose_outputs).  Plain torch operators (boolean-mask gathers synchronise): these
        terms are off in the shipped configuration and are not part of the measured path.  Keys: "<term>/ref", "<term>/zero123" -- the
        reference evaluates the regularisers once per substep over that substep's views and adds the substeps' losses."""
        import torch.nn.functional as F_

        from . import renderer as R
        from .static_stage import tv_loss

        on = self._weight_is_set
        H, W = self.r.H, self.r.W
        rays_o = rays_d = None
        if on("normal_depth_consistency"):
            dirs = R.ray_directions(H, W, 0.5 * H / self.ref_camera.tanfov, device=self.dev)      # data/temporal_image.py: get_ray_directions + get_rays
            rays_o, rays_d = R.rays(dirs, b["c2w"])
        img = R.compose_outputs(out["color"], out["depth"], out["alpha"], rays_o, rays_d)
        res = {}
        groups = [("ref", b["ref_idx"]), ("zero123", b["rnd_idx"])]
        for name, idx in groups:
            if idx.numel() == 0:
                continue
            sel = {k: v.index_select(0, idx) for k, v in img.items()}
            if on("normal_smooth"):                                              # :236-246
                n = sel["comp_normal"]
                res[f"normal_smooth/{name}"] = (n[:, 1:] - n[:, :-1]).square().mean() + (n[:, :, 1:] - n[:, :, :-1]).square().mean()
            for k, key in (("rgb_tv", "comp_rgb"), ("depth_tv", "comp_depth"), ("normal_tv", "comp_normal")):    # :248-266
                if on(k):
                    res[f"{k}/{name}"] = tv_loss(sel[key].permute(0, 3, 1, 2))
            if on("normal_depth_consistency"):                                   # :268-282
                rn, rd = sel["comp_normal"] * 2 - 1, sel["comp_normal_from_dist"] * 2 - 1
                res[f"normal_depth_consistency/{name}"] = (1 - (rn.unsqueeze(-2) @ rd.unsqueeze(-1))).mean()
            if on("obj_centric"):                                                # :292-300 (the step's deformed meshes)
                vx = out["vxyz"]
                res[f"obj_centric/{name}"] = vx[..., 0].mean().abs() + vx[..., 1].mean().abs()
        if b["n_ref"]:
            sel = {k: v.index_select(0, b["ref_idx"]) for k, v in img.items()}
            gt_mask = self.ref_masks.index_select(0, b["fidx_ref"]) > 0.5                    # [n,H,W,1] bool
            if on("depth") or on("depth_rel"):
                gt = self.ref_depths.index_select(0, b["fidx_ref"])
                valid_gt, valid_pred = gt[gt_mask], sel["comp_depth"][gt_mask]
                if on("depth"):                                                  # :181-192: least-squares scale / shift of the ground truth
                    with torch.no_grad():
                        A = torch.stack([valid_gt, torch.ones_like(valid_gt)], dim=-1)
                        X = torch.linalg.lstsq(A, valid_pred.unsqueeze(1)).solution
                        fit = (A @ X)
                    res["depth/ref"] = F_.mse_loss(fit, valid_pred.unsqueeze(1))
                if on("depth_rel"):                                              # :194-200: 1 - Pearson correlation (torchmetrics.PearsonCorrCoef)
                    x, y = valid_pred - valid_pred.mean(), valid_gt - valid_gt.mean()
                    res["depth_rel/ref"] = 1 - (x * y).sum() / (x.square().sum().sqrt() * y.square().sum().sqrt())
            if on("normal"):                                                     # :202-213
                m = gt_mask.squeeze(-1)
                gt_n = 1 - 2 * self.ref_normals.index_select(0, b["fidx_ref"])[m]
                pr_n = 2 * sel["comp_normal"][m] - 1
                res["normal/ref"] = 1 - F_.cosine_similarity(pr_n, gt_n).mean()
            if on("laplacian_smoothing"):                                        # :227-230
                res["laplacian_smoothing/ref"] = self.laplacian_smoothing(out["vxyz"])
            if on("ref_xyz"):                                                    # :286-290: the mesh at t = 0 against the rest mesh
                from . import ops

                dx, dr, ds, do = self.net.node_outputs(self.nodes, torch.zeros(1, device=self.dev))
                x0, _ = ops.skin_vertices(self.r.graph, dx[0], dr[0], None if ds is None else ds[0], None if do is None else do[0],
                                          self.r.method_name, grad_mode=self.r.grad_mode)
                res["ref_xyz/ref"] = (x0 - self.r.graph.verts).abs().mean()
        return res

    def state_for_checkpoint(self):
        """Parameters as a replicated AdamW would hold them (the sharded optimiser defers the weight decay of the HexPlane
        texels no node touches: distributed.ShardedAdamW.materialize)."""
        if self.sharded is not None:
            self.sharded.materialize()
        return self.net.state_dict()

    def optimizer_state_dict(self):
        """What a checkpoint's ``optimizer_states`` entry holds for this stage: the optimiser that actually STEPS.  With the
        message-space optimiser that is ``ShardedAdamW.state_dict()`` (this rank's slice of the moments, the per-segment step counters
        and pending decay) -- ``self.opt``, the torch AdamW a host would save by default, never steps then and its state is empty."""
        from .distributed import stage_optimizer_state

        return stage_optimizer_state(self.sharded, self.opt, self.global_step, self.gen)      # (COLLECTIVE when world > 1: every rank calls it)

    def load_optimizer_state_dict(self, sd):
        """Resume: the moments, step counters (bias corrections), the iteration count and this rank's batch sampler continue where the
        checkpoint left them."""
        from .distributed import load_stage_optimizer_state

        gs = load_stage_optimizer_state(sd, self.sharded, self.opt, self.gen)
        if gs is not None:
            self.global_step = int(gs)

    @classmethod
    def from_cfg(cls, system_cfg, renderer, net, nodes, static, timestamps, ref_images, ref_masks, ref_camera, **kw):
        """The stage as `system:` of configs/sugar_dynamic_dg.yaml describes it (the block of the YAML as a dict, resolved):
        loss weights from `system.loss` (lambda_*; lists are C() schedules, system/sugar_4dgen.py:296-330), `system.freq`
        (milestone_arap_reg, inter_frame_reg), `num_inter_frames` / `length_inter_frames` (:51-52), the learning rates of
        `system.geometry` (deformation_lr, grid_lr).  Terms whose weight is 0 in the configuration (lambda_depth ...) are
        not part of this loop."""
        loss = dict(system_cfg.get("loss", {}))
        lam = {k[len("lambda_"):]: v for k, v in loss.items() if k.startswith("lambda_")}
        unknown = {k for k, v in lam.items() if k not in LAMBDA and k not in OPTIONAL_TERMS and v not in (0, 0.0, None)}
        if unknown:
            raise NotImplementedError(f"loss terms with a non-zero weight that this loop does not compute: {sorted(unknown)}")
        freq, geo = system_cfg.get("freq", {}), system_cfg.get("geometry", {})
        args = dict(lambdas={k: v for k, v in lam.items() if k in LAMBDA or (k in OPTIONAL_TERMS and v not in (0, 0.0, None))},
                    milestone_arap_reg=int(freq.get("milestone_arap_reg", 100)), inter_frame_reg=int(freq.get("inter_frame_reg", 0)),
                    num_inter_frames=int(system_cfg.get("num_inter_frames", 10)),
                    length_inter_frames=float(system_cfg.get("length_inter_frames", 0.1)),
                    deformation_lr=geo.get("deformation_lr", 0.00032), grid_lr=geo.get("grid_lr", 0.0032))
        args.update(kw)
        return cls(renderer, net, nodes, static, timestamps, ref_images, ref_masks, ref_camera, **args)
