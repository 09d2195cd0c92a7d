"""One static-stage (SuGaR refinement) training iteration: host-side restatement of ``SuGaRStatic.training_step``
(custom/threestudio-dreammesh4d/system/sugar_static.py:110-340, stage "sugar") with the loss weights of
configs/sugar_static_refine.yaml:105-133 and the optimiser of geometry/sugar.py:329-416 (an AdamW constructed with betas
[0.9, 0.99], eps 1e-15 whose groups points / f_dc / f_rest / all_densities / scales / quaternions effectively run betas
(0.9, 0.999) without weight decay: sugar.SuGaR.merge_optimizer).

Per iteration: a reference substep -- the reference view, rgb and mask MSE against the input image (:151-160) -- and a
random substep -- `random_camera.batch_size` views (elev U[-10,80], azim U[-180,180], dist 3.8, fovy 20 deg; yaml:21-28):
Zero123 SDS (:197-205), the mesh regularisers of the surface mesh (normal consistency 10, Laplacian smoothing 1,
:246-254) and the total-variation terms of rgb / depth / normal (:274-290, threestudio/utils/loss.py:8-16).  Every view is
one fused RGB + normal call of the HIP rasterizer through ``renderer.DiffSuGaRNormal``; the geometry's properties are
torch ops (they are learnt here), the regularisers run on csrc/meshreg.hip.  Data-parallel training (static stage:
~1.7 MB of gradients) uses the same ``distributed.GradAllReducer``.
"""
import os

import torch
import torch.nn.functional as F

from . import distributed as D
from . import synthetic as syn
from .loss_sum import weighted_sum
from .schedule import C

# sugar_static_refine.yaml:105-133 (terms with lambda 0 and the stage-"gaussian" SuGaR regularisers, which start at
# step 3000 of a 2000-step schedule, are not part of the loop)
LAMBDA = {"sds": 0.01, "rgb": 1000.0, "mask": 100.0, "normal_consistency": 10.0, "laplacian_smoothing": 1.0, "rgb_tv": 1.0,
          "normal_tv": 1.0, "depth_tv": 1.0}


def tv_loss(x):
    """threestudio/utils/loss.py:8-16 on [B,C,H,W]."""
    b, c, h, w = x.shape
    h_tv = (x[:, :, 1:, :] - x[:, :, :h - 1, :]).pow(2).sum()
    w_tv = (x[:, :, :, 1:] - x[:, :, :, :w - 1]).pow(2).sum()
    return 2 * (h_tv / (c * (h - 1) * w) + w_tv / (c * h * (w - 1))) / b


class StaticStage:
    def __init__(self, geometry, renderer, ref_image, ref_mask, H, W, guidance=None, random_views=4, normal_consistency=None,
                 laplacian_smoothing=None, seed=0, lambdas=None, message_adamw=None):
        self.g, self.r = geometry, renderer
        self.lam = dict(LAMBDA)            # `system.loss` of the configuration (from_cfg); defaults: sugar_static_refine.yaml
        self.lam.update(lambdas or {})
        self.ref_image, self.ref_mask = ref_image, ref_mask            # [1,H,W,3], [1,H,W,1]
        self.H, self.W = H, W
        self.guidance = guidance
        self.rv = random_views
        self.nc, self.lap = normal_consistency, laplacian_smoothing    # mesh_reg.MeshNormalConsistency / MeshLaplacianSmoothing
        self.dev = geometry.device
        self.gen = torch.Generator(device="cpu").manual_seed(seed + 977 * D.rank())
        self.opt = geometry.merge_optimizer(None)
        self.reducer = D.GradAllReducer([p for p in geometry.parameters() if p.requires_grad and p.numel()])
        # One process on a HIP device: the AdamW step as the dynamic stage's fused kernel over the reducer's (here dense) message --
        # distributed.ShardedAdamW._step_fused, two launches -- instead of torch's multi-tensor kernel, which runs one launch per
        # parameter group with a handful of workgroups each: 5 x 35 us for these six small tensors, 11.21 -> 11.05 ms per iteration
        # (same per-element arithmetic, tests/test_adamw_gpu.py; message_adamw=False / DM4D_MESSAGE_ADAMW=0: torch's)
        if message_adamw is None:
            message_adamw = self.dev.type == "cuda" and D.world() == 1 and os.environ.get("DM4D_MESSAGE_ADAMW", "1") != "0"
        # (betas / eps / weight_decay PER GROUP, from the optimiser's own groups: geometry.merge_optimizer keeps the reference's effective
        #  mix -- geometry groups (0.9, 0.999) / no decay, appended groups (0.9, 0.99) / 0.01)
        self.sharded = D.ShardedAdamW(self.opt.param_groups, self.reducer, betas=(0.9, 0.99), eps=1e-15) if message_adamw else None
        self.ref_cam = syn.make_camera(H, W, elev_deg=5.0, azim_deg=0.0)                # yaml:11-14
        self.global_step = 0
        self.poll_every, self.overflow_skipped = 8, 0
        self.fused_head = os.environ.get("DM4D_STATIC_FUSED_HEAD", "1") != "0"      # the image-space terms as one operator (static_head.py)
        self._head = None

    def _head_positions(self):
        """The operator's view tables for "one reference view, then `rv` random views" and the references as float32 [1,H,W,.]."""
        if self._head is None:
            dev = self.dev
            B = 1 + self.rv
            self._head = dict(ref_pos=torch.tensor([0] + [-1] * self.rv, dtype=torch.int32, device=dev),
                              rnd_pos=torch.tensor([-1] + list(range(self.rv)), dtype=torch.int32, device=dev),
                              fidx_ref=torch.zeros(1, dtype=torch.int64, device=dev))
            self._ref_image_f = self.ref_image.to(dev, torch.float32).reshape(1, self.H, self.W, 3).contiguous()
            self._ref_mask_f = self.ref_mask.to(dev, torch.float32).reshape(1, self.H, self.W, 1).contiguous()
            assert B == len(self._head["ref_pos"])
        return self._head

    def _batch(self, cams):
        c2w = torch.stack([torch.tensor(c.c2w, dtype=torch.float32) for c in cams])
        return {"c2w": c2w, "fovy": torch.tensor([c.fovy for c in cams], dtype=torch.float32), "height": self.H, "width": self.W}

    def iteration(self):
        g, it = self.g, self.global_step
        g.update_learning_rate(it)
        self.opt.zero_grad(set_to_none=True)
        terms = {}
        # ---- the reference view and the random views as ONE batch of the renderer (the reference renders the two substeps one
        #      after the other from the same parameters and adds their losses, sugar_static.py:300-340: the same images and the same
        #      sum; one operator call each way instead of two)
        u = torch.rand(self.rv, 2, generator=self.gen)
        elev, azim = -10.0 + 90.0 * u[:, 0], -180.0 + 360.0 * u[:, 1]
        cams = [syn.make_camera(self.H, self.W, elev_deg=float(e), azim_deg=float(a)) for e, a in zip(elev, azim)]
        batch = self._batch([self.ref_cam] + cams)
        if self.guidance is not None and hasattr(self.guidance, "prefetch"):
            self.guidance.update_step(0, it)
            self.guidance.prefetch(elev, azim)          # the conditioning of this iteration's guidance call, ahead of the render
        raw = None
        if self.fused_head and self.dev.type == "cuda" and self.H % 2 == 0 and self.W % 2 == 0 and hasattr(self.r, "render_batch_raw"):
            raw = self.r.render_batch_raw(batch)
        if raw is not None:
            # the image-space terms as ONE operator each way (static_head.py, csrc/statichead.hip): clamp, the normal map and the
            # masked depth, the two masked MSEs of the reference view, the three total-variation terms and the random views at half
            # the size (what the guidance's first step, a bilinear resize to 256 x 256, makes of 512 x 512 views)
            from .static_head import static_head

            hp = self._head_positions()
            t5, half = static_head(raw["color"], raw["depth"], raw["alpha"], hp["ref_pos"], hp["rnd_pos"], self._ref_image_f, self._ref_mask_f,
                                   hp["fidx_ref"], 1, self.rv)
            # (the five terms enter the loss as ONE vector, loss_sum.weighted_sum: unbinding them costs a stack of five gradients backward)
            terms["rgb"], terms["mask"], tv_rgb, tv_depth, tv_normal = t5.detach().unbind(0)
            out = {"half": half, "tv": {"rgb_tv": tv_rgb, "depth_tv": tv_depth, "normal_tv": tv_normal}}
        else:
            both = self.r.batch_forward(batch)
            # ---- reference substep
            m = self.ref_mask.float()
            terms["rgb"] = F.mse_loss(self.ref_image * m, both["comp_rgb"][:1] * m)
            terms["mask"] = F.mse_loss(m, both["comp_mask"][:1])
            # ---- random substep
            out = {k: v[1:] for k, v in both.items() if torch.is_tensor(v)}
        # `loss = lambda_rgb * rgb + lambda_mask * mask + ...` as ONE launch each way (loss_sum.weighted_sum), evaluated left to right
        if raw is not None:
            pairs = [(tuple(C(self.lam[k], 0, it) for k in ("rgb", "mask", "rgb_tv", "depth_tv", "normal_tv")), t5)]
        else:
            pairs = [(C(self.lam["rgb"], 0, it), terms["rgb"]), (C(self.lam["mask"], 0, it), terms["mask"])]
        if self.guidance is not None:
            self.guidance.update_step(0, it)
            # (elevation / azimuth stay on the host: they only feed the four-number camera embedding, as in DynamicStage)
            if raw is None:
                views = out["comp_rgb"]
            elif self.H == 512 and self.W == 512:
                views = out["half"]                                                  # the guidance's own resize would produce exactly these
            else:
                views = raw["color"][1:, :3].clamp(0, 1).permute(0, 2, 3, 1)         # any other size: the guidance resizes
            go = self.guidance(views, elev, azim, torch.full((self.rv,), 3.8))
            terms["sds"] = go["loss_sds"]
            pairs.append((C(self.lam["sds"], 0, it), terms["sds"]))
        if self.nc is not None:
            terms["normal_consistency"] = self.nc(g.get_xyz_verts)
            pairs.append((C(self.lam["normal_consistency"], 0, it), terms["normal_consistency"]))
        if self.lap is not None:
            terms["laplacian_smoothing"] = self.lap(g.get_xyz_verts)
            pairs.append((C(self.lam["laplacian_smoothing"], 0, it), terms["laplacian_smoothing"]))
        for k, key in (("rgb_tv", "comp_rgb"), ("depth_tv", "comp_depth"), ("normal_tv", "comp_normal")):
            terms[k] = out["tv"][k] if raw is not None else tv_loss(out[key].permute(0, 3, 1, 2))
            if raw is None:
                pairs.append((C(self.lam[k], 0, it), terms[k]))
        loss = weighted_sum(pairs)
        loss.backward()
        self.reducer()
        # the batched renderer sizes its duplicate / record lists without a host synchronisation: a forward that overflowed them
        # rendered a wrong image, and the optimiser step is skipped ON THE DEVICE (found_inf, as in DynamicStage); poll() notices a
        # step or two later and enlarges the capacities
        vr = getattr(self.r, "views_renderer", None)
        if vr is not None and vr.last is not None:
            flag = vr.overflow_flag()
            if D.world() > 1:
                D.all_reduce_max(flag)
            self.opt.found_inf, self.opt.grad_scale = flag, None
        if self.sharded is not None:
            for gs, go in zip(self.sharded.param_groups, self.opt.param_groups):
                gs.update({k: go[k] for k in ("lr", "betas", "eps", "weight_decay") if k in go})
            self.sharded.step(found_inf=flag if (vr is not None and vr.last is not None) else None)
        else:
            self.opt.step()
        self.global_step += 1
        terms = {k: v.detach() for k, v in terms.items()}
        if vr is not None and self.global_step % self.poll_every == 0:
            from ._lib import Dm4dError

            try:
                vr.poll()
            except Dm4dError as e:
                if not getattr(e, "overflow", False):
                    raise
                self.overflow_skipped += 1
                terms["overflow_skipped"] = torch.tensor(float(self.overflow_skipped))
        return {"loss": loss.detach(), **terms}

    def state_for_checkpoint(self):
        """The geometry's parameters as a replicated optimiser would hold them (any deferred weight decay applied)."""
        if self.sharded is not None:
            self.sharded.materialize()
        return self.g.state_dict()

    def optimizer_state_dict(self):
        """The state of the optimiser that actually steps (DynamicStage.optimizer_state_dict: with the message-space optimiser
        ``self.opt`` never steps and a host saving ITS state would save nothing)."""
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
    def from_cfg(cls, system_cfg, geometry, renderer, ref_image, ref_mask, H, W, **kw):
        """The stage as `system:` of configs/sugar_static_refine.yaml describes it: loss weights from `system.loss`
        (lambda_*, C() schedules allowed); terms this loop does not compute must have weight 0 (or start after the run:
        the stage-"gaussian" SuGaR regularisers begin at freq.start_sugar_reg = 3000 of a 2000-step schedule)."""
        loss = dict(system_cfg.get("loss", {}))
        lam = {k[len("lambda_"):]: v for k, v in loss.items() if k.startswith("lambda_")}
        later = ("opacity_max", "opacity_binary", "sugar_density_reg", "sugar_sdf_normal_reg")          # SuGaR terms gated by freq.start_sugar_reg
        unknown = {k for k, v in lam.items() if k not in LAMBDA and k not in later and v not in (0, 0.0, None)}
        if unknown:
            raise NotImplementedError(f"loss terms with a non-zero weight that this loop does not compute: {sorted(unknown)}")
        args = dict(lambdas={k: v for k, v in lam.items() if k in LAMBDA})
        args.update(kw)
        return cls(geometry, renderer, ref_image, ref_mask, H, W, **args)
