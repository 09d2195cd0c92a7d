"""-m gpu: the plugins of the hot path constructed the way threestudio constructs them -- ``find(<x>_type)(cfg.<x>, *args)``
-- from the ``system`` blocks of the two shipped configurations, then driven through one step.

The blocks below are the keys / values of custom/threestudio-dreammesh4d/configs/sugar_dynamic_dg.yaml (:50-170) and
configs/sugar_static_refine.yaml (:30-150), typed in (only what a run has to provide -- ``???`` entries, the mesh and the
Zero123 checkpoint -- is filled with test stand-ins: a small sphere PLY, a reduced-width random-weight Zero123, seeded
conditioning embeddings)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DATA = {"default_camera_distance": 3.8, "default_elevation_deg": 5.0, "default_azimuth_deg": 0.0, "default_fovy_deg": 20.0,
        "video_length": 32, "height": 512, "width": 512}

DYNAMIC_SYSTEM = {     # sugar_dynamic_dg.yaml, system:
    "stage": "motion", "num_inter_frames": 10, "length_inter_frames": 0.1,
    "geometry_type": "dynamic-sugar",
    "geometry": {"num_frames": 32, "use_deform_graph": True, "dynamic_mode": "deformation", "n_dg_nodes": 1000, "dg_node_connectivity": 4,
                 "dg_trans_lr": 0.002, "dg_rot_lr": 0.001, "dg_scale_lr": 0.001, "vert_trans_lr": 0.001, "vert_rot_lr": 0.001,
                 "vert_scale_lr": 0.001, "deformation_lr": 0.00032, "grid_lr": 0.0032, "d_scale": False,
                 "spatial_extent": "${data.default_camera_distance}", "spatial_lr_scale": 1, "surface_mesh_to_bind_path": "",
                 "n_gaussians_per_surface_triangle": 6, "dist_mode": "geodisc", "skinning_method": "hybrid"},
    "renderer_type": "diff-sugar-rasterizer-temporal", "renderer": {"debug": False, "invert_bg_prob": 1.0},
    "material_type": "no-material", "material": {"n_output_dims": 0},
    "background_type": "solid-color-background",
    "guidance_zero123_type": "temporal-stable-zero123-guidance",
    "guidance_zero123": {"num_frames": "${data.video_length}", "pretrained_config": "./load/zero123/sd-objaverse-finetune-c_concat-256.yaml",
                         "pretrained_model_name_or_path": "???", "vram_O": "${not:${gt0:${system.freq.guidance_eval}}}",
                         "cond_video_dir": "???", "cond_elevation_deg": "${data.default_elevation_deg}",
                         "cond_azimuth_deg": "${data.default_azimuth_deg}", "cond_camera_distance": "${data.default_camera_distance}",
                         "guidance_scale": 3.0, "min_step_percent": 0.02, "max_step_percent": 0.5, "chunk_size": None},
    "freq": {"ref_only_steps": 0, "guidance_eval": 0, "inter_frame_reg": 0, "milestone_arap_reg": 100},
    "loss": {"lambda_sds_zero123": 0.1, "lambda_rgb": 5000.0, "lambda_mask": [200, 500.0, 5000.0, 1000], "lambda_depth": 0.0,
             "lambda_normal_consistency": 100.0, "lambda_arap_reg_key_frame": 10.0, "lambda_arap_reg_inter_frame": 10.0},
    "optimizer": {"name": "Adam", "args": {"lr": 0.01, "betas": [0.9, 0.99], "eps": 1.0e-15}, "params": {"background": {"lr": 0.001}}},
}

STATIC_SYSTEM = {      # sugar_static_refine.yaml, system:
    "stage": "sugar",
    "geometry_type": "sugar",
    "geometry": {"position_lr": 0.00048, "feature_lr": 0.001, "opacity_lr": 0.02, "scaling_lr": 0.005, "rotation_lr": 0.001,
                 "spatial_extent": "${data.default_camera_distance}", "spatial_lr_scale": 1, "n_gaussians_per_surface_triangle": 6,
                 "learnable_positions": True, "surface_mesh_to_bind_path": "???", "init_gs_opacity": 0.9, "init_gs_scales_s": 1.3},
    "renderer_type": "diff-sugar-rasterizer-normal", "renderer": {"debug": False, "invert_bg_prob": 1.0},
    "material_type": "no-material", "material": {"n_output_dims": 0},
    "background_type": "solid-color-background",
    "guidance_type": "stable-zero123-guidance",
    "guidance": {"pretrained_config": "./load/zero123/sd-objaverse-finetune-c_concat-256.yaml", "pretrained_model_name_or_path": "???",
                 "vram_O": "${not:${gt0:${system.freq.guidance_eval}}}", "cond_image_path": "load/images/x_rgba.png",
                 "cond_elevation_deg": "${data.default_elevation_deg}", "cond_azimuth_deg": "${data.default_azimuth_deg}",
                 "cond_camera_distance": "${data.default_camera_distance}", "guidance_scale": 3.5, "min_step_percent": 0.02,
                 "max_step_percent": 0.2},
    "freq": {"ref_only_steps": 0, "guidance_eval": 0, "input_normal": 10000, "start_sugar_reg": 3000, "reset_neighbors": 50},
    "loss": {"lambda_sds": 0.01, "lambda_rgb": 1000.0, "lambda_mask": 100.0, "lambda_normal_consistency": 10.0, "lambda_laplacian_smoothing": 1.0},
    "optimizer": {"name": "Adam", "args": {"lr": 0.01, "betas": [0.9, 0.99], "eps": 1.0e-15}, "params": {"background": {"lr": 0.001}}},
}


def _stand_ins(tmp_path, n_frames, dev):
    """What a run provides next to the YAML: the refined mesh (PLY), the Zero123 model, the conditioning embeddings."""
    from dreammesh4d_amd import synthetic as syn, wire_formats as wf, zero123 as z

    v, f = syn.uv_sphere(6000, radius=0.6)
    mesh = str(tmp_path / "exported_mesh.ply")
    wf.write_ply(mesh, np.asarray(v), np.asarray(f), colors=np.random.default_rng(0).random((len(v), 3)))
    torch.manual_seed(0)
    model = z.Zero123(unet_kwargs=dict(model_channels=32, context_dim=32, num_heads=4), vae_kwargs=dict(ch=32)).to(dev)
    for p in model.model.diffusion_model.out.parameters():
        torch.nn.init.normal_(p, std=0.05)
    emb = str(tmp_path / "cond_embeddings.pt")
    torch.save({"c_crossattn": torch.randn(n_frames, 1, 32), "c_concat": torch.randn(n_frames, 4, 32, 32)}, emb)
    return mesh, model, emb


def _batch(B, H, W, dev, timestamps=None):
    from dreammesh4d_amd import renderer as R, synthetic as syn

    fovy = math.radians(20.0)
    c2w = torch.stack([torch.tensor(syn.orbit_c2w(10.0 + 15 * b, 40.0 * b, 3.8), dtype=torch.float32) for b in range(B)]).to(dev)
    dirs = R.ray_directions(H, W, 0.5 * H / math.tan(0.5 * fovy), device=dev)
    rays_o, rays_d = R.rays(dirs, c2w)
    batch = {"c2w": c2w, "fovy": torch.full((B,), fovy, device=dev), "height": H, "width": W, "rays_o": rays_o, "rays_d": rays_d}
    if timestamps is not None:
        batch["timestamp"] = timestamps
        batch["frame_indices"] = torch.arange(B, device=dev)
    return batch


def test_dynamic_stage_plugins_from_the_shipped_config_block(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import threestudio_host as ts
    from dreammesh4d_amd.schedule import C

    dev = torch.device("cuda:0")
    mesh, model, emb = _stand_ins(tmp_path, 32, dev)
    cfg = ts.resolve({"data": DATA, "system": DYNAMIC_SYSTEM})["system"]
    cfg["geometry"]["surface_mesh_to_bind_path"] = mesh                      # the `???` / command-line entries of a run
    cfg["guidance_zero123"].update(pretrained_model_name_or_path="(test: model passed in)", cond_video_dir="(test)", cond_embeddings_path=emb)
    # ---- what BaseLift3DSystem.configure does (threestudio/systems/base.py:262-282): find(type)(cfg, ...)
    geometry = ts.find(cfg["geometry_type"])(cfg["geometry"])
    material = ts.find(cfg["material_type"])(cfg["material"])
    background = ts.find(cfg["background_type"])(cfg.get("background"))
    renderer = ts.find(cfg["renderer_type"])(cfg["renderer"], geometry=geometry, material=material, background=background)
    guidance = ts.find(cfg["guidance_zero123_type"])(cfg["guidance_zero123"], model=model)
    assert geometry.cfg.spatial_extent == 3.8 and geometry.cfg.dg_node_connectivity == 4 and geometry.skinning_method == "hybrid"
    assert geometry._xyz_neighbor_node_idx.shape == (geometry.n_verts, 4) and geometry._deform_graph_node_xyz.shape == (1000, 3)
    assert abs(float(geometry.surface_mesh_thickness) - 3.8e-6) < 1e-12 and geometry.n_gaussians == geometry.n_faces * 6
    assert not any(p.requires_grad for p in (geometry._points, geometry._scales, geometry.all_densities, geometry._sh_coordinates_dc))
    assert renderer.geometry is geometry and renderer.material is material and renderer.background is background
    assert guidance.cfg.num_frames == 32 and guidance.cfg.vram_O is True and guidance.guidance_scale == 3.0 and guidance.max_step == 500
    # ---- configure_optimizers (C/system/sugar_4dgen.py:66-76): parse_optimizer(cfg.optimizer, self) merged into the geometry's AdamW
    system = torch.nn.Module()
    system.geometry, system.background = geometry, background
    opt = geometry.merge_optimizer(ts.parse_optimizer(cfg["optimizer"], system))
    names = [g.get("name") for g in opt.param_groups]
    assert isinstance(opt, torch.optim.AdamW) and names[:2] == ["deformation", "grid"] and len(opt.param_groups) == 3
    # perturb the zero-initialised heads so that the mesh moves
    with torch.no_grad():
        for n, p in geometry._deformation.named_parameters():
            if "_deform" in n:
                p.add_(0.01 * torch.randn_like(p))
    # ---- one training substep the way the system drives the plugins
    for m in (geometry, renderer, guidance):
        m.do_update_step(0, 0)
    geometry.update_learning_rate(0)
    B, H, W = 4, 256, 256
    ts_ = torch.linspace(0, 1, 34, device=dev)[1:-1][[3, 3, 17, 17]]
    out = renderer.batch_forward(_batch(B, H, W, dev, timestamps=ts_))
    assert set(out) >= {"comp_rgb", "comp_normal", "comp_normal_from_dist", "comp_depth", "comp_mask", "viewspace_points", "visibility_filter", "radii"}
    assert out["comp_rgb"].shape == (B, H, W, 3) and out["comp_mask"].shape == (B, H, W, 1)
    g = guidance(out["comp_rgb"], torch.tensor([10.0, 25.0, 40.0, 55.0], device=dev), torch.tensor([0.0, 40.0, 80.0, 120.0], device=dev),
                 torch.full((B,), 3.8, device=dev), frame_indices=torch.tensor([3, 3, 17, 17], device=dev))
    loss = C(cfg["loss"]["lambda_sds_zero123"], 0, 0) * g["loss_sds"] + C(cfg["loss"]["lambda_mask"], 0, 0) * out["comp_mask"].mean()
    before = [p.detach().clone() for p in geometry._deformation.get_mlp_parameters()]
    opt.zero_grad()
    loss.backward()
    opt.step()
    assert any(not torch.equal(a, b) for a, b in zip(before, geometry._deformation.get_mlp_parameters()))
    assert all(torch.isfinite(p).all() for p in geometry._deformation.parameters())
    assert out["viewspace_points"][0].grad is not None


def test_static_stage_plugins_from_the_shipped_config_block(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import threestudio_host as ts

    dev = torch.device("cuda:0")
    mesh, model, emb = _stand_ins(tmp_path, 1, dev)
    cfg = ts.resolve({"data": DATA, "system": STATIC_SYSTEM})["system"]
    with pytest.raises(ValueError):
        ts.find("sugar")(cfg["geometry"])                                     # surface_mesh_to_bind_path: ??? not provided
    cfg["geometry"]["surface_mesh_to_bind_path"] = mesh
    cfg["guidance"].update(pretrained_model_name_or_path="(test: model passed in)", cond_embeddings_path=emb)
    geometry = ts.find(cfg["geometry_type"])(cfg["geometry"])
    renderer = ts.find(cfg["renderer_type"])(cfg["renderer"], geometry=geometry, material=ts.find(cfg["material_type"])(cfg["material"]),
                                             background=ts.find(cfg["background_type"])(None))
    guidance = ts.find(cfg["guidance_type"])(cfg["guidance"], model=model)
    assert abs(float(torch.sigmoid(geometry.all_densities[0])) - 0.9) < 1e-6                     # init_gs_opacity: 0.9
    lr = {g["name"]: g["lr"] for g in geometry.optimizer.param_groups}
    assert lr["points"] == 0.00048 * 1 and lr["f_dc"] == 0.001 and lr["all_densities"] == 0.02 and lr["quaternions"] == 0.001
    assert geometry.color_clip == 2.0 and guidance.guidance_scale == 3.5 and guidance.max_step == 200 and guidance.weights_dtype == torch.float32
    # colours leave [-clip, clip] / go negative: the rendered colour is clamped with a zero gradient there (advisor finding)
    with torch.no_grad():
        geometry._sh_coordinates_dc[0] = 5.0
        geometry._sh_coordinates_dc[1] = -3.0
    rgb = geometry.get_rendered_rgb()
    assert abs(float(rgb[0, 0]) - (0.28209479177387814 * 2.0 + 0.5)) < 1e-6 and float(rgb[1, 0]) == 0.0
    B, H, W = 2, 256, 256
    out = renderer.batch_forward(_batch(B, H, W, dev))
    g = guidance(out["comp_rgb"], torch.tensor([10.0, 25.0], device=dev), torch.tensor([0.0, 40.0], device=dev), torch.full((B,), 3.8, device=dev))
    opt = geometry.merge_optimizer(None)
    (g["loss_sds"] + out["comp_mask"].mean() + out["comp_normal"].mean()).backward()
    assert not geometry._sh_coordinates_dc.grad[0].any() and not geometry._sh_coordinates_dc.grad[1].any()    # clipped / clamped: no gradient
    assert geometry._sh_coordinates_dc.grad[2:].abs().sum() > 0 and geometry._points.grad.abs().sum() > 0
    opt.step()
    assert all(torch.isfinite(p).all() for p in geometry.parameters())


def test_checkpoint_hand_off_static_stage_to_dynamic_stage_and_resume(tmp_path):
    """the stage-2 checkpoint (``geometry.*`` keys of a Lightning ``state_dict``, threestudio/utils/misc.py:33-63) is
    what the dynamic stage starts from (``system.weights``, loaded non-strict: the deformation network and the graph are new) --
    the same Gaussians must come out of both geometries -- and a dynamic-stage checkpoint resumes bit-identically."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import threestudio_host as ts, wire_formats as wf

    dev = torch.device("cuda:0")
    mesh, _, _ = _stand_ins(tmp_path, 1, dev)
    scfg = ts.resolve({"data": DATA, "system": STATIC_SYSTEM})["system"]
    scfg["geometry"]["surface_mesh_to_bind_path"] = mesh
    static = ts.find("sugar")(scfg["geometry"])
    bg = ts.find("solid-color-background")(None)
    torch.manual_seed(3)
    with torch.no_grad():                                                       # "trained": every learnable tensor moved
        # (colours kept inside (0, 1): the static renderer clamps SH colours at 0, the dynamic one blends SH2RGB as is --
        #  sugar.py:640-661 vs the rasterizer's SH path -- so the two stages only agree where nothing is clamped)
        static._sh_coordinates_dc.copy_((torch.rand_like(static._sh_coordinates_dc) * 0.9 + 0.05 - 0.5) / 0.28209479177387814)
        static._scales.add_(0.2 * torch.randn_like(static._scales))
        static._quaternions.add_(0.2 * torch.randn_like(static._quaternions))
        static.all_densities.add_(torch.randn_like(static.all_densities))
        static._points.add_(0.002 * torch.randn_like(static._points))
    ck = tmp_path / "static_last.ckpt"
    wf.save_checkpoint(ck, {"geometry": static, "background": bg}, epoch=0, global_step=2000)
    keys = set(torch.load(ck, weights_only=False)["state_dict"])
    assert {"geometry._points", "geometry._surface_mesh_faces", "geometry.surface_mesh_thickness", "geometry.all_densities",
            "geometry._sh_coordinates_dc", "geometry._sh_coordinates_rest", "geometry._scales", "geometry._quaternions"} <= keys
    # ---- dynamic stage: constructed from the mesh, then the stage-2 weights (the vertices in the file win over the PLY's)
    dcfg = ts.resolve({"data": DATA, "system": DYNAMIC_SYSTEM})["system"]
    dcfg["geometry"]["surface_mesh_to_bind_path"] = mesh
    dyn = ts.find("dynamic-sugar")(dcfg["geometry"])
    missing, unexpected, epoch, step = wf.load_geometry(dyn, ck, strict=False)
    assert (epoch, step) == (0, 2000)
    assert all(k.startswith(("_deformation", "_deform_graph", "_xyz_neighbor", "_dg_", "_vert_")) for k in missing), missing
    assert all(k in ("_bary",) for k in unexpected), unexpected
    for name in ("_points", "_scales", "_quaternions", "all_densities", "_sh_coordinates_dc", "_surface_mesh_faces"):
        assert torch.equal(getattr(dyn, name), getattr(static, name)), name
    mat = ts.find("no-material")({"n_output_dims": 0})
    r_static = ts.find("diff-sugar-rasterizer-normal")(scfg["renderer"], geometry=static, material=mat, background=bg)
    r_dyn = ts.find("diff-sugar-rasterizer-temporal")(dcfg["renderer"], geometry=dyn, material=mat, background=bg)
    B, H, W = 2, 192, 192
    t_mid = torch.tensor([0.4, 0.4], device=dev)
    with torch.no_grad():
        a = r_static.batch_forward(_batch(B, H, W, dev))
        b = r_dyn.batch_forward(_batch(B, H, W, dev, timestamps=t_mid))
    # zero-initialised deformation heads: the dynamic geometry at any timestamp IS the static one (skinning by identity transforms)
    assert float((a["comp_rgb"] - b["comp_rgb"]).abs().max()) < 2e-3 and float((a["comp_rgb"] - b["comp_rgb"]).abs().mean()) < 1e-5
    assert float((a["comp_mask"] - b["comp_mask"]).abs().mean()) < 1e-5
    # ---- resume: a dynamic checkpoint into a freshly constructed geometry (the graph is not in the file -- as in the reference it is
    #      rebuilt from the mesh at construction; `dg_node_seed` makes the node samples the same ones)
    with torch.no_grad():
        for n, p in dyn._deformation.named_parameters():
            if "_deform" in n or "grids" in n:
                p.add_(0.02 * torch.randn_like(p))
    ck2 = tmp_path / "dynamic_last.ckpt"
    wf.save_checkpoint(ck2, {"geometry": dyn}, epoch=1, global_step=700)
    dcfg2 = ts.resolve({"data": DATA, "system": DYNAMIC_SYSTEM})["system"]
    dcfg2["geometry"].update(surface_mesh_to_bind_path=mesh)
    dyn2 = ts.find("dynamic-sugar")(dcfg2["geometry"])
    missing, unexpected, epoch, step = wf.load_geometry(dyn2, ck2, strict=True)
    assert not missing and not unexpected and (epoch, step) == (1, 700)
    r_dyn2 = ts.find("diff-sugar-rasterizer-temporal")(dcfg2["renderer"], geometry=dyn2, material=mat, background=bg)
    with torch.no_grad():
        c = r_dyn.batch_forward(_batch(B, H, W, dev, timestamps=t_mid))
        d = r_dyn2.batch_forward(_batch(B, H, W, dev, timestamps=t_mid))
    assert float((c["comp_rgb"] - b["comp_rgb"]).abs().max()) > 1e-3           # the deformation moved the mesh
    for k in ("comp_rgb", "comp_mask", "comp_depth", "comp_normal"):
        assert torch.equal(c[k], d[k]), k


def test_systems_and_datamodules_by_name_drive_an_iteration_from_the_config_block(tmp_path):
    """`sugar-4dgen-system` / `sugar-static-system` / `temporal-image-datamodule` / `single-image-datamodule`
    are registered names; constructed as find(type)(cfg, ...) they read `system.loss`, `system.freq`, `num_inter_frames`,
    `length_inter_frames` and the geometry's learning rates from the block (C/system/sugar_4dgen.py:28-76,296-330) and one
    training_step is one full iteration (render, losses, backward, optimiser)."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import copy

    from dreammesh4d_amd import threestudio_host as ts

    dev = torch.device("cuda:0")
    L = 8
    mesh, model, emb = _stand_ins(tmp_path, L, dev)
    data_cfg = dict(DATA, video_length=L, height=128, width=128, num_frames=4, random_camera={"batch_size": 1})
    g = torch.Generator().manual_seed(0)
    frames, masks = torch.rand(L, 128, 128, 3, generator=g), (torch.rand(L, 128, 128, 1, generator=g) > 0.5).float()
    data = ts.find("temporal-image-datamodule")(data_cfg, frames=frames, masks=masks)
    assert torch.allclose(data.timestamps, torch.linspace(0, 1, L + 2)[1:-1]) and data.random_views_per_frame == 1
    cfg = copy.deepcopy(ts.resolve({"data": DATA, "system": DYNAMIC_SYSTEM})["system"])
    cfg["geometry"].update(surface_mesh_to_bind_path=mesh, n_dg_nodes=120, num_frames=L)
    cfg["guidance_zero123"].update(pretrained_model_name_or_path="(test)", cond_video_dir="(test)", cond_embeddings_path=emb, num_frames=L)
    cfg["loss"].update(lambda_rgb=1234.0, lambda_mask=[0, 7.0, 70.0, 10])          # not the defaults: they must come from the block
    cfg["freq"].update(milestone_arap_reg=1)
    cfg["num_inter_frames"], cfg["length_inter_frames"] = 3, 0.2
    system = ts.find("sugar-4dgen-system")(cfg, data, model=model)
    st = system.stage
    assert st.lam["rgb"] == 1234.0 and st.lam["mask"] == [0, 7.0, 70.0, 10] and st.lam["sds_zero123"] == 0.1
    assert st.milestone_arap_reg == 1 and st.num_inter_frames == 3 and st.length_inter_frames == 0.2 and st.frames_per_step == 4
    assert st.sched == {"deformation": 0.00032, "grid": 0.0032} and st.arap is not None and st.normal_consistency is not None
    with torch.no_grad():
        for n, p in system.geometry._deformation.named_parameters():
            if "_deform" in n:
                p.add_(0.01 * torch.randn_like(p))
    before = [p.detach().clone() for p in system.geometry._deformation.get_mlp_parameters()]
    t0 = system.training_step()
    t1 = system.training_step()
    assert {"rgb", "mask", "sds", "normal_consistency"} <= set(t0) and "arap_reg_key_frame" in t1
    assert all(torch.isfinite(v).all() for v in t1.values() if torch.is_tensor(v)) and st.global_step == 2
    assert any(not torch.equal(a, b) for a, b in zip(before, system.geometry._deformation.get_mlp_parameters()))
    # a term the loop cannot compute must not be silently dropped: lambda_depth needs the data module's ref_depth (round 5: the optional
    # terms of system/sugar_4dgen.py:181-300 are computed when they have a weight AND their inputs), an unknown name is refused
    bad = copy.deepcopy(cfg)
    bad["loss"]["lambda_depth"] = 0.05
    with pytest.raises(ValueError):
        ts.find("sugar-4dgen-system")(bad, data, model=None)
    bad = copy.deepcopy(cfg)
    bad["loss"]["lambda_not_a_term_of_the_reference"] = 0.05
    with pytest.raises(NotImplementedError):
        ts.find("sugar-4dgen-system")(bad, data, model=None)
    tv = copy.deepcopy(cfg)
    tv["loss"]["lambda_rgb_tv"] = 1.0
    tvs = ts.find("sugar-4dgen-system")(tv, data, model=None)
    assert "rgb_tv/ref" in tvs.training_step()
    # ---- static stage by name
    scfg = copy.deepcopy(ts.resolve({"data": DATA, "system": STATIC_SYSTEM})["system"])
    scfg["geometry"]["surface_mesh_to_bind_path"] = mesh
    scfg["guidance"].update(pretrained_model_name_or_path="(test)", cond_embeddings_path=emb)
    scfg["loss"].update(lambda_rgb=77.0, lambda_opacity_binary=1.0)             # (a stage-"gaussian" term: gated by start_sugar_reg)
    sdata = ts.find("single-image-datamodule")(dict(DATA, height=128, width=128, random_camera={"batch_size": 2}), image=frames[0], mask=masks[0])
    s_model = _stand_ins(tmp_path, 1, dev)[1]
    ssys = ts.find("sugar-static-system")(scfg, sdata, model=s_model)
    assert ssys.stage.lam["rgb"] == 77.0 and ssys.stage.rv == 2
    out = ssys.training_step()
    assert torch.isfinite(out["loss"]) and ssys.stage.global_step == 1


def test_static_learnable_dynamic_geometry_gets_appearance_gradients(tmp_path):
    """`static_learnable: true` (dynamic_sugar.py:47,79-87): the static appearance tensors receive gradients through the
    FULL blend backward and -- as in the reference's training_setup_dynamic -- stay out of the optimiser."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import copy

    from dreammesh4d_amd import threestudio_host as ts

    dev = torch.device("cuda:0")
    mesh, _, _ = _stand_ins(tmp_path, 1, dev)
    cfg = copy.deepcopy(ts.resolve({"data": DATA, "system": DYNAMIC_SYSTEM})["system"])
    cfg["geometry"].update(surface_mesh_to_bind_path=mesh, n_dg_nodes=100, static_learnable=True)
    geometry = ts.find("dynamic-sugar")(cfg["geometry"])
    assert all(p.requires_grad for p in (geometry._scales, geometry.all_densities, geometry._sh_coordinates_dc))
    opt_params = {id(p) for g in geometry.optimizer.param_groups for p in g["params"]}
    assert not any(id(p) in opt_params for p in (geometry._scales, geometry.all_densities, geometry._sh_coordinates_dc))
    renderer = ts.find("diff-sugar-rasterizer-temporal")(cfg["renderer"], geometry=geometry, material=ts.find("no-material")(None),
                                                         background=ts.find("solid-color-background")(None))
    B, H, W = 2, 128, 128
    out = renderer.batch_forward(_batch(B, H, W, dev, timestamps=torch.tensor([0.3, 0.6], device=dev)))
    (out["comp_rgb"].mean() + out["comp_mask"].mean()).backward()
    for p in (geometry._scales, geometry.all_densities, geometry._sh_coordinates_dc):
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0


def test_guidance_constructs_on_the_device_from_the_shipped_keys_in_half_precision(tmp_path):
    """`temporal-stable-zero123-guidance` from its YAML block's OWN keys on the HIP device with `half_precision_weights: true` (the
    shipped default): frames from cond_video_dir, CLIP tower + VAE encoder from the checkpoint, no cond_embeddings_path -- the float16
    embeddings agree with the float32 ones computed on the CPU to float16 accuracy, and a guidance call runs on them."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from tests.test_prepare_embeddings_cpu import _tiny_checkpoint, _write_frames

    from dreammesh4d_amd import clip_vit, threestudio_host as ts

    dev = torch.device("cuda:0")
    ckpt, yml, model, tower = _tiny_checkpoint(tmp_path)
    vdir = str(tmp_path / "video")
    _write_frames(vdir, 2)
    cfg = {"num_frames": 2, "pretrained_config": yml, "pretrained_model_name_or_path": ckpt, "vram_O": True, "cond_video_dir": vdir,
           "cond_elevation_deg": 5.0, "cond_azimuth_deg": 0.0, "cond_camera_distance": 3.8, "guidance_scale": 3.0, "min_step_percent": 0.02,
           "max_step_percent": 0.5, "chunk_size": None, "half_precision_weights": True}
    g = ts.find("temporal-stable-zero123-guidance")(cfg)
    assert g.c_crossattn.is_cuda and g.c_crossattn.dtype == torch.float16 and tuple(g.c_concat.shape) == (2, 4, 32, 32)
    paths = [clip_vit.video_frame_path(vdir, i) for i in range(2)]
    _, cc, ct = clip_vit.prepare_embeddings(model, tower.eval(), paths, torch.device("cpu"), torch.float32)
    for got, want in ((g.c_crossattn, cc), (g.c_concat, ct)):
        err = float((got.float().cpu() - want).abs().max()) / max(float(want.abs().max()), 1e-6)
        assert err < 3e-2, err
    rgb = torch.rand(2, 64, 64, 3, device=dev, requires_grad=True)
    out = g(rgb, torch.tensor([10.0, 20.0]), torch.tensor([30.0, -40.0]), torch.full((2,), 3.8), frame_indices=torch.tensor([0, 1]))
    assert torch.isfinite(out["loss_sds"])
