"""Static (refinement) stage surface: ``sugar.SuGaR`` + ``renderer.DiffSuGaRNormal`` (the reference's ``sugar`` geometry and
``diff-sugar-rasterizer-normal`` renderer).  The fused 6-channel call must equal the reference's formulation -- an
SH pass and a normal pass through the 3-channel operator (…_normal.py:161-195) -- in images and in the gradients of
every learnt parameter."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_static_renderer_equals_the_two_pass_formulation():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import dreammesh4d_amd.diff_gaussian_rasterization as dgr
    from dreammesh4d_amd import renderer as R, sugar, synthetic as syn

    dev = torch.device("cuda:0")
    sc = syn.mesh_bound_scene(1200, n_nodes=20, k=4, seed=7)
    g = sugar.SuGaR(sc["verts"], sc["faces"], vertex_colors=np.random.default_rng(0).random((len(sc["verts"]), 3)), device=dev)
    # (this test is about the RASTERIZER call: both sides read the attributes from the properties' torch operators; the fused
    # attribute kernel agrees with them to 2e-6, test_fused_sugar_attributes..., which is not bit-identical planes)
    g.fused_attributes = False
    with torch.no_grad():
        g._scales.add_(1.0)                                  # visible splats at 96^2
        g._scales[:, 1].add_(0.6)                            # anisotropic discs: the in-plane rotation (_quaternions) matters
        g._quaternions.add_(0.3 * torch.randn(g._quaternions.shape, generator=torch.Generator().manual_seed(3)).to(dev))
    # the reference's groups, the (empty) f_rest included (sugar.py:333-377)
    assert [d["name"] for d in g.optimize_list] == ["points", "f_dc", "f_rest", "all_densities", "scales", "quaternions"]
    H = W = 96
    cam = syn.make_camera(H, W, elev_deg=20.0, azim_deg=35.0)
    T = lambda a: torch.tensor(a, device=dev)
    fov = torch.tensor(cam.fovy)
    vc = R.Camera(FoVx=fov, FoVy=fov, camera_center=T(cam.campos), image_width=W, image_height=H,
                  world_view_transform=T(cam.viewmatrix), full_proj_transform=T(cam.projmatrix))
    r = R.DiffSuGaRNormal(g, invert_bg_prob=1.0)
    gen = torch.Generator().manual_seed(0)
    gw = {k: torch.randn(s, generator=gen).to(dev) for k, s in (("render", (3, H, W)), ("normal", (3, H, W)), ("depth", (1, H, W)), ("mask", (1, H, W)))}

    def loss_of(o):
        return sum((o[k] * gw[k]).sum() for k in gw)

    out = r.forward(vc, None, compute_normal_from_dist=False)
    loss_of(out).backward()
    params = {n: p for n, p in g.named_parameters() if p.requires_grad and p.numel()}
    got = {n: p.grad.clone() for n, p in params.items()}
    got_vsp = out["viewspace_points"].grad.clone()
    g.zero_grad(set_to_none=True)
    # the reference's two passes through the 3-channel operator
    bg = torch.ones(3, device=dev)
    rs = dgr.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam.tanfov, tanfovy=cam.tanfov, bg=bg,
                                           scale_modifier=1.0, viewmatrix=T(cam.viewmatrix), projmatrix=T(cam.projmatrix),
                                           sh_degree=0, campos=T(cam.campos), prefiltered=False, debug=False)
    rast = dgr.GaussianRasterizer(rs)
    xyz = g.get_xyz
    vsp = torch.zeros_like(xyz, requires_grad=True)
    img, radii, depth, alpha = rast(means3D=xyz, means2D=vsp, shs=g.get_features, opacities=g.get_opacity, scales=g.get_scaling,
                                    rotations=g.get_rotation)
    vsp2 = torch.zeros_like(xyz, requires_grad=True)     # (the reference passes plain zeros here, …_normal.py:188)
    nrm, _, _, _ = rast(means3D=xyz, means2D=vsp2, colors_precomp=g.get_gs_normals, opacities=g.get_opacity,
                        scales=g.get_scaling, rotations=g.get_rotation)
    mask = alpha > 0.99
    n_map = F.normalize(nrm, dim=0) * 0.5 * alpha + 0.5
    n_map = torch.where(mask.expand(3, H, W), n_map, n_map.detach())
    ref = {"render": img.clamp(0, 1), "normal": n_map, "depth": torch.where(mask, depth, depth.detach()), "mask": alpha}
    for k in ref:
        assert torch.equal(out[k], ref[k]), k                       # same kernels on the same values: bit-identical planes
    assert torch.equal(out["radii"], radii) and bool(out["visibility_filter"].any())
    loss_of(ref).backward()
    for n, p in params.items():
        a, b = got[n], p.grad
        assert float(b.abs().max()) > 0, n
        assert float((a - b).abs().max()) <= 5e-5 * float(b.abs().max()), (n, float((a - b).abs().max()), float(b.abs().max()))
    # viewspace_points.grad of the fused pass = the screen-space mean gradient of BOTH passes (the reference's leaf only
    # sees the RGB pass; it feeds densification statistics, which the mesh-bound stages do not use -- DESIGN.md)
    both = vsp.grad + vsp2.grad
    assert float((got_vsp - both).abs().max()) <= 5e-5 * float(both.abs().max())
    # batch_forward: reference batch dict -> stacked [B,H,W,C] outputs
    c2w = torch.tensor(np.stack([cam.c2w, syn.make_camera(H, W, azim_deg=120.0).c2w]), dtype=torch.float32)
    bo = r.eval().batch_forward({"c2w": c2w, "fovy": torch.tensor([cam.fovy, cam.fovy]), "height": H, "width": W})
    assert bo["comp_rgb"].shape == (2, H, W, 3) and bo["comp_mask"].shape == (2, H, W, 1) and len(bo["radii"]) == 2
    assert "comp_normal_from_dist" not in bo and float(bo["comp_mask"].max()) > 0.9


def test_static_stage_iteration_fits_the_reference_view_and_keeps_the_mesh_smooth():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import renderer as R, sugar, synthetic as syn
    from dreammesh4d_amd.mesh_reg import MeshLaplacianSmoothing, MeshNormalConsistency
    from dreammesh4d_amd.static_stage import StaticStage, tv_loss

    dev = torch.device("cuda:0")
    H = W = 96
    sc = syn.mesh_bound_scene(1200, n_nodes=20, k=4, seed=8)
    V = len(sc["verts"])
    rng = np.random.default_rng(1)
    target = sugar.SuGaR(sc["verts"], sc["faces"], vertex_colors=rng.random((V, 3)), device=dev)
    with torch.no_grad():
        target._scales.add_(1.0)
        ref = R.DiffSuGaRNormal(target).eval().batch_forward(
            {"c2w": torch.tensor(syn.make_camera(H, W, elev_deg=5.0, azim_deg=0.0).c2w, dtype=torch.float32)[None],
             "fovy": torch.tensor([math.radians(20.0)]), "height": H, "width": W})
        # (eval inverts the background: white -> black; the fit below only needs a fixed target)
        ref_img, ref_mask = ref["comp_rgb"].clone(), (ref["comp_mask"] > 0.5).float()
    # the shipped rates (sugar_static_refine.yaml:50-58) except a faster colour rate, so that 30 steps show a clear fit
    g = sugar.SuGaR(sc["verts"], sc["faces"], vertex_colors=np.full((V, 3), 0.5), device=dev, position_lr=0.00048, scaling_lr=0.005,
                    feature_lr=0.05, opacity_lr=0.02, rotation_lr=0.001, spatial_lr_scale=1.0)
    with torch.no_grad():
        g._scales.add_(1.0)
    stage = StaticStage(g, R.DiffSuGaRNormal(g, back_ground_color=(0.0, 0.0, 0.0)), ref_img, ref_mask, H, W, guidance=None, random_views=2,
                        normal_consistency=MeshNormalConsistency(sc["faces"], V, dev), laplacian_smoothing=MeshLaplacianSmoothing(sc["faces"], V, dev))
    first = stage.iteration()
    assert {"rgb", "mask", "normal_consistency", "laplacian_smoothing", "rgb_tv", "depth_tv", "normal_tv", "loss"} <= set(first)
    assert all(torch.isfinite(v) for v in first.values())
    hist = [float(stage.iteration()["rgb"]) for _ in range(30)]
    assert np.isfinite(hist).all() and np.mean(hist[-5:]) < 0.7 * float(first["rgb"]), (float(first["rgb"]), hist)
    assert {d["name"] for d in stage.opt.param_groups if "name" in d} == {"points", "f_dc", "f_rest", "all_densities", "scales", "quaternions"}
    assert float(g._points.grad.abs().max()) > 0 and stage.global_step == 31
    # tv_loss == the reference's formula on a case worked by hand: one step of height 1 across a 2x2 single-channel image
    x = torch.tensor([[[[0.0, 0.0], [1.0, 1.0]]]])
    assert abs(float(tv_loss(x)) - 2 * (2.0 / 2 + 0.0 / 2) / 1) < 1e-7


def test_message_adamw_step_equals_the_torch_optimiser():
    """One process on the HIP device: the static stage's AdamW as the fused message-space kernel (distributed.ShardedAdamW over the
    reducer's dense message) against torch's fused AdamW -- the same iteration from the same seeds.  After ONE step the parameters
    agree to rounding; after four the bulk still does, the rotations -- whose gradients are mostly rounding noise, and a step is
    +-lr whatever the gradient's size at eps = 1e-15 -- only as far as two runs of the SAME optimiser agree (measured: mean
    differences of 2e-5 torch against torch, 1e-4 message against message, 4e-4 across, at lr 1e-3)."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import renderer as R, sugar, synthetic as syn
    from dreammesh4d_amd.mesh_reg import MeshLaplacianSmoothing, MeshNormalConsistency
    from dreammesh4d_amd.static_stage import StaticStage

    dev = torch.device("cuda:0")
    H = W = 96
    sc = syn.mesh_bound_scene(900, n_nodes=20, k=4, seed=5)
    V = len(sc["verts"])
    gen = torch.Generator().manual_seed(2)
    ref_img, ref_mask = torch.rand(1, H, W, 3, generator=gen).to(dev), (torch.rand(1, H, W, 1, generator=gen) > 0.5).float().to(dev)

    def run(message, steps):
        g = sugar.SuGaR(sc["verts"], sc["faces"], vertex_colors=np.random.default_rng(3).random((V, 3)), device=dev, position_lr=0.00048,
                        scaling_lr=0.005, feature_lr=0.01, opacity_lr=0.02, rotation_lr=0.001, spatial_lr_scale=1.0)
        with torch.no_grad():
            g._scales.add_(1.0)
        stage = StaticStage(g, R.DiffSuGaRNormal(g), ref_img, ref_mask, H, W, guidance=None, random_views=2, message_adamw=message,
                            normal_consistency=MeshNormalConsistency(sc["faces"], V, dev), laplacian_smoothing=MeshLaplacianSmoothing(sc["faces"], V, dev))
        assert (stage.sharded is not None) == message
        losses = [float(stage.iteration()["loss"]) for _ in range(steps)]
        lrs = {n: max(float(gq["lr"]) for gq in stage.opt.param_groups if any(q is p for q in gq["params"])) for n, p in g.named_parameters()
               if p.requires_grad and p.numel()}
        return losses, {n: p.detach().clone() for n, p in g.named_parameters() if p.requires_grad and p.numel()}, lrs

    (l0, p0, lrs), (l1, p1, _) = run(False, 1), run(True, 1)
    assert abs(l0[0] - l1[0]) <= 1e-6 * abs(l0[0])                       # the first iteration is the same computation
    for n in p0:
        d = (p0[n] - p1[n]).abs()
        assert float(d.mean()) <= 1e-3 * lrs[n], (n, float(d.mean()), lrs[n])
        assert float(d.max()) <= 2.0 * lrs[n] + 1e-7, (n, float(d.max()), lrs[n])      # (an element whose gradient is noise may step the other way)
    (_, q0, _), (_, q1, _) = run(False, 4), run(True, 4)
    for n in q0:
        d = (q0[n] - q1[n]).abs()
        assert float(d.max()) <= 8.0 * lrs[n] + 1e-7, (n, float(d.max()), lrs[n])
        assert float(d.mean()) <= (1.0 if n == "_quaternions" else 0.02) * lrs[n], (n, float(d.mean()), lrs[n])


def test_batched_views_equal_the_loop_of_per_view_operator_calls():
    """``DiffSuGaRNormal.batch_forward`` with the views of the batch as ONE operator call (gviews.render_gaussian_views,
    dm4d_gviews_forward / _backward: no host synchronisation) against the loop of per-view drop-in operator calls: the same
    kernels on the same values -- every output plane bit-identical, radii identical; the parameter gradients equal up to the
    order in which the views' contributions are summed; viewspace_points.grad per view; normal-from-depth included."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import renderer as R, sugar, synthetic as syn

    dev = torch.device("cuda:0")
    sc = syn.mesh_bound_scene(1500, n_nodes=20, k=4, seed=11)
    g = sugar.SuGaR(sc["verts"], sc["faces"], vertex_colors=np.random.default_rng(2).random((len(sc["verts"]), 3)), device=dev)
    with torch.no_grad():
        g._scales.add_(1.0)
        g._scales[:, 1].add_(0.5)
        g._quaternions.add_(0.3 * torch.randn(g._quaternions.shape, generator=torch.Generator().manual_seed(4)).to(dev))
    H = W = 128
    cams = [syn.make_camera(H, W, elev_deg=e, azim_deg=a) for e, a in ((5.0, 0.0), (40.0, 100.0), (-5.0, -150.0))]
    B = len(cams)
    gen = torch.Generator().manual_seed(9)
    batch = {"c2w": torch.tensor(np.stack([c.c2w for c in cams]), dtype=torch.float32), "fovy": torch.tensor([c.fovy for c in cams]),
             "height": H, "width": W, "rays_o": torch.randn(B, H, W, 3, generator=gen).to(dev),
             "rays_d": F.normalize(torch.randn(B, H, W, 3, generator=gen), dim=-1).to(dev)}
    keys = ("comp_rgb", "comp_normal", "comp_depth", "comp_mask", "comp_normal_from_dist")
    gw = {k: torch.randn(B, H, W, 3 if k in ("comp_rgb", "comp_normal", "comp_normal_from_dist") else 1, generator=gen).to(dev) for k in keys}
    res = {}
    for mode in ("loop", "batched"):
        r = R.DiffSuGaRNormal(g)
        r.batched = mode == "batched"
        g.zero_grad(set_to_none=True)
        out = r.batch_forward(batch)
        sum((out[k] * gw[k]).sum() for k in keys).backward()
        assert (getattr(r, "views_renderer", None) is not None) == (mode == "batched")
        res[mode] = (out, {n: p.grad.clone() for n, p in g.named_parameters() if p.requires_grad and p.numel()},
                     [v.grad.clone() for v in out["viewspace_points"]])
    (a, ga, va), (b, gb, vb) = res["loop"], res["batched"]
    for k in keys:
        assert torch.equal(a[k], b[k]), k
    for i in range(B):
        assert torch.equal(a["radii"][i], b["radii"][i]) and torch.equal(a["visibility_filter"][i], b["visibility_filter"][i])
        assert float(va[i].abs().max()) > 0 and float((va[i] - vb[i]).abs().max()) <= 1e-6 * float(va[i].abs().max())
    for n in ga:
        assert float(ga[n].abs().max()) > 0, n
        assert float((ga[n] - gb[n]).abs().max()) <= 2e-5 * float(ga[n].abs().max()), (n, float((ga[n] - gb[n]).abs().max()), float(ga[n].abs().max()))
    # the capacity monitor of the batched path: no overflow on this scene, counters readable without a synchronisation
    vr = r.views_renderer
    assert float(vr.overflow_flag()) == 0.0 and vr.calibrated


@pytest.mark.parametrize("shape", [(5, 64, 48), (3, 32, 32)])
def test_static_head_equals_the_torch_composition(shape):
    """static_head.static_head (csrc/statichead.hip: one launch each way) against the operators it replaces -- the static renderer's
    epilogue (clamp, normal map, masks, detach rules) + StaticStage's masked MSEs, total-variation terms and the guidance's resize --
    on random images whose opacity straddles 0.99 and whose colours leave [0, 1]: the five terms and the gradients of colour, depth
    and opacity under random upstream weights."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd.renderer import _where_detached
    from dreammesh4d_amd.static_head import static_head
    from dreammesh4d_amd.static_stage import tv_loss

    dev = torch.device("cuda:0")
    B, H, W = shape
    n_ref, n_rnd = 1, B - 1
    g = torch.Generator().manual_seed(B * 100 + H)
    color = (torch.rand(B, 6, H, W, generator=g) * 1.6 - 0.3)
    color[:, 3:] = torch.randn(B, 3, H, W, generator=g)
    depth = torch.rand(B, 1, H, W, generator=g) * 3 + 1
    alpha = torch.rand(B, 1, H, W, generator=g)
    alpha[alpha > 0.5] = 0.992 + 0.008 * torch.rand(int((alpha > 0.5).sum()), generator=g)      # half the pixels solid
    ref_img = torch.rand(1, H, W, 3, generator=g).to(dev)
    ref_mask = (torch.rand(1, H, W, 1, generator=g) > 0.4).float().to(dev)
    w5 = torch.rand(5, generator=g).to(dev) + 0.5
    wh = torch.randn(n_rnd, H // 2, W // 2, 3, generator=g).to(dev)
    ref_pos = torch.tensor([0] + [-1] * n_rnd, dtype=torch.int32, device=dev)
    rnd_pos = torch.tensor([-1] + list(range(n_rnd)), dtype=torch.int32, device=dev)
    fidx = torch.zeros(1, dtype=torch.int64, device=dev)
    res = {}
    for mode in ("torch", "hip"):
        c, d, a = (t.clone().to(dev).requires_grad_(True) for t in (color, depth, alpha))
        if mode == "hip":
            t5, half = static_head(c, d, a, ref_pos, rnd_pos, ref_img, ref_mask, fidx, n_ref, n_rnd)
        else:
            mask = a > 0.99
            rgb = c[:, :3].clamp(0, 1)
            n = F.normalize(c[:, 3:], dim=1)
            n_map = _where_detached(n * 0.5 * a + 0.5, mask.expand(B, 3, H, W))
            dd = _where_detached(d, mask)
            m = ref_mask
            mse_rgb = F.mse_loss(ref_img * m, rgb[:1].permute(0, 2, 3, 1) * m)
            mse_mask = F.mse_loss(m, a[:1].permute(0, 2, 3, 1))
            t5 = torch.stack([mse_rgb, mse_mask, tv_loss(rgb[1:]), tv_loss(dd[1:]), tv_loss(n_map[1:])])
            half = F.interpolate(rgb[1:], (H // 2, W // 2), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
        ((t5 * w5).sum() + (half * wh).sum()).backward()
        res[mode] = (t5.detach(), half.detach(), c.grad.clone(), d.grad.clone(), a.grad.clone())
    ref, got = res["torch"], res["hip"]
    assert float((ref[0] - got[0]).abs().max() / ref[0].abs().max()) <= 2e-6, (ref[0], got[0])
    assert torch.equal(ref[1], got[1])                                  # the 2 x 2 mean in the library's order
    for k, name in ((2, "color"), (3, "depth"), (4, "alpha")):
        scale = float(ref[k].abs().max())
        assert scale > 0 and float((ref[k] - got[k]).abs().max()) <= 2e-5 * scale, (name, float((ref[k] - got[k]).abs().max()), scale)
    # where the opacity is below the threshold depth and the normal channels receive nothing
    thin = (alpha <= 0.99).to(dev)
    assert float(got[3][thin].abs().max()) == 0.0 and float(got[2][:, 3:][thin.expand(B, 3, H, W)].abs().max()) == 0.0


@pytest.mark.parametrize("G", [6, 3, 1])
def test_fused_sugar_attributes_equal_the_torch_properties(G):
    """``SuGaR.render_attributes`` through csrc/sugar_attr.hip (one launch each way) against the properties' torch operators and
    their autograd: every attribute the renderer reads, and the gradient of every learnt parameter under random upstream weights --
    with rotated in-plane frames, colours on both sides of the SH clip and of the zero clamp."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import sugar, synthetic as syn

    dev = torch.device("cuda:0")
    sc = syn.mesh_bound_scene(900, n_nodes=10, k=4, seed=5)
    res = {}
    for mode in ("torch", "hip"):
        g = sugar.SuGaR(sc["verts"], sc["faces"], n_gaussians_per_surface_triangle=G, vertex_colors=np.random.default_rng(3).random((len(sc["verts"]), 3)),
                        device=dev, color_clip=1.2)
        gen = torch.Generator().manual_seed(7)
        with torch.no_grad():
            g._quaternions.copy_(torch.randn(g._quaternions.shape, generator=gen).to(dev))
            g._scales.add_(0.3 * torch.randn(g._scales.shape, generator=gen).to(dev))
            g.all_densities.add_(torch.randn(g.all_densities.shape, generator=gen).to(dev))
            g._sh_coordinates_dc.mul_(2.5)                                     # beyond the clip on both sides, below the zero clamp
            g._points.add_(0.01 * torch.randn(g._points.shape, generator=gen).to(dev))
        g.fused_attributes = mode == "hip"
        a = g.render_attributes()
        assert ("colors6" in a) == (mode == "hip")
        w = {k: torch.randn(a[k].shape, generator=gen).to(dev) for k in ("xyz", "opacity", "scaling", "rotation", "rgb", "normals")}
        sum((a[k] * w[k]).sum() for k in w).backward()
        res[mode] = ({k: a[k].detach().clone() for k in w}, {n: p.grad.clone() for n, p in g.named_parameters() if p.requires_grad and p.numel()})
    (va, ga), (vb, gb) = res["torch"], res["hip"]
    for k in va:
        assert float((va[k] - vb[k]).abs().max()) <= 2e-6 * max(float(va[k].abs().max()), 1.0), (k, float((va[k] - vb[k]).abs().max()))
    assert set(ga) == set(gb) and len(ga) >= 5
    for n in ga:
        scale = float(ga[n].abs().max())
        assert scale > 0 and float((ga[n] - gb[n]).abs().max()) <= 3e-5 * scale, (n, float((ga[n] - gb[n]).abs().max()), scale)
