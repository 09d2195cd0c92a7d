"""BASELINE.json `configs` at their stated sizes through the iteration harnesses (configs[0] and configs[3..4] are
covered by tests/test_raster_gpu.py, bench.py and tools/stress_cfg5.py):

  configs[1]  sugar_static_refine: 50k mesh-bound Gaussians, Zero123 SDS, 512^2 single view
  configs[2]  sugar_dynamic_dg: 16-frame synthetic video, 100k Gaussians, LBS skinning

Full-size geometry and image; the Zero123 networks are reduced-width random-weight instances (the full-size run is
bench.py's `dynamic_stage_iters_per_sec`)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _small_zero123(L, dev):
    from dreammesh4d_amd import zero123 as z

    model = z.Zero123(unet_kwargs=dict(model_channels=32, context_dim=32, num_heads=4), vae_kwargs=dict(ch=32))
    for p in model.model.diffusion_model.out.parameters():
        torch.nn.init.normal_(p, std=0.05)
    return model


def test_config1_static_refine_50k_gaussians_512():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import renderer as R, sugar, synthetic as syn, zero123 as z
    from dreammesh4d_amd.mesh_reg import MeshLaplacianSmoothing, MeshNormalConsistency
    from dreammesh4d_amd.static_stage import StaticStage

    dev = torch.device("cuda:0")
    H = W = 512
    sc = syn.mesh_bound_scene(8334, n_nodes=50, k=4, seed=0)
    V = len(sc["verts"])
    g = sugar.SuGaR(sc["verts"], sc["faces"], vertex_colors=np.random.default_rng(0).random((V, 3)), device=dev, position_lr=0.00048,
                    scaling_lr=0.005, feature_lr=0.001, opacity_lr=0.02, rotation_lr=0.001, spatial_lr_scale=1.0)
    assert g.n_gaussians == 50004
    torch.manual_seed(0)
    guid = z.StableZero123Guidance(_small_zero123(1, dev), torch.randn(1, 1, 32), torch.randn(1, 4, 32, 32), cond_elevation_deg=5.0,
                                   half_precision_weights=False).to(dev)
    ref_img = torch.rand(1, H, W, 3, device=dev)
    ref_mask = (torch.rand(1, H, W, 1, device=dev) > 0.5).float()
    stage = StaticStage(g, R.DiffSuGaRNormal(g), ref_img, ref_mask, H, W, guidance=guid, random_views=1,
                        normal_consistency=MeshNormalConsistency(sc["faces"], V, dev), laplacian_smoothing=MeshLaplacianSmoothing(sc["faces"], V, dev))
    before = g._points.detach().clone()
    for _ in range(2):
        out = stage.iteration()
        assert {"rgb", "mask", "sds", "normal_consistency", "laplacian_smoothing", "rgb_tv", "depth_tv", "normal_tv"} <= set(out)
        assert all(torch.isfinite(v) for v in out.values()), out
    assert all(torch.isfinite(p).all() for p in g.parameters())
    assert not torch.equal(before, g._points)


def test_config2_dynamic_16_frames_100k_gaussians_lbs():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import geometry as geo, ops, synthetic as syn, views, zero123 as z
    from dreammesh4d_amd.deformation import DeformationNetwork
    from dreammesh4d_amd.dynamic_stage import DynamicStage
    from dreammesh4d_amd.mesh_reg import ARAPCoach, MeshNormalConsistency

    dev = torch.device("cuda:0")
    H = W = 512
    M, L = 1000, 16
    sc = syn.mesh_bound_scene(16667, n_nodes=M, k=4, seed=0)
    T = lambda a: torch.tensor(a, device=dev)
    graph = ops.DeformGraph(sc["verts"], sc["nbr_idx"], sc["nbr_w"], M, dev)
    topo = ops.MeshTopology(sc["faces"], len(sc["verts"]), 6, dev)
    verts, faces = T(sc["verts"]), T(sc["faces"])
    static = {"q_static": geo.quaternions(verts, faces, T(sc["complex"]), 6), "scales": geo.scaling(T(sc["log_scales"]), syn.THICKNESS),
              "opacities": geo.strengths(T(sc["densities"])), "rgb": geo.points_rgb(T(sc["sh_dc"]))}
    assert abs(static["q_static"].shape[0] - 100002) < 100          # the seeded sphere tessellation rounds the face count
    cam = syn.make_camera(H, W, elev_deg=5.0, azim_deg=0.0)
    r = views.ViewRenderer(graph, topo, H, W, cam.tanfov, method="lbs")
    torch.manual_seed(0)
    net = DeformationNetwork(no_ds=False, no_dr=False, no_do=True).to(dev)          # lbs: no opacity head (dynamic_sugar.py:145)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if "_deform" in name:
                p.add_(0.01 * torch.randn_like(p))
    ts = torch.linspace(0, 1, L + 2, device=dev)[1:-1]
    guid = z.TemporalStableZero123Guidance(_small_zero123(L, dev), torch.randn(L, 1, 32), torch.randn(L, 4, 32, 32), cond_elevation_deg=5.0,
                                           half_precision_weights=False).to(dev)
    ref_img = torch.rand(L, H, W, 3, device=dev)
    ref_mask = (torch.rand(L, H, W, 1, device=dev) > 0.5).float()
    stage = DynamicStage(r, net, T(sc["nodes"]), static, ts, ref_img, ref_mask, cam, guidance=guid, frames_per_step=4, random_views_per_frame=1,
                         normal_consistency=MeshNormalConsistency(sc["faces"], len(sc["verts"]), dev),
                         arap=ARAPCoach(sc["verts"], sc["faces"], dev), milestone_arap_reg=0)
    for _ in range(2):
        out = stage.iteration()
        assert {"rgb", "mask", "sds", "normal_consistency", "arap_reg_key_frame"} <= set(out)
        assert all(torch.isfinite(v) for v in out.values()), out
    r.check()
    assert all(torch.isfinite(p).all() for p in net.parameters())


def test_config4_dense_stress_1m_gaussians_1024_dqs_fp16_unet():
    """BASELINE configs[4] (single-GPU share): ~1 M mesh-bound Gaussians (166,667 faces x 6), 1024^2, DQS skinning,
    fp16 Zero123 (reduced width, random weights).  (i) the batched path: 2 views forward + backward, no duplicate /
    record overflow, finite gradients, deterministic; (ii) one full dynamic-stage iteration with the DQS head layout
    (no strain / opacity heads, dynamic_sugar.py:144-145) and the SDS step in half precision.  Operator-level oracle
    parity at this size: tests/test_raster_gpu.py::test_cfg5_one_million_gaussians_1024_oracle_parity."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import geometry as geo, ops, synthetic as syn, views, zero123 as z
    from dreammesh4d_amd.deformation import DeformationNetwork
    from dreammesh4d_amd.dynamic_stage import DynamicStage

    dev = torch.device("cuda:0")
    F_, M, H, W, B, L = 166_667, 1000, 1024, 1024, 2, 8
    sc = syn.mesh_bound_scene(F_, n_nodes=M, k=4, seed=0)
    T = lambda a: torch.tensor(a, device=dev)
    # the deformation graph of the PRODUCT at this size (round 4): the reference's heat-method geodesics (`dist_mode: geodisc`,
    # geometry/dynamic_sugar.py:745-861) on 83.3k vertices, not synthetic.mesh_bound_scene's Euclidean stand-in
    from dreammesh4d_amd.graph_build import build_deformation_graph

    g_idx, g_w = build_deformation_graph(sc["verts"], sc["faces"], sc["nodes"], 4, "geodisc", dev)
    assert tuple(g_idx.shape) == (len(sc["verts"]), 4) and float((g_w.sum(1) - 1).abs().max()) < 1e-5
    agree = float((torch.sort(g_idx, 1).values.cpu() == torch.sort(torch.as_tensor(np.asarray(sc["nbr_idx"]), dtype=torch.int64), 1).values).all(1).float().mean())
    print(f"cfg 5 graph: heat-method and Euclidean neighbour sets agree on {agree:.4f} of the vertices")
    assert agree > 0.5, agree             # (on a sphere the chord is monotone in the arc: the two rankings differ only near ties)
    graph = ops.DeformGraph(sc["verts"], g_idx, g_w, M, dev)
    del g_idx, g_w
    torch.cuda.empty_cache()
    topo = ops.MeshTopology(sc["faces"], len(sc["verts"]), 6, dev)
    verts, faces = T(sc["verts"]), T(sc["faces"])
    static = {"q_static": geo.quaternions(verts, faces, T(sc["complex"]), 6), "scales": geo.scaling(T(sc["log_scales"]), syn.THICKNESS),
              "opacities": geo.strengths(T(sc["densities"])), "rgb": geo.points_rgb(T(sc["sh_dc"]))}
    N = static["q_static"].shape[0]
    assert abs(N - 1_000_002) < 2000, N
    ts, motion = syn.node_motion(M, B, seed=0)
    raw = {k: torch.stack([T(m[k]) for m in motion]).requires_grad_(True) for k in ("trans", "d_rot", "strain", "d_opacity")}
    cams = [syn.make_camera(H, W, elev_deg=10 + 20 * b, azim_deg=40 * b) for b in range(B)]
    vm = torch.stack([T(c.viewmatrix) for c in cams]); pm = torch.stack([T(c.projmatrix) for c in cams])
    r = views.ViewRenderer(graph, topo, H, W, cams[0].tanfov, method="dqs")
    grads = []
    for _ in range(2):
        for v in raw.values():
            v.grad = None
        out = views.render_views(r, raw["trans"], raw["d_rot"], raw["strain"], raw["d_opacity"].squeeze(-1), static["q_static"],
                                 static["scales"], static["opacities"], static["rgb"], vm, pm, torch.ones(6, device=dev))
        w = torch.linspace(0.5, 1.5, H * W, device=dev).view(1, 1, H, W)
        ((out["color"] * w).sum() + out["alpha"].sum() + (out["depth"] * w).sum()).backward()
        D = r.check()                                   # raises on duplicate / record overflow
        assert len(D) == B and all(1_000_000 < d <= r.capacity for d in D), (D, r.capacity)
        assert all(n <= r.record_capacity for n in r.last_num_records)
        grads.append({k: v.grad.clone() for k, v in raw.items() if v.grad is not None})
    assert set(grads[0]) == {"trans", "d_rot"}            # dqs: strain / opacity heads unused
    for k in grads[0]:
        assert torch.isfinite(grads[0][k]).all() and grads[0][k].abs().sum() > 0, k
        assert torch.equal(grads[0][k], grads[1][k]), k    # deterministic at 1 M Gaussians too
    cover = float((out["alpha"] > 0.5).float().mean())
    assert 0.3 < cover < 0.9, cover                       # the sphere fills the frame as at 512^2
    # (ii) one iteration of the stage, DQS head layout, fp16 SDS
    torch.manual_seed(0)
    net = DeformationNetwork(no_ds=True, no_dr=False, no_do=True).to(dev)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if "_deform" in name:
                p.add_(0.01 * torch.randn_like(p))
    model = z.Zero123(unet_kwargs=dict(model_channels=32, context_dim=32, num_heads=4), vae_kwargs=dict(ch=32))
    for p in model.model.diffusion_model.out.parameters():
        torch.nn.init.normal_(p, std=0.05)
    guid = z.TemporalStableZero123Guidance(model, torch.randn(L, 1, 32), torch.randn(L, 4, 32, 32), cond_elevation_deg=5.0,
                                           half_precision_weights=True).to(dev)
    assert next(guid.model.model.diffusion_model.parameters()).dtype == torch.float16
    tsL = torch.linspace(0, 1, L + 2, device=dev)[1:-1]
    stage = DynamicStage(r, net, T(sc["nodes"]), static, tsL, torch.rand(L, H, W, 3, device=dev),
                         (torch.rand(L, H, W, 1, device=dev) > 0.5).float(), syn.make_camera(H, W, elev_deg=5.0, azim_deg=0.0),
                         guidance=guid, frames_per_step=2, random_views_per_frame=1)
    res = stage.iteration()
    assert {"rgb", "mask", "sds"} <= set(res) and all(torch.isfinite(v) for v in res.values()), res
    r.check()
    assert all(torch.isfinite(p).all() for p in net.parameters())
