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
