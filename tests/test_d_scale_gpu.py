"""-m gpu: the reference's `d_scale: true` branch (C/geometry/dynamic_sugar.py:593-611 vertex scale matrices, :697-704 Gaussian
scales, :717-720 the scales handed to the rasterizer): the device operators against the float64 restatement in
oracle/skinning.py (values and gradients), per-frame scales through the batched-view path against the per-view operator, and
the `dynamic-sugar` plugin constructed with `d_scale: true`."""
import numpy as np
import pytest
import torch

from dreammesh4d_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")


@pytest.mark.parametrize("method", ["lbs", "hybrid"])
def test_vertex_and_gaussian_scales_against_the_oracle(method):
    _need_gpu()
    from dreammesh4d_amd import geometry as geo, ops
    from oracle import skinning as sk

    dev = torch.device("cuda:0")
    M, NF = 80, 3
    sc = syn.mesh_bound_scene(900, n_nodes=M, k=4, seed=2)
    graph = ops.DeformGraph(sc["verts"], sc["nbr_idx"], sc["nbr_w"], M, dev)
    topo = ops.MeshTopology(sc["faces"], len(sc["verts"]), 6, dev)
    g = torch.Generator().manual_seed(1)
    ds = (0.2 * torch.randn(NF, M, 6, generator=g)).to(dev).requires_grad_(True)
    do = torch.randn(NF, M, generator=g).to(dev).requires_grad_(True)
    scaling = geo.scaling(torch.tensor(sc["log_scales"], device=dev), syn.THICKNESS).clone().requires_grad_(True)
    Sv = ops.vertex_scale_matrices(graph, ds, do if method == "hybrid" else None, method)
    gs = ops.gaussian_scales(topo, Sv, scaling)
    assert Sv.shape == (NF, graph.V, 3, 3) and gs.shape == (NF, topo.F * 6, 3)
    gS = torch.randn(Sv.shape, generator=g).to(dev)
    gG = torch.randn(gs.shape, generator=g).to(dev)
    torch.autograd.backward([Sv, gs], [gS, gG])
    D = torch.float64
    idx, w, faces = torch.tensor(sc["nbr_idx"]), torch.tensor(sc["nbr_w"], dtype=D), torch.tensor(sc["faces"])
    ds64, do64 = ds.detach().cpu().to(D).requires_grad_(True), do.detach().cpu().to(D).requires_grad_(True)
    sc64 = scaling.detach().cpu().to(D).requires_grad_(True)
    Sv64, gs64 = [], []
    for f in range(NF):
        _, _, S, op = sk.node_attributes(torch.zeros(M, 3, dtype=D), torch.zeros(M, 4, dtype=D), ds64[f], do64[f].reshape(M, 1))
        v = sk.vertex_scales(idx, w, S, op, method)
        Sv64.append(v)
        gs64.append(sk.gaussian_scales(faces, 6, v, sc64))
    Sv64, gs64 = torch.stack(Sv64), torch.stack(gs64)
    torch.autograd.backward([Sv64, gs64], [gS.cpu().to(D), gG.cpu().to(D)])
    assert float((Sv.detach().cpu().to(D) - Sv64).abs().max()) < 2e-6
    assert float((gs.detach().cpu().to(D) - gs64).abs().max()) < 2e-6 * float(gs64.abs().max()) + 1e-9
    for a, b in ((ds.grad, ds64.grad), (scaling.grad, sc64.grad)) + (((do.grad, do64.grad),) if method == "hybrid" else ()):
        assert float((a.cpu().to(D) - b).abs().max()) <= 2e-5 * float(b.abs().max())
    # no strain: every vertex matrix is a multiple of the identity (lbs: exactly I) and the Gaussians keep their scaling
    z = torch.zeros(1, M, 6, device=dev)
    Sv0 = ops.vertex_scale_matrices(graph, z, torch.zeros(1, M, device=dev), method)
    off = Sv0 - torch.diag_embed(torch.diagonal(Sv0, dim1=-2, dim2=-1))
    assert float(off.abs().max()) == 0.0
    if method == "lbs":
        assert torch.allclose(ops.gaussian_scales(topo, Sv0, scaling.detach())[0], scaling.detach(), rtol=1e-6, atol=0)
    with pytest.raises(ValueError):
        ops.vertex_scale_matrices(graph, z, None, "dqs")


def test_per_frame_scales_through_the_batched_views():
    """views.render_views with scales [n_frames, N, 3]: every view renders its frame's scales (bit-identical to the per-view
    operator fed those scales) and the gradient comes back per frame, summed over the frame's views."""
    _need_gpu()
    from dreammesh4d_amd import geometry as geo, ops, views
    from tests.hip_raster import HipRaster

    dev = torch.device("cuda:0")
    B, NF, H, W, M = 4, 2, 128, 160, 60
    sc = syn.mesh_bound_scene(1500, n_nodes=M, k=4, seed=3)
    T = lambda a: torch.tensor(a, device=dev)
    graph = ops.DeformGraph(sc["verts"], sc["nbr_idx"], sc["nbr_w"], M, dev)
    topo = ops.MeshTopology(sc["faces"], len(sc["verts"]), 6, dev)
    verts, faces = T(sc["verts"]), T(sc["faces"])
    qs = geo.quaternions(verts, faces, T(sc["complex"]), 6)
    scaling = geo.scaling(T(sc["log_scales"]) + 1.0, syn.THICKNESS)
    opac, rgb = geo.strengths(T(sc["densities"])), geo.points_rgb(T(sc["sh_dc"]))
    ts, motion = syn.node_motion(M, NF, seed=3)
    raw = {k: torch.stack([T(m[k]) for m in motion]) for k in ("trans", "d_rot", "strain", "d_opacity")}
    ds = (raw["strain"] * 4.0).clone().requires_grad_(True)
    do = raw["d_opacity"].squeeze(-1).clone().requires_grad_(True)
    fidx = torch.tensor([0, 1, 1, 0], device=dev, dtype=torch.int32)
    cams = [syn.make_camera(H, W, elev_deg=10 + 9 * b, azim_deg=-100 + 70 * b) for b in range(B)]
    vm, pm = torch.stack([T(c.viewmatrix) for c in cams]), torch.stack([T(c.projmatrix) for c in cams])
    r = views.ViewRenderer(graph, topo, H, W, cams[0].tanfov, method="hybrid")
    scales = ops.gaussian_scales(topo, ops.vertex_scale_matrices(graph, ds, do, "hybrid"), scaling)      # [NF, N, 3]
    scales.retain_grad()
    out = views.render_views(r, raw["trans"], raw["d_rot"], ds, do, qs, scales, opac, rgb, vm, pm, torch.ones(6, device=dev),
                             frame_index=fidx)
    r.check()
    gen = torch.Generator().manual_seed(0)
    gC = torch.randn(B, 6, H, W, generator=gen).to(dev)
    gA = torch.randn(B, 1, H, W, generator=gen).to(dev)
    torch.autograd.backward([out["color"], out["alpha"]], [gC, gA])
    assert float((scales[0] - scales[1]).abs().max()) > 1e-6                   # the two frames really differ
    want = torch.zeros_like(scales)
    for b in range(B):
        f = int(fidx[b])
        xyz, vrot = ops.skin_vertices(graph, raw["trans"][f], raw["d_rot"][f], ds[f].detach(), do[f].detach(), "hybrid")
        means, rots, normals = ops.face_gaussians(topo, xyz, vrot, qs)
        h = HipRaster(cams[b], bg=(1, 1, 1, 1, 1, 1))
        color, radii, depth, alpha = h.forward(means.cpu().numpy(), opac.view(-1).cpu().numpy(),
                                               colors=torch.cat([rgb, normals], dim=1).cpu().numpy(),
                                               scales=scales[f].detach().cpu().numpy(), rotations=rots.cpu().numpy())
        assert np.array_equal(out["color"][b].detach().cpu().numpy().view(np.uint32), color.view(np.uint32))
        assert np.array_equal(out["alpha"][b, 0].detach().cpu().numpy().view(np.uint32), alpha.view(np.uint32))
        g = h.backward(gC[b].cpu().numpy(), None, gA[b, 0].cpu().numpy())
        want[f] += torch.tensor(g["dL_dscales"], device=dev)
    assert float((scales.grad - want).abs().max()) <= 1e-5 * float(want.abs().max())
    assert ds.grad is not None and float(ds.grad.abs().max()) > 0 and torch.isfinite(ds.grad).all()
    with pytest.raises(ValueError):
        views.render_views(r, raw["trans"], raw["d_rot"], ds, do, qs, scales[:, :100], opac, rgb, vm, pm, torch.ones(6, device=dev),
                           frame_index=fidx)


def test_dynamic_sugar_plugin_with_d_scale(tmp_path):
    _need_gpu()
    import tests.test_plugins_from_cfg_gpu as P
    from dreammesh4d_amd import threestudio_host as ts

    dev = torch.device("cuda:0")
    mesh, _, _ = P._stand_ins(tmp_path, 1, dev)
    cfg = ts.resolve({"data": P.DATA, "system": P.DYNAMIC_SYSTEM})["system"]
    geo_cfg = dict(cfg["geometry"], surface_mesh_to_bind_path=mesh, d_scale=True)
    geometry = ts.find("dynamic-sugar")(geo_cfg)
    with pytest.raises(ValueError):
        ts.find("dynamic-sugar")(dict(geo_cfg, skinning_method="dqs"))
    renderer = ts.find("diff-sugar-rasterizer-temporal")(cfg["renderer"], geometry=geometry, material=ts.find("no-material")({"n_output_dims": 0}),
                                                         background=ts.find("solid-color-background")(None))
    with torch.no_grad():
        for n, p in geometry._deformation.named_parameters():
            if "_deform" in n:
                p.add_(0.02 * torch.randn_like(p))
    B, H, W = 2, 160, 160
    t = torch.tensor([0.3, 0.7], device=dev)
    out = renderer.batch_forward(P._batch(B, H, W, dev, timestamps=t))
    a = geometry.get_timed_gs_attributes(t)
    assert a["scale"].shape == (2, geometry.n_gaussians, 3) and float((a["scale"][0] - geometry.get_scaling).abs().max()) > 0
    m3, s3, r3, o3, c3 = geometry.get_timed_gs_all_single_time(t[:1])
    assert torch.allclose(s3, a["scale"][0])
    (out["comp_rgb"].mean() + out["comp_mask"].mean()).backward()
    scale_head = [p for n, p in geometry._deformation.named_parameters() if "scales_deform" in n]
    assert scale_head and all(p.grad is not None and torch.isfinite(p.grad).all() for p in scale_head)
    assert any(float(p.grad.abs().max()) > 0 for p in scale_head)
