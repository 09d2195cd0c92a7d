"""Validation sweep (dreammesh4d_amd/validation.py): the frame-batched forward-only path must give the images the
training path's forward gives for the same (timestamp, camera) units."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_sweep_equals_per_frame_rendering():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import geometry as geo, ops, synthetic as syn, validation, views
    from dreammesh4d_amd.deformation import DeformationNetwork

    dev = torch.device("cuda:0")
    H = W = 96
    M, L = 60, 6
    sc = syn.mesh_bound_scene(1500, n_nodes=M, k=4, seed=5)
    T = lambda a: torch.tensor(a, device=dev)
    graph = ops.DeformGraph(sc["verts"], sc["nbr_idx"], sc["nbr_w"], M, dev)
    topo = ops.MeshTopology(sc["faces"], len(sc["verts"]), 6, dev)
    verts, faces = T(sc["verts"]), T(sc["faces"])
    static = {"q_static": geo.quaternions(verts, faces, T(sc["complex"]), 6),
              "scales": geo.scaling(T(sc["log_scales"]) + 1.0, syn.THICKNESS),
              "opacities": geo.strengths(T(sc["densities"])), "rgb": geo.points_rgb(T(sc["sh_dc"]))}
    cam = syn.make_camera(H, W)
    r = views.ViewRenderer(graph, topo, H, W, cam.tanfov, method="hybrid")
    torch.manual_seed(0)
    net = DeformationNetwork(resolution=(16, 16, 16, 9), multires=(1, 2), no_ds=False, no_dr=False, no_do=False).to(dev)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if "_deform" in name:
                p.add_(0.05 * torch.randn_like(p))
    nodes = T(sc["nodes"])
    ts = torch.linspace(0, 1, L + 2, device=dev)[1:-1]
    az = (0.0, 120.0, 240.0)
    got = validation.sweep(r, net, nodes, static, ts, azimuths_deg=az, frames_per_call=4)
    assert got["comp_rgb"].shape == (L, len(az), H, W, 3) and got["opacity"].shape == (L, len(az), H, W, 1)
    assert float(got["comp_rgb"].min()) >= 0.0 and float(got["comp_rgb"].max()) <= 1.0
    assert float(got["opacity"].max()) > 0.9 and float(got["opacity"][:, :, 0, 0].max()) == 0.0      # object in the middle, black corners
    # every unit separately, through the same operator one view at a time
    bg6 = torch.zeros(6, device=dev)
    with torch.no_grad():
        for f in (0, 3, 5):
            dx, dr, ds, do = net.node_outputs(nodes, ts[f:f + 1])
            for a, azim in enumerate(az):
                c = syn.make_camera(H, W, elev_deg=0.0, azim_deg=azim)
                o = views.render_views(r, dx, dr, ds, do, static["q_static"], static["scales"], static["opacities"], static["rgb"],
                                       T(c.viewmatrix)[None], T(c.projmatrix)[None], bg6)
                assert torch.equal(o["color"][0, :3].clamp(0, 1).permute(1, 2, 0), got["comp_rgb"][f, a])
                assert torch.equal(o["depth"][0].permute(1, 2, 0), got["depth"][f, a])
    # one unit against the CPU oracle: frame 3 from azimuth 120.  Two stages, each at its own bar (the round-2 version compared
    # the image of float32 HIP skinning with the image of float64 oracle skinning at 2e-3: a pixel-level decision flips where a
    # vertex moves by 1e-7).  (i) the rasterizer: the oracle blends the HIP path's OWN Gaussians -- the forward contract is
    # bit-identity, north_star's bar is 1e-4; (ii) skinning + face->Gaussian: float32 kernels against the float64 oracle.
    from oracle import raster as orc, skinning as sk

    D = torch.float64
    tt = lambda a: torch.tensor(np.asarray(a), dtype=D)
    f, a = 3, 1
    c = syn.make_camera(H, W, elev_deg=0.0, azim_deg=az[a])
    with torch.no_grad():
        dx, dr, ds, do = net.node_outputs(nodes, ts[f:f + 1])
        unit = views.render_views(r, dx, dr, ds, do, static["q_static"], static["scales"], static["opacities"], static["rgb"],
                                  T(c.viewmatrix)[None], T(c.projmatrix)[None], bg6)
        hm, hq, hn = ops.face_gaussians(topo, unit["vxyz"][0], unit["vrot"][0], static["q_static"])
    o = orc.RasterOracle(image_height=H, image_width=W, tanfovx=c.tanfov, tanfovy=c.tanfov, bg=(0, 0, 0), scale_modifier=1.0,
                         viewmatrix=c.viewmatrix, projmatrix=c.projmatrix, campos=c.campos)
    o.forward(hm.cpu().numpy(), static["opacities"].view(-1).cpu().numpy(), colors_precomp=static["rgb"].cpu().numpy(),
              scales=static["scales"].cpu().numpy(), rotations=hq.cpu().numpy())
    want = np.clip(np.moveaxis(o.s["out_color"], 0, -1), 0, 1)
    mine = got["comp_rgb"][f, a].cpu().numpy()
    assert np.abs(mine - want).max() <= 1e-4 and np.array_equal(mine.view(np.uint32), want.view(np.uint32))      # (in fact bit-identical)
    assert np.array_equal(got["opacity"][f, a, :, :, 0].cpu().numpy().view(np.uint32), o.s["out_alpha"].view(np.uint32))
    assert (o.s["out_alpha"] > 0.5).mean() > 0.1
    o2 = orc.RasterOracle(image_height=H, image_width=W, tanfovx=c.tanfov, tanfovy=c.tanfov, bg=(0, 0, 0), scale_modifier=1.0,
                          viewmatrix=c.viewmatrix, projmatrix=c.projmatrix, campos=c.campos)
    o2.forward(hm.cpu().numpy(), static["opacities"].view(-1).cpu().numpy(), colors_precomp=hn.cpu().numpy(),
               scales=static["scales"].cpu().numpy(), rotations=hq.cpu().numpy())
    nrm = torch.nn.functional.normalize(torch.tensor(np.moveaxis(o2.s["out_color"], 0, -1)), dim=-1).numpy()
    want_n = nrm * 0.5 * o.s["out_alpha"][..., None] + 0.5
    assert np.abs(got["comp_normal"][f, a].cpu().numpy() - want_n).max() <= 1e-4
    # (ii) the deformation stage of the same unit, float32 on the device against float64 on the host
    trans, q, S, op = sk.node_attributes(dx[0].cpu().to(D), dr[0].cpu().to(D), ds[0].cpu().to(D), do[0].cpu().to(D).reshape(M, -1))
    xyz, vrot = sk.skin_vertices(tt(sc["verts"]), torch.tensor(sc["nbr_idx"]), tt(sc["nbr_w"]), trans, q, S, op, "hybrid")
    assert np.abs(unit["vxyz"][0].cpu().numpy() - xyz.numpy()).max() < 2e-6 and np.abs(unit["vrot"][0].cpu().numpy() - vrot.numpy()).max() < 2e-6
    qs64 = sk.static_quaternions(tt(sc["verts"]), torch.tensor(sc["faces"]), tt(sc["complex"]))
    om, oq, on = sk.face_gaussians(unit["vxyz"][0].cpu().to(D), unit["vrot"][0].cpu().to(D), torch.tensor(sc["faces"]), qs64)
    assert np.abs(hm.cpu().numpy() - om.numpy()).max() < 2e-6 and np.abs(hq.cpu().numpy() - oq.numpy()).max() < 4e-6
    # streaming variant hands the same chunks to the callback
    seen = []
    assert validation.sweep(r, net, nodes, static, ts, azimuths_deg=az, frames_per_call=4,
                            on_chunk=lambda fr, ch: seen.append((fr, ch["comp_rgb"].clone()))) is None
    assert [fr for fr, _ in seen] == [[0, 1, 2, 3], [4, 5]] and torch.equal(torch.cat([c for _, c in seen]), got["comp_rgb"])
