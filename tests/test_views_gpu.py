"""GPU parity of the batched fast path (dm4d_views_forward/backward) against the composition of
the per-call operators (which are themselves checked against the oracle): identical kernels, so
the images and gradients must agree bit-for-bit; plus a direct oracle check of one view."""
import math

import numpy as np
import pytest
import torch

from dreammesh4d_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")


def _scene(n_faces, M, K, B, H, W, dev, seed=0):
    from dreammesh4d_amd import geometry as geo, ops

    sc = syn.mesh_bound_scene(n_faces, n_nodes=M, k=K, seed=seed)
    V, F = len(sc["verts"]), len(sc["faces"])
    graph = ops.DeformGraph(sc["verts"], sc["nbr_idx"], sc["nbr_w"], M, dev)
    topo = ops.MeshTopology(sc["faces"], V, 6, dev)
    T = lambda a: torch.tensor(a, device=dev)
    verts, faces = T(sc["verts"]), T(sc["faces"])
    qs = geo.quaternions(verts, faces, T(sc["complex"]), 6)
    scales = geo.scaling(T(sc["log_scales"]), syn.THICKNESS)
    opac = geo.strengths(T(sc["densities"]))
    rgb = geo.points_rgb(T(sc["sh_dc"]))
    ts, motion = syn.node_motion(M, B, seed=seed)
    raw = {k: torch.stack([T(m[k]) for m in motion]) for k in ("trans", "d_rot", "strain", "d_opacity")}
    cams = [syn.make_camera(H, W, elev_deg=10 + 7 * b, azim_deg=-120 + 67 * b) for b in range(B)]
    vm = torch.stack([T(c.viewmatrix) for c in cams])
    pm = torch.stack([T(c.projmatrix) for c in cams])
    return sc, graph, topo, qs, scales, opac, rgb, raw, cams, vm, pm


def test_batched_views_equal_per_call_composition():
    _need_gpu()
    from dreammesh4d_amd import ops, views
    from tests.hip_raster import HipRaster

    dev = torch.device("cuda:0")
    B, H, W, M = 3, 160, 208, 120
    sc, graph, topo, qs, scales, opac, rgb, raw, cams, vm, pm = _scene(3000, M, 4, B, H, W, dev)
    r = views.ViewRenderer(graph, topo, H, W, cams[0].tanfov, method="hybrid")
    leaves = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    bg6 = torch.ones(6, device=dev)
    out = views.render_views(r, leaves["trans"], leaves["d_rot"], leaves["strain"], leaves["d_opacity"].squeeze(-1), qs,
                             scales, opac, rgb, vm, pm, bg6)
    nr = r.check()
    gen = torch.Generator().manual_seed(0)
    gC = torch.randn(B, 6, H, W, generator=gen).to(dev)
    gD = (0.1 * torch.randn(B, 1, H, W, generator=gen)).to(dev)
    gA = torch.randn(B, 1, H, W, generator=gen).to(dev)
    gV = (0.01 * torch.randn(B, graph.V, 3, generator=gen)).to(dev)
    torch.autograd.backward([out["color"], out["depth"], out["alpha"], out["vxyz"]], [gC, gD, gA, gV])
    for b in range(B):
        l2 = {k: raw[k][b].clone().requires_grad_(True) for k in raw}
        xyz, vrot = ops.skin_vertices(graph, l2["trans"], l2["d_rot"], l2["strain"], l2["d_opacity"].view(-1), "hybrid", grad_mode=r.grad_mode)
        means, rots, normals = ops.face_gaussians(topo, xyz, vrot, qs, grad_mode=r.grad_mode)
        assert torch.equal(out["vxyz"][b], xyz) and torch.equal(out["vrot"][b], vrot)
        h = HipRaster(cams[b], bg=(1, 1, 1, 1, 1, 1))
        col6 = torch.cat([rgb, normals.detach()], dim=1).cpu().numpy()
        color, radii, depth, alpha = h.forward(means.detach().cpu().numpy(), opac.view(-1).cpu().numpy(), colors=col6,
                                               scales=scales.cpu().numpy(), rotations=rots.detach().cpu().numpy())
        assert h.D == nr[b]
        assert np.array_equal(out["color"][b].detach().cpu().numpy().view(np.uint32), color.view(np.uint32))
        assert np.array_equal(out["depth"][b, 0].detach().cpu().numpy().view(np.uint32), depth.view(np.uint32))
        assert np.array_equal(out["alpha"][b, 0].detach().cpu().numpy().view(np.uint32), alpha.view(np.uint32))
        assert np.array_equal(out["radii"][b].cpu().numpy(), radii)
        g = h.backward(gC[b].cpu().numpy(), gD[b, 0].cpu().numpy(), gA[b, 0].cpu().numpy())
        t = lambda a: torch.tensor(a, device=dev)
        torch.autograd.backward([means, rots, normals, xyz],
                                [t(g["dL_dmeans3D"]), t(g["dL_drots"]), t(g["dL_dcolors"][:, 3:]).contiguous(), gV[b]])
        for k in raw:
            # same kernels; only the position where the external vertex gradient is added differs
            a, c = leaves[k].grad[b], l2[k].grad
            assert (a - c).abs().max() <= 1e-5 * c.abs().max(), (k, b)


def test_views_sharing_a_frame_equal_per_view_inputs():
    """frame_index: views of one timestamp share skinning + face transform; images are bit-identical to passing
    the frame's node outputs once per view, node gradients equal the sum over the frame's views."""
    _need_gpu()
    from dreammesh4d_amd import views

    dev = torch.device("cuda:0")
    B, H, W, M, NF = 5, 128, 160, 90, 2
    sc, graph, topo, qs, scales, opac, rgb, raw, cams, vm, pm = _scene(2500, M, 4, B, H, W, dev, seed=4)
    fidx = torch.tensor([0, 1, 1, 0, 1], device=dev)
    keys = ("trans", "d_rot", "strain", "d_opacity")
    fr = {k: raw[k][:NF].clone().requires_grad_(True) for k in keys}           # per-frame node outputs
    pv = {k: raw[k][:NF].clone().requires_grad_(True) for k in keys}
    bg6 = torch.ones(6, device=dev)
    gen = torch.Generator().manual_seed(3)
    gC = torch.randn(B, 6, H, W, generator=gen).to(dev)
    gA = torch.randn(B, 1, H, W, generator=gen).to(dev)
    r1 = views.ViewRenderer(graph, topo, H, W, cams[0].tanfov, method="hybrid")
    o1 = views.render_views(r1, fr["trans"], fr["d_rot"], fr["strain"], fr["d_opacity"].squeeze(-1), qs, scales, opac, rgb,
                            vm, pm, bg6, frame_index=fidx)
    assert o1["vxyz"].shape[0] == NF
    torch.autograd.backward([o1["color"], o1["alpha"]], [gC, gA])
    r2 = views.ViewRenderer(graph, topo, H, W, cams[0].tanfov, method="hybrid")
    o2 = views.render_views(r2, pv["trans"][fidx], pv["d_rot"][fidx], pv["strain"][fidx], pv["d_opacity"].squeeze(-1)[fidx],
                            qs, scales, opac, rgb, vm, pm, bg6)
    torch.autograd.backward([o2["color"], o2["alpha"]], [gC, gA])
    for k in ("color", "depth", "alpha", "radii"):
        assert torch.equal(o1[k], o2[k]), k
    assert torch.equal(o1["vxyz"][fidx], o2["vxyz"])
    for k in keys:
        a, c = fr[k].grad, pv[k].grad
        assert (a - c).abs().max() <= 2e-5 * c.abs().max(), k


@pytest.mark.parametrize("H,W", [(144, 176), (40, 48)])
def test_frozen_static_appearance_uses_lean_records_with_identical_gradients(H, W):
    """With scales / opacities / rgb frozen (the reference's dynamic stage) the blend backward keeps 9 of the 13 values
    per record; every gradient that is still produced must be bit-identical to the full backward's.  The 40x48 image
    crowds the 14,400 splats into a few cells, so most of them are LONG cells (the long-cell blocks of k_render_bwd, both variants)."""
    _need_gpu()
    from dreammesh4d_amd import views

    dev = torch.device("cuda:0")
    B, M = 3, 100
    sc, graph, topo, qs, scales, opac, rgb, raw, cams, vm, pm = _scene(2400, M, 4, B, H, W, dev, seed=2)
    gen = torch.Generator().manual_seed(1)
    gC = torch.randn(B, 6, H, W, generator=gen).to(dev)
    gD = (0.1 * torch.randn(B, 1, H, W, generator=gen)).to(dev)
    gA = torch.randn(B, 1, H, W, generator=gen).to(dev)
    res = {}
    for mode in ("frozen", "learnable"):
        r = views.ViewRenderer(graph, topo, H, W, cams[0].tanfov, method="hybrid")
        leaves = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
        st = [t.clone().requires_grad_(mode == "learnable") for t in (scales, opac, rgb)]
        m2 = torch.zeros(B, r.N, 3, device=dev, requires_grad=True)
        out = views.render_views(r, leaves["trans"], leaves["d_rot"], leaves["strain"], leaves["d_opacity"].squeeze(-1), qs,
                                 st[0], st[1], st[2], vm, pm, torch.ones(6, device=dev), means2D=m2)
        r.check()
        torch.autograd.backward([out["color"], out["depth"], out["alpha"]], [gC, gD, gA])
        res[mode] = {k: v.grad.clone() for k, v in leaves.items()}
        res[mode]["m2"] = m2.grad.clone()
        res[mode]["normal_grad"] = r.last_grads["col"][:, :, 3:].clone()
        res[mode]["rgb_grad"] = r.last_grads["col"][:, :, :3].clone()
        res[mode]["static"] = [t.grad for t in st]
    assert all(g is None for g in res["frozen"]["static"]) and all(g is not None for g in res["learnable"]["static"])
    assert float(res["learnable"]["rgb_grad"].abs().max()) > 0 and float(res["frozen"]["rgb_grad"].abs().max()) == 0
    for k in ("trans", "d_rot", "strain", "d_opacity", "m2", "normal_grad"):
        assert float(res["frozen"][k].abs().max()) > 0, k
        assert torch.equal(res["frozen"][k], res["learnable"][k]), k


def test_steps_do_not_leak_device_memory():
    """Outputs kept as plain ctx attributes once formed an uncollectable output -> grad_fn -> ctx cycle that
    leaked every step's graph; device memory must be flat from the second step on."""
    _need_gpu()
    from dreammesh4d_amd import views

    dev = torch.device("cuda:0")
    B, H, W, M = 2, 96, 96, 60
    sc, graph, topo, qs, scales, opac, rgb, raw, cams, vm, pm = _scene(1200, M, 4, B, H, W, dev, seed=5)
    r = views.ViewRenderer(graph, topo, H, W, cams[0].tanfov)
    leaves = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    used = []
    for _ in range(6):
        out = views.render_views(r, leaves["trans"], leaves["d_rot"], leaves["strain"], leaves["d_opacity"].squeeze(-1), qs,
                                 scales, opac, rgb, vm, pm, torch.ones(6, device=dev))
        (out["color"].sum() + out["alpha"].sum()).backward()
        for v in leaves.values():
            v.grad = None
        del out
        torch.cuda.synchronize()
        used.append(torch.cuda.memory_allocated())
    assert max(used[2:]) - min(used[2:]) < (1 << 20), used


def test_batched_view_against_oracle():
    _need_gpu()
    from dreammesh4d_amd import views
    from oracle import raster as orc, skinning as sk

    dev = torch.device("cuda:0")
    B, H, W, M = 2, 128, 128, 90
    sc, graph, topo, qs, scales, opac, rgb, raw, cams, vm, pm = _scene(1800, M, 4, B, H, W, dev, seed=4)
    r = views.ViewRenderer(graph, topo, H, W, cams[0].tanfov, method="hybrid")
    out = views.render_views(r, raw["trans"], raw["d_rot"], raw["strain"], raw["d_opacity"].squeeze(-1), qs, scales, opac,
                             rgb, vm, pm, torch.ones(6, device=dev))
    r.check()
    D = torch.float64
    b = 1
    tt = lambda a: torch.tensor(np.asarray(a), dtype=D)
    trans, q, S, op = sk.node_attributes(*(raw[k][b].cpu().to(D) for k in ("trans", "d_rot", "strain", "d_opacity")))
    xyz, vrot = sk.skin_vertices(tt(sc["verts"]), torch.tensor(sc["nbr_idx"]), tt(sc["nbr_w"]), trans, q, S, op, "hybrid")
    qs64 = sk.static_quaternions(tt(sc["verts"]), torch.tensor(sc["faces"]), tt(sc["complex"]))
    assert np.abs(qs64.numpy() - qs.cpu().numpy()).max() < 2e-5
    means, rots, normals = sk.face_gaussians(xyz, vrot, torch.tensor(sc["faces"]), qs64)
    assert np.abs(out["vxyz"][b].cpu().numpy() - xyz.numpy()).max() < 2e-6
    cam = cams[b]
    o = orc.RasterOracle(image_height=H, image_width=W, tanfovx=cam.tanfov, tanfovy=cam.tanfov, bg=(1, 1, 1),
                         scale_modifier=1.0, viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix, campos=cam.campos)
    o.forward(means.float().numpy(), opac.view(-1).cpu().numpy(), colors_precomp=rgb.cpu().numpy(),
              scales=scales.cpu().numpy(), rotations=rots.float().numpy())
    # float32 skinning vs float64 oracle skinning feeding a float32 rasterizer: a few 1e-5 in the image
    assert np.abs(out["color"][b, :3].cpu().numpy() - o.s["out_color"]).max() < 2e-3
    assert np.abs(out["alpha"][b, 0].cpu().numpy() - o.s["out_alpha"]).max() < 2e-3
    assert (o.s["out_alpha"] > 0.5).mean() > 0.2


def test_capacity_overflow_reported():
    _need_gpu()
    from dreammesh4d_amd import _lib, views

    dev = torch.device("cuda:0")
    B, H, W, M = 2, 96, 96, 60
    sc, graph, topo, qs, scales, opac, rgb, raw, cams, vm, pm = _scene(1200, M, 4, B, H, W, dev, seed=2)
    r = views.ViewRenderer(graph, topo, H, W, cams[0].tanfov)
    r.capacity = 500
    r.calibrated = True          # skip the first-call calibration: this test wants the lazy check() path
    views.render_views(r, raw["trans"], raw["d_rot"], raw["strain"], raw["d_opacity"].squeeze(-1), qs, scales, opac, rgb,
                       vm, pm, torch.ones(6, device=dev))
    with pytest.raises(_lib.Dm4dError):
        r.check()
    assert r.capacity > 500
    views.render_views(r, raw["trans"], raw["d_rot"], raw["strain"], raw["d_opacity"].squeeze(-1), qs, scales, opac, rgb,
                       vm, pm, torch.ones(6, device=dev))
    assert min(r.check()) > 500
    # the first call of a fresh renderer calibrates both capacities by itself (one sync), records included
    r2 = views.ViewRenderer(graph, topo, H, W, cams[0].tanfov)
    r2.capacity, r2.record_capacity = 500, 100
    out = views.render_views(r2, raw["trans"], raw["d_rot"], raw["strain"], raw["d_opacity"].squeeze(-1), qs, scales, opac,
                             rgb, vm, pm, torch.ones(6, device=dev))
    assert r2.calibrated and r2.capacity > 500 and r2.record_capacity >= max(r2.last_num_records)
    ref = views.render_views(r, raw["trans"], raw["d_rot"], raw["strain"], raw["d_opacity"].squeeze(-1), qs, scales, opac,
                             rgb, vm, pm, torch.ones(6, device=dev))
    assert torch.equal(out["color"], ref["color"])


@pytest.mark.parametrize("H,W,shared", [(144, 176, True), (40, 48, False)])
def test_fused_gather_and_face_backward_equals_the_two_kernel_path(H, W, shared):
    """`fuse_face_backward`: B2 and the face part of the face->Gaussian backward as ONE kernel (csrc/gather_face.hip), nothing
    materialised per view; default: the two-kernel path.  Same code per (view, Gaussian); round 4: the fused kernel keeps B2's
    thread per (view, Gaussian) and the views of a frame are added when the vertex kernel sums the corner records (the
    backward is linear) instead of before the face's finish -- the node gradients, the external vertex gradient path and the
    optional screen-space gradient agree to float32 rounding of that reordered sum (bit for bit where a frame has one view and
    for the screen-space gradient, which B2 writes itself)."""
    _need_gpu()
    from dreammesh4d_amd import views

    dev = torch.device("cuda:0")
    B, M = 4, 100
    sc, graph, topo, qs, scales, opac, rgb, raw, cams, vm, pm = _scene(2400, M, 4, B, H, W, dev, seed=5)
    fidx = torch.tensor([0, 1, 1, 0], device=dev, dtype=torch.int32) if shared else None
    NF = 2 if shared else B
    gen = torch.Generator().manual_seed(2)
    gC = torch.randn(B, 6, H, W, generator=gen).to(dev)
    gD = (0.1 * torch.randn(B, 1, H, W, generator=gen)).to(dev)
    gA = torch.randn(B, 1, H, W, generator=gen).to(dev)
    gV = (0.01 * torch.randn(NF, graph.V, 3, generator=gen)).to(dev)
    res = []
    for keep in (False, True):
        for want_m2 in (False, True):
            r = views.ViewRenderer(graph, topo, H, W, cams[0].tanfov, method="hybrid")
            r.fuse_face_backward = not keep
            leaves = {k: v[:NF].clone().requires_grad_(True) for k, v in raw.items()}
            m2 = torch.zeros(B, r.N, 3, device=dev, requires_grad=True) if want_m2 else None
            out = views.render_views(r, leaves["trans"], leaves["d_rot"], leaves["strain"], leaves["d_opacity"].squeeze(-1), qs, scales,
                                     opac, rgb, vm, pm, torch.ones(6, device=dev), frame_index=fidx, means2D=m2)
            torch.autograd.backward([out["color"], out["depth"], out["alpha"], out["vxyz"]], [gC, gD, gA, gV])
            assert (r.last_grads["m3"] is not None) == keep
            res.append(({k: v.grad.clone() for k, v in leaves.items()}, None if m2 is None else m2.grad.clone()))
    # res: [fused, fused + m2, unfused, unfused + m2]
    for k in res[0][0]:
        assert torch.equal(res[0][0][k], res[1][0][k]) and torch.equal(res[2][0][k], res[3][0][k]), k      # asking for m2 changes nothing
        a, b = res[0][0][k], res[2][0][k]
        assert float(b.abs().max()) > 0, k
        if shared:
            assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), (k, float((a - b).abs().max()), float(b.abs().max()))
        else:
            assert torch.equal(a, b), k          # one view per frame: nothing is reordered
    assert torch.equal(res[1][1], res[3][1]) and float(res[1][1].abs().max()) > 0


@pytest.mark.parametrize("H,W", [(144, 176), (40, 48)])
def test_no_depth_gradient_uses_32_byte_records_with_identical_gradients(H, W):
    """The shipped dynamic configuration has no depth loss (configs/sugar_dynamic_dg.yaml:142-154: lambda_depth,
    lambda_depth_rel, lambda_depth_tv = 0), so autograd hands the renderer NO gradient for the depth image: with the static
    appearance frozen the blend backward then keeps 8 values per record (32 bytes, two 16-byte pieces instead of three).  Every
    gradient must be bit-identical to the 48-byte records fed with an all-zero depth gradient (a zero adds nothing to any sum)."""
    _need_gpu()
    from dreammesh4d_amd import views

    dev = torch.device("cuda:0")
    B, M = 3, 100
    sc, graph, topo, qs, scales, opac, rgb, raw, cams, vm, pm = _scene(2400, M, 4, B, H, W, dev, seed=2)
    gen = torch.Generator().manual_seed(1)
    gC = torch.randn(B, 6, H, W, generator=gen).to(dev)
    gA = torch.randn(B, 1, H, W, generator=gen).to(dev)
    res = {}
    for mode in ("none", "zeros"):
        r = views.ViewRenderer(graph, topo, H, W, cams[0].tanfov, method="hybrid")
        leaves = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
        m2 = torch.zeros(B, r.N, 3, device=dev, requires_grad=True)
        out = views.render_views(r, leaves["trans"], leaves["d_rot"], leaves["strain"], leaves["d_opacity"].squeeze(-1), qs,
                                 scales, opac, rgb, vm, pm, torch.ones(6, device=dev), means2D=m2)
        r.check()
        if mode == "none":
            torch.autograd.backward([out["color"], out["alpha"]], [gC, gA])
        else:
            torch.autograd.backward([out["color"], out["depth"], out["alpha"]], [gC, torch.zeros(B, 1, H, W, device=dev), gA])
        res[mode] = {k: v.grad.clone() for k, v in leaves.items()}
        res[mode]["m2"] = m2.grad.clone()
        res[mode]["normal_grad"] = r.last_grads["col"][:, :, 3:].clone()
        res[mode]["records"] = list(r.last_num_records)
    assert res["none"]["records"] == res["zeros"]["records"]
    for k in ("trans", "d_rot", "strain", "d_opacity", "m2", "normal_grad"):
        assert float(res["none"][k].abs().max()) > 0, k
        assert torch.equal(res["none"][k], res["zeros"][k]), k


@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("H,W", [(144, 176), (40, 48)])
def test_rgb_gradient_only_equals_the_full_backward_fed_zeros_on_the_normal_channels(H, W, fuse):
    """`renderer.rgb_gradient_only` (dm4d_views_backward_rgb, k_render_bwd<6, 3>): no loss reads the normal image, so channels 3..5
    of the upstream gradient are not read (here they hold GARBAGE, NaN included) and the blend backward carries 5 per-entry sums
    instead of 8.  Every gradient must be bit-identical to the 32-byte-record backward fed exact zeros on those channels (a zero
    adds nothing to any sum; the normals then receive no gradient), which is what autograd does in the reference when every
    normal weight of the configuration is 0 (configs/sugar_dynamic_dg.yaml:145-157).  With a depth gradient present the flag
    falls back to the full call (the lean configuration it needs does not apply)."""
    _need_gpu()
    from dreammesh4d_amd import views

    dev = torch.device("cuda:0")
    B, M = 4, 100
    sc, graph, topo, qs, scales, opac, rgb, raw, cams, vm, pm = _scene(2400, M, 4, B, H, W, dev, seed=7)
    fidx = torch.tensor([0, 1, 1, 0], device=dev, dtype=torch.int32)
    gen = torch.Generator().manual_seed(4)
    gC = torch.randn(B, 6, H, W, generator=gen).to(dev)
    gA = torch.randn(B, 1, H, W, generator=gen).to(dev)
    gD = (0.1 * torch.randn(B, 1, H, W, generator=gen)).to(dev)
    gC_zero = gC.clone()
    gC_zero[:, 3:] = 0.0
    gC_junk = gC.clone()
    gC_junk[:, 3:] = float("nan")
    gC_junk[:, 4, ::3] = 1.0e30
    res = {}
    for mode in ("zeros", "rgb", "rgb_with_depth", "zeros_with_depth"):
        r = views.ViewRenderer(graph, topo, H, W, cams[0].tanfov, method="hybrid")
        r.fuse_face_backward = fuse
        r.rgb_gradient_only = mode.startswith("rgb")
        leaves = {k: v[:2].clone().requires_grad_(True) for k, v in raw.items()}
        m2 = torch.zeros(B, r.N, 3, device=dev, requires_grad=True)
        out = views.render_views(r, leaves["trans"], leaves["d_rot"], leaves["strain"], leaves["d_opacity"].squeeze(-1), qs,
                                 scales, opac, rgb, vm, pm, torch.ones(6, device=dev), frame_index=fidx, means2D=m2)
        if mode.endswith("with_depth"):
            # (the flag does not apply: the full call reads all six channels, so it gets the zeros)
            torch.autograd.backward([out["color"], out["depth"], out["alpha"]], [gC_zero, gD, gA])
        else:
            torch.autograd.backward([out["color"], out["alpha"]], [gC_junk if mode == "rgb" else gC_zero, gA])
        res[mode] = {k: v.grad.clone() for k, v in leaves.items()}
        res[mode]["m2"] = m2.grad.clone()
    for a, b in (("zeros", "rgb"), ("zeros_with_depth", "rgb_with_depth")):
        for k in ("trans", "d_rot", "strain", "d_opacity", "m2"):
            assert torch.isfinite(res[b][k]).all() and float(res[a][k].abs().max()) > 0, (a, k)
            assert torch.equal(res[a][k], res[b][k]), (a, b, k, float((res[a][k] - res[b][k]).abs().max()))
    assert not torch.equal(res["zeros"]["trans"], res["zeros_with_depth"]["trans"])


@pytest.mark.parametrize("mode", ["lean32", "lean48", "full64"])
def test_bench_scene_one_view_against_two_oracle_passes(mode):
    """The TIMED kernel instantiations at the TIMED size: one (frame, view) unit of bench.py's 20-unit step (mesh-bound 199,980
    Gaussians, 33,330 faces, 1000 nodes, hybrid skinning, 512 x 512, bench.py's own camera and node outputs) through
    ``render_views`` in the three record modes the bench reports --
        lean32: static appearance frozen, no depth gradient   k_render_bwd<6, 2>   (the headline step)
        lean48: static appearance frozen, depth gradient       k_render_bwd<6, 1>   (`with_depth_gradient`)
        full64: static appearance learnable, depth gradient    k_render_bwd<6, 0>   (`roofline_full`, sugar_static_refine)
    -- against TWO oracle passes (RGB pass with dL/d(colour, depth, alpha), normal pass with dL/dcolour: the reference's
    two rasterizer calls per view, renderer/diff_sugar_rasterizer_temporal.py:169-178,202-211) fed the HIP path's own
    Gaussians: forward image bit-identical, every per-Gaussian gradient within tests/test_raster_gpu.py::_assert_grads' bars."""
    _need_gpu()
    import bench
    from dreammesh4d_amd import ops, views
    from oracle import raster as orc
    from tests.test_raster_gpu import _assert_grads

    dev = torch.device("cuda:0")
    wl = bench.Workload(dev, 0, 1)
    # (bench.Workload is the 20-unit step of BASELINE configs[3]: 4 frames x (4 SDS views + 1 reference view))
    H, W, u = bench.H, bench.W, 9                      # unit 9: frame 1, its fifth camera
    assert wl.views_per_step == 20 and int(wl.fidx[u]) == 1
    f = int(wl.fidx[u])
    with torch.no_grad():
        dx, dr, ds, do = (t[f:f + 1].contiguous() for t in wl.net.node_outputs(wl.nodes, wl.frame_t))
    learn = mode == "full64"
    st = [t.detach().clone().requires_grad_(learn) for t in (wl.scales, wl.opac, wl.rgb)]
    leaves = [t.clone().requires_grad_(True) for t in (dx, dr, ds, do)]
    r = views.ViewRenderer(wl.graph, wl.topo, H, W, wl.cams[0].tanfov, method="hybrid")
    m2 = torch.zeros(1, r.N, 3, device=dev, requires_grad=True)
    out = views.render_views(r, *leaves, wl.qs, st[0], st[1], st[2], wl.vm[u:u + 1], wl.pm[u:u + 1], wl.bg6, means2D=m2)
    nr = r.check()
    gC, gD, gA = wl.gC[u:u + 1], wl.gD[u:u + 1], wl.gA[u:u + 1]
    if mode == "lean32":
        torch.autograd.backward([out["color"], out["alpha"]], [gC, gA])
    else:
        torch.autograd.backward([out["color"], out["depth"], out["alpha"]], [gC, gD, gA])
    torch.cuda.synchronize()
    lg = r.last_grads
    # the Gaussians the HIP path rendered (same kernels as inside render_views: bit-identical)
    with torch.no_grad():
        means, rots, normals = ops.face_gaussians(wl.topo, out["vxyz"][0], out["vrot"][0], wl.qs, grad_mode=r.grad_mode)
    cam = wl.cams[u]
    n = lambda t: t.detach().cpu().numpy()
    ok = dict(image_height=H, image_width=W, tanfovx=cam.tanfov, tanfovy=cam.tanfov, bg=(1, 1, 1), scale_modifier=1.0,
              viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix, campos=cam.campos)
    o1, o2 = orc.RasterOracle(**ok), orc.RasterOracle(**ok)
    o1.forward(n(means), n(st[1]).reshape(-1), colors_precomp=n(st[2]), scales=n(st[0]), rotations=n(rots))
    o2.forward(n(means), n(st[1]).reshape(-1), colors_precomp=n(normals), scales=n(st[0]), rotations=n(rots))
    assert o1.D == nr[0] == o2.D
    assert np.array_equal(n(out["radii"][0]), o1.s["radii"])
    col = n(out["color"][0])
    assert np.array_equal(col[:3].view(np.uint32), o1.s["out_color"].view(np.uint32))
    assert np.array_equal(col[3:].view(np.uint32), o2.s["out_color"].view(np.uint32))
    assert np.array_equal(n(out["depth"][0, 0]).view(np.uint32), o1.s["out_depth"].view(np.uint32))
    assert np.array_equal(n(out["alpha"][0, 0]).view(np.uint32), o1.s["out_alpha"].view(np.uint32))
    g1 = o1.backward(n(gC[0, :3]), n(gD[0, 0]) if mode != "lean32" else None, n(gA[0, 0]))
    g2 = o2.backward(n(gC[0, 3:]), None, None)
    og = {k: g1[k] + g2[k] for k in ("dL_dmeans2D", "dL_dmeans3D", "dL_drots", "dL_dopacity", "dL_dscales")}
    og["dL_dcolors"] = g2["dL_dcolors"] if not learn else np.concatenate([g1["dL_dcolors"], g2["dL_dcolors"]], axis=1)
    g = {"dL_dmeans2D": n(m2.grad[0]), "dL_dmeans3D": n(lg["m3"][0]), "dL_drots": n(lg["rot"][0]),
         "dL_dcolors": n(lg["col"][0]) if learn else n(lg["col"][0, :, 3:])}
    keys = ["dL_dmeans2D", "dL_dcolors", "dL_dmeans3D", "dL_drots"]
    if learn:
        g["dL_dopacity"], g["dL_dscales"] = n(lg["op"][0]), n(lg["sc"][0])
        keys += ["dL_dopacity", "dL_dscales"]
    else:
        assert lg["op"] is None and float(lg["col"][0, :, :3].abs().max()) == 0.0      # lean records: not reduced, not recorded
    for k in keys:
        assert float(np.abs(g[k]).max()) > 0, k
    _assert_grads(g, og, keys=tuple(keys), o=[o1, o2])


def test_bench_step_every_unit_forward_against_two_oracle_passes():
    """The headline step of bench.py (BASELINE configs[3]'s per-GPU share: 4 frames x (4 SDS + 1 reference view) = 20 units, 199,980 Gaussians,
    512 x 512) through the STEP OBJECT, every one of its 20 units against two oracle passes fed the HIP path's own Gaussians: RGB image, normal
    image, depth, alpha bit-identical, radii and duplicate counts equal.  (Gradients of a unit against the oracle:
    test_bench_scene_one_view_against_two_oracle_passes; units of the batch against units alone, bit for bit, with their gradients:
    tests/test_step_gpu.py::test_twenty_unit_step_equals_per_view_and_per_frame_composition.)"""
    _need_gpu()
    import bench
    from dreammesh4d_amd import ops
    from oracle import raster as orc

    dev = torch.device("cuda:0")
    wl = bench.Workload(dev, 0, 1)
    assert wl.views_per_step == 20
    H, W = bench.H, bench.W
    out = wl.dstep(wl.frame_t, wl.vm16, wl.pm16, wl.fidx)
    D = wl.renderer.check()
    torch.cuda.synchronize()
    n = lambda t: t.detach().cpu().numpy()
    col, dep, alp, rad = n(out["color"]), n(out["depth"]), n(out["alpha"]), n(out["radii"])
    gaussians = {}
    for u in range(20):
        f = int(wl.fidx[u])
        if f not in gaussians:
            with torch.no_grad():
                means, rots, normals = ops.face_gaussians(wl.topo, out["vxyz"][f], out["vrot"][f], wl.qs, grad_mode=wl.renderer.grad_mode)
            gaussians[f] = (n(means), n(rots), n(normals))
        means, rots, normals = gaussians[f]
        cam = wl.cams[u]
        ok = dict(image_height=H, image_width=W, tanfovx=cam.tanfov, tanfovy=cam.tanfov, bg=(1, 1, 1), scale_modifier=1.0,
                  viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix, campos=cam.campos)
        o1, o2 = orc.RasterOracle(**ok), orc.RasterOracle(**ok)
        o1.forward(means, n(wl.opac).reshape(-1), colors_precomp=n(wl.rgb), scales=n(wl.scales), rotations=rots)
        o2.forward(means, n(wl.opac).reshape(-1), colors_precomp=normals, scales=n(wl.scales), rotations=rots)
        assert o1.D == D[u] == o2.D, u
        assert np.array_equal(rad[u], o1.s["radii"]), u
        assert np.array_equal(col[u, :3].view(np.uint32), o1.s["out_color"].view(np.uint32)), u
        assert np.array_equal(col[u, 3:].view(np.uint32), o2.s["out_color"].view(np.uint32)), u
        assert np.array_equal(dep[u, 0].view(np.uint32), o1.s["out_depth"].view(np.uint32)), u
        assert np.array_equal(alp[u, 0].view(np.uint32), o1.s["out_alpha"].view(np.uint32)), u
