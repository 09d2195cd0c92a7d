"""Tile-record mode of the batched path (`ViewRenderer(deterministic=False)`, dm4d_views.record_mode = DM4D_RECORDS_TILE):
one backward record per (Gaussian, tile), the sixteen cells of a tile summed in LDS with float atomics by
k_render_bwd_tile.  The forward must not change a bit; the gradients must meet the SAME per-element bar against the CPU
oracle as the deterministic (Gaussian, cell) kernels (tests/test_raster_gpu.py::_assert_grads), for the lean records of
the dynamic stage and for the full ones, on tiles of one window, of several windows (> 512 entries) and with wide cells."""
import numpy as np
import pytest
import torch

from tests.test_raster_gpu import _assert_grads, _oracle
from tests.test_views_gpu import _need_gpu, _scene

pytestmark = pytest.mark.gpu


def _run(r, raw, qs, st, vm, pm, gC, gD, gA, dev):
    from dreammesh4d_amd import views

    leaves = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    out = views.render_views(r, leaves["trans"], leaves["d_rot"], leaves["strain"], leaves["d_opacity"].squeeze(-1), qs,
                             st[0], st[1], st[2], vm, pm, torch.ones(6, device=dev))
    r.check()
    torch.autograd.backward([out["color"], out["depth"], out["alpha"]], [gC, gD, gA])
    torch.cuda.synchronize()
    g = {k: (None if v is None else v.detach().cpu().numpy()) for k, v in r.last_grads.items()}
    return out, leaves, g


@pytest.mark.parametrize("H,W,learnable", [(144, 176, False), (144, 176, True), (40, 48, False), (40, 48, True)])
def test_tile_record_mode_meets_the_oracle_bar(H, W, learnable):
    """40x48: 14,400 splats on 12 tiles -- every tile list spans several LDS windows and most cells are wide."""
    _need_gpu()
    from dreammesh4d_amd import ops, views

    dev = torch.device("cuda:0")
    B, M = 2, 100
    sc, graph, topo, qs, scales, opac, rgb, raw, cams, vm, pm = _scene(2400, M, 4, B, H, W, dev, seed=2)
    gen = torch.Generator().manual_seed(1)
    gC = torch.randn(B, 6, H, W, generator=gen).to(dev)
    gD = (0.1 * torch.randn(B, 1, H, W, generator=gen)).to(dev)
    gA = torch.randn(B, 1, H, W, generator=gen).to(dev)
    res = {}
    for det in (True, False):
        r = views.ViewRenderer(graph, topo, H, W, cams[0].tanfov, method="hybrid", deterministic=det)
        st = [t.clone().requires_grad_(learnable) for t in (scales, opac, rgb)]
        res[det] = _run(r, raw, qs, st, vm, pm, gC, gD, gA, dev) + (r,)
    for k in ("color", "depth", "alpha", "radii"):
        assert torch.equal(res[True][0][k], res[False][0][k]), k          # the forward does not depend on the record mode
    nrec = {det: res[det][3].last_num_records for det in res}
    assert all(a <= b for a, b in zip(nrec[False], nrec[True])) and sum(nrec[False]) < 0.7 * sum(nrec[True]), nrec
    g = res[False][2]
    for b in range(B):
        with torch.no_grad():
            xyz, vrot = ops.skin_vertices(graph, raw["trans"][b], raw["d_rot"][b], raw["strain"][b], raw["d_opacity"][b].view(-1), "hybrid")
            means, rots, normals = ops.face_gaussians(topo, xyz, vrot, qs)
        s_ = dict(means3D=means.cpu().numpy(), opacities=opac.view(-1).cpu().numpy(), scales=scales.cpu().numpy(),
                  rotations=rots.cpu().numpy())
        o1 = _oracle(s_, cams[b], (1, 1, 1), 1.0, colors_precomp=rgb.cpu().numpy(), scales=s_["scales"], rotations=s_["rotations"])
        o2 = _oracle(s_, cams[b], (1, 1, 1), 1.0, colors_precomp=normals.cpu().numpy(), scales=s_["scales"], rotations=s_["rotations"])
        g1 = o1.backward(gC[b, :3].cpu().numpy(), gD[b, 0].cpu().numpy(), gA[b, 0].cpu().numpy())
        g2 = o2.backward(gC[b, 3:].cpu().numpy(), None, None)
        keys = ["dL_dmeans2D", "dL_dmeans3D", "dL_drots"] + (["dL_dopacity", "dL_dscales"] if learnable else [])
        summed = {k: g1[k] + g2[k] for k in keys}
        mine = {"dL_dmeans2D": g["m2"][b], "dL_dmeans3D": g["m3"][b], "dL_drots": g["rot"][b]}
        if learnable:
            mine.update({"dL_dopacity": g["op"][b].reshape(summed["dL_dopacity"].shape), "dL_dscales": g["sc"][b]})
        _assert_grads(mine, summed, keys=tuple(keys), o=[o1, o2])
        cols = {"n": g["col"][b][:, 3:]}
        ocol = {"n": g2["dL_dcolors"]}
        if learnable:
            cols["c"], ocol["c"] = g["col"][b][:, :3], g1["dL_dcolors"]
        _assert_grads(cols, ocol, keys=tuple(cols))
    # the node gradients (what training uses) of the two modes agree to float32 summation noise
    for k in raw:
        a, c = res[False][1][k].grad, res[True][1][k].grad
        assert (a - c).abs().max() <= 2e-5 * c.abs().max() + 1e-12, k
