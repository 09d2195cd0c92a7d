"""GPU parity tests: HIP rasterizer (through the C ABI) vs the CPU oracle on identical seeded
splat sets.  Bars (BASELINE.json north_star): bit-exact tile/sort indices and radii;
RGB/depth/alpha L-inf <= 1e-4 (we additionally assert the forward is bit-identical, which the
shared arithmetic contract makes possible); gradients within a relative tolerance
(different summation order, float32 vs double accumulation in the checker)."""
import math

import numpy as np
import pytest
import torch

from dreammesh4d_amd import synthetic as syn

pytestmark = pytest.mark.gpu

TOL = 1e-4      # north_star tolerance for RGB L-inf
GRAD_RTOL = 1e-4   # relative, per element
GRAD_ATOL = 5e-6   # x max|oracle gradient| of the tensor, per element


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")


def _oracle(sc, cam, bg, scale_mod, **kw):
    from oracle import raster as orc

    o = orc.RasterOracle(image_height=cam.H, image_width=cam.W, tanfovx=cam.tanfov, tanfovy=cam.tanfov, bg=bg,
                         scale_modifier=scale_mod, viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix,
                         campos=cam.campos)
    o.forward(sc["means3D"], sc["opacities"], **kw)
    return o


def _both(sc, cam, bg=(1, 1, 1), scale_mod=1.0, mode="colors"):
    from tests.hip_raster import HipRaster

    if mode == "colors":
        okw = dict(colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
        hkw = dict(colors=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
    elif mode == "sh":
        sh = ((sc["colors"] - 0.5) / 0.28209479177387814)[:, None, :] * 1.6 - 0.3   # some clamp below 0
        okw = dict(shs=sh, scales=sc["scales"], rotations=sc["rotations"])
        hkw = dict(shs=sh, scales=sc["scales"], rotations=sc["rotations"])
    o = _oracle(sc, cam, bg, scale_mod, **okw)
    h = HipRaster(cam, bg=bg, scale_mod=scale_mod)
    out = h.forward(sc["means3D"], sc["opacities"], **hkw)
    return o, h, out


def _assert_forward_parity(o, h, out, exact=True):
    color, radii, depth, alpha = out
    s = h.state()
    assert h.D == o.D
    assert np.array_equal(radii, o.s["radii"])
    assert np.array_equal(s["tiles_touched"], o.s["tiles_touched"])
    vis = o.s["radii"] > 0
    for k in ("xy", "depths", "conic_opacity"):
        assert np.array_equal(s[k][vis].view(np.uint32), o.s[k][vis].view(np.uint32)), k
    assert np.array_equal(s["ranges"], o.s["ranges"]) or np.array_equal(
        s["ranges"][o.s["ranges"][:, 1] > o.s["ranges"][:, 0]], o.s["ranges"][o.s["ranges"][:, 1] > o.s["ranges"][:, 0]])
    assert np.array_equal(s["values"], o.s["values"])
    assert np.array_equal(s["keys"], o.s["keys"])
    assert np.abs(color - o.s["out_color"]).max() <= TOL
    assert np.abs(alpha - o.s["out_alpha"]).max() <= TOL
    assert np.abs(depth - o.s["out_depth"]).max() <= TOL * 10
    if exact:
        assert np.array_equal(s["n_contrib"], o.s["n_contrib"])
        assert np.array_equal(s["final_T"].view(np.uint32), o.s["final_T"].view(np.uint32))
        assert np.array_equal(color.view(np.uint32), o.s["out_color"].view(np.uint32))
        assert np.array_equal(depth.view(np.uint32), o.s["out_depth"].view(np.uint32))
        assert np.array_equal(alpha.view(np.uint32), o.s["out_alpha"].view(np.uint32))


BLEND_KEYS = ("dL_dmeans2D", "dL_dopacity", "dL_dcolors")          # sums of per-pixel terms, straight out of the blend backward
DERIVED_KEYS = ("dL_dmeans3D", "dL_dscales", "dL_drots", "dL_dcov3D", "dL_dsh")   # pushed through the preprocess backward
NOISE = 2e-6       # relative rounding noise of a float32 blend-level sum (what GRAD_RTOL / GRAD_ATOL admit is larger)


def _assert_grads(g, og, keys=("dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dscales", "dL_drots"), o=None):
    """Per ELEMENT: |hip - oracle| <= GRAD_RTOL * |oracle| + GRAD_ATOL * max|oracle| (of that tensor) [+ 8 * sens].

    Blend-level gradients (means2D, opacity, colours): the kernel sums a Gaussian's per-pixel terms in float32 in its own
    order (16-pixel cell sums, then cells), the oracle accumulates them in double in upstream's pixel order; the
    difference is rounding noise of the largest terms of the sum, hence the absolute part.
    Derived gradients (means3D, scales, rotations, cov3D): the preprocess backward is ill-conditioned for the thin
    mesh-bound splats (scale 3.8e-6 along the normal, conic -> cov2D -> cov3D -> scale / rotation), so identical code
    fed with blend-level sums that differ in the last bits gives visibly different results.  `sens` measures exactly
    that ON THE ORACLE: its own stage 2 re-run with its blend-level gradients perturbed by NOISE (relative, two seeds);
    the bar then admits 8 x the larger response.  Requires `o` (the RasterOracle whose backward() produced `og`)."""
    worst = {}
    report = {}
    sens = {}
    if o is not None and any(k in DERIVED_KEYS for k in keys):
        for oi in (o if isinstance(o, (list, tuple)) else [o]):      # several oracles: `og` is the SUM of their gradients
            base = oi.preprocess_backward(0.0)
            if oi is o:
                for k in keys:
                    if k in DERIVED_KEYS and og.get(k) is not None:
                        assert np.array_equal(base[k], og[k]), k               # stage 2 alone reproduces the full backward
            p1, p2 = oi.preprocess_backward(NOISE, seed=1), oi.preprocess_backward(NOISE, seed=2)
            for k in keys:
                if k in DERIVED_KEYS and base.get(k) is not None:
                    sens[k] = sens.get(k, 0.0) + np.maximum(np.abs(p1[k] - base[k]), np.abs(p2[k] - base[k]))
    for k in keys:
        a, b = g[k], og[k]
        assert a is not None and b is not None, k
        assert np.isfinite(a).all(), k
        b = b.reshape(a.shape)
        bound = GRAD_RTOL * np.abs(b) + GRAD_ATOL * (np.abs(b).max() + 1e-30)
        if k in DERIVED_KEYS:
            assert k in sens, f"{k}: derived gradients need the oracle object (o=...) for the conditioning term"
            # a Gaussian's sensitivity is a property of the Gaussian: use the largest response over its components
            s_ = sens[k].reshape(a.shape[0], -1).max(axis=1)
            bound = bound + 8.0 * s_.reshape((-1,) + (1,) * (a.ndim - 1))
        ratio = np.abs(a - b) / bound
        worst[k] = float(ratio.max()) if ratio.size else 0.0
        i = int(ratio.argmax()) if ratio.size else 0
        # the bound is wide by construction for the derived gradients: what the error IS, in absolute terms and relative to the tensor
        # and to the element, is printed per tensor so that a regression inside the bound is visible
        err = np.abs(a - b)
        big = np.abs(b) >= 1e-3 * (np.abs(b).max() + 1e-30)          # elements that are not rounding dust themselves
        report[k] = {"max_abs_err": float(err.max()) if err.size else 0.0, "max_abs_oracle": float(np.abs(b).max()) if b.size else 0.0,
                     "err_over_tensor_max": float(err.max() / (np.abs(b).max() + 1e-30)) if err.size else 0.0,
                     "max_rel_err_of_elements_above_1e-3_of_max": float((err[big] / np.abs(b[big])).max()) if big.any() else 0.0,
                     "max_bound_at_worst": float(bound.reshape(-1)[i]) if ratio.size else 0.0}
        assert worst[k] <= 1.0, (k, worst[k], i, a.reshape(-1)[i], b.reshape(-1)[i], report[k])
    print("grad error / bound:", {k: round(v, 3) for k, v in worst.items()})
    for k, r_ in report.items():
        print(f"  {k}: " + ", ".join(f"{n_} {v:.3e}" for n_, v in r_.items()))
    return worst


@pytest.mark.parametrize("n,H,W,seed,lsm", [
    (10_000, 256, 256, 0, math.log(0.008)),      # BASELINE configs[0]
    (3_000, 100, 173, 1, math.log(0.03)),        # ragged image (not a multiple of 16)
    (500, 33, 17, 2, math.log(0.1)),             # tiny image, big splats
])
def test_forward_backward_parity(n, H, W, seed, lsm):
    _need_gpu()
    sc = syn.random_splat_scene(n, seed=seed, log_scale_mean=lsm, log_scale_std=0.6)
    cam = syn.make_camera(H, W, elev_deg=15.0, azim_deg=40.0)
    o, h, out = _both(sc, cam, bg=(0.2, 0.7, 1.0), scale_mod=1.0)
    _assert_forward_parity(o, h, out)
    rng = np.random.default_rng(seed)
    gC = rng.normal(size=(3, H, W)).astype(np.float32)
    gD = rng.normal(size=(H, W)).astype(np.float32)
    gA = rng.normal(size=(H, W)).astype(np.float32)
    og = o.backward(gC, gD, gA)
    g = h.backward(gC, gD, gA)
    _assert_grads(g, og, o=o)
    g2 = h.backward(gC, gD, gA)   # deterministic: no float atomics anywhere
    for k in g:
        if g[k] is not None:
            assert np.array_equal(g[k].view(np.uint32), g2[k].view(np.uint32)), k


def test_sh_degree0_path_and_null_grads():
    _need_gpu()
    sc = syn.random_splat_scene(2000, seed=5, log_scale_mean=math.log(0.03), log_scale_std=0.5)
    cam = syn.make_camera(96, 96)
    o, h, out = _both(sc, cam, mode="sh")
    _assert_forward_parity(o, h, out)
    gC = np.random.default_rng(0).normal(size=(3, 96, 96)).astype(np.float32)
    og = o.backward(gC, None, None)
    g = h.backward(gC, None, None)
    _assert_grads(g, og, keys=("dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dscales", "dL_drots", "dL_dsh"), o=o)
    assert (o.s["clamped"].sum() > 0)


def test_cov3d_precomp_path():
    _need_gpu()
    from tests.hip_raster import HipRaster

    sc = syn.random_splat_scene(1500, seed=6, log_scale_mean=math.log(0.03), log_scale_std=0.5)
    cam = syn.make_camera(80, 64)
    o0 = _oracle(sc, cam, (0, 0, 0), 1.0, colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
    cov = o0.s["cov3D"].copy()
    cov[o0.s["radii"] <= 0] = np.eye(3, dtype=np.float32)[[0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2]] * 1e-4
    o = _oracle(sc, cam, (0, 0, 0), 1.0, colors_precomp=sc["colors"], cov3D_precomp=cov)
    h = HipRaster(cam, bg=(0, 0, 0))
    out = h.forward(sc["means3D"], sc["opacities"], colors=sc["colors"], cov3D=cov)
    _assert_forward_parity(o, h, out)
    gC = np.random.default_rng(1).normal(size=(3, 80, 64)).astype(np.float32)
    og = o.backward(gC)
    g = h.backward(gC)
    _assert_grads(g, og, keys=("dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D"), o=o)


def test_edge_cases_empty_and_culled():
    _need_gpu()
    from tests.hip_raster import HipRaster

    cam = syn.make_camera(40, 56)
    h = HipRaster(cam, bg=(0.1, 0.2, 0.3))
    color, radii, depth, alpha = h.forward(np.zeros((0, 3), np.float32), np.zeros((0,), np.float32),
                                           colors=np.zeros((0, 3), np.float32), scales=np.zeros((0, 3), np.float32),
                                           rotations=np.zeros((0, 4), np.float32))
    assert h.D == 0 and np.allclose(color[0], 0.1) and np.allclose(color[2], 0.3) and not alpha.any()
    # everything behind the camera
    sc = syn.random_splat_scene(300, seed=3)
    sc["means3D"] = sc["means3D"] + cam.campos[None, :] * 2.0
    o, h2, out = _both(sc, cam, bg=(0.1, 0.2, 0.3))
    assert o.D == 0 and h2.D == 0 and not out[1].any()
    g = h2.backward(np.ones((3, 40, 56), np.float32))
    assert all(v is None or not np.any(v) for v in g.values())


def test_depth_ties_break_by_gaussian_id():
    _need_gpu()
    sc = syn.random_splat_scene(800, seed=7, log_scale_mean=math.log(0.05), log_scale_std=0.3)
    sc["means3D"][1::2] = sc["means3D"][0::2]           # pairs of coincident splats: identical depth bits
    sc["colors"][1::2] = 1.0 - sc["colors"][0::2]
    cam = syn.make_camera(64, 64)
    o, h, out = _both(sc, cam)
    _assert_forward_parity(o, h, out)
    d = o.s["depths"]
    assert np.array_equal(d[0::2].view(np.uint32), d[1::2].view(np.uint32))


def test_oversized_tile_uses_global_sort_path():
    _need_gpu()
    # > 4096 duplicates in one 16x16 tile: the HBM-resident bitonic path of k_tile_sort
    n = 6000
    rng = np.random.default_rng(9)
    sc = syn.random_splat_scene(n, seed=9, log_scale_mean=math.log(0.002), log_scale_std=0.2)
    sc["means3D"] = (rng.normal(size=(n, 3)) * 0.004).astype(np.float32)
    cam = syn.make_camera(64, 64)
    o, h, out = _both(sc, cam)
    assert (o.s["ranges"][:, 1].astype(np.int64) - o.s["ranges"][:, 0]).max() > 4096
    _assert_forward_parity(o, h, out)
    gC = rng.normal(size=(3, 64, 64)).astype(np.float32)
    _assert_grads(h.backward(gC), o.backward(gC), o=o)


@pytest.mark.parametrize("n", [1500, 5000])
def test_deep_translucent_stack_long_cells(n):
    """Thousands of faint splats on a few pixels: cell lists far beyond kLongCell (384) that the forward really
    consumes to the end.  n = 1500 keeps every tile within the small sort variant (long cells blended by the regular
    forward, the long-cell blocks of k_render_bwd in the backward); n = 5000 goes through the large variant, whose long cells take the
    early forward kernel (k_render_fwd_long) as well.  Forward bit-identical, gradients within the usual bar."""
    _need_gpu()
    rng = np.random.default_rng(n)
    sc = syn.random_splat_scene(n, seed=n, log_scale_mean=math.log(0.004), log_scale_std=0.3)
    sc["means3D"] = (rng.normal(size=(n, 3)) * 0.006).astype(np.float32)
    sc["opacities"] = rng.uniform(0.004, 0.02, size=sc["opacities"].shape).astype(np.float32)
    cam = syn.make_camera(64, 64)
    o, h, out = _both(sc, cam)
    per_tile = (o.s["ranges"][:, 1].astype(np.int64) - o.s["ranges"][:, 0]).max()
    assert (per_tile <= 2048) if n == 1500 else (per_tile > 2048)
    assert o.s["n_contrib"].max() > 384 * 2          # the blend really walks the long lists
    assert 0.0 < o.s["final_T"].min() and o.s["final_T"].min() < 0.5
    _assert_forward_parity(o, h, out)
    gC = rng.normal(size=(3, 64, 64)).astype(np.float32)
    gD = rng.normal(size=(64, 64)).astype(np.float32)
    gA = rng.normal(size=(64, 64)).astype(np.float32)
    g = h.backward(gC, gD, gA)
    _assert_grads(g, o.backward(gC, gD, gA), o=o)
    g2 = h.backward(gC, gD, gA)
    for k in g:
        if g[k] is not None:
            assert np.array_equal(g[k].view(np.uint32), g2[k].view(np.uint32)), k


def test_more_than_256_binning_workgroups():
    """270,000 Gaussians = 264 workgroups of K1: the per-tile column scan (k_colscan) keeps a column segment in
    registers up to 256 workgroups and falls back to a two-pass walk beyond (BASELINE configs[4] has 977)."""
    _need_gpu()
    sc = syn.random_splat_scene(270_000, seed=77, log_scale_mean=math.log(0.003), log_scale_std=0.4)
    cam = syn.make_camera(80, 112)
    o, h, out = _both(sc, cam)
    assert o.D > 400_000
    _assert_forward_parity(o, h, out)


def test_giant_splat_covers_every_tile():
    _need_gpu()
    sc = syn.random_splat_scene(64, seed=10, log_scale_mean=math.log(0.02), log_scale_std=0.3)
    sc["scales"][0] = (0.5, 0.5, 0.5)
    sc["means3D"][0] = 0
    sc["opacities"][0] = 0.6
    cam = syn.make_camera(128, 96)
    o, h, out = _both(sc, cam)
    assert o.s["tiles_touched"][0] == 8 * 6
    _assert_forward_parity(o, h, out)
    gC = np.random.default_rng(2).normal(size=(3, 128, 96)).astype(np.float32)
    _assert_grads(h.backward(gC), o.backward(gC), o=o)


def test_capacity_overflow_is_flagged_not_fatal():
    _need_gpu()
    from tests.hip_raster import HipRaster

    sc = syn.random_splat_scene(4000, seed=4, log_scale_mean=math.log(0.03), log_scale_std=0.4)
    cam = syn.make_camera(96, 96)
    h = HipRaster(cam)
    h.forward(sc["means3D"], sc["opacities"], colors=sc["colors"], scales=sc["scales"], rotations=sc["rotations"],
              capacity=100)
    assert h.D > 100 and h.overflowed() == 1
    h.backward(np.ones((3, 96, 96), np.float32))   # memory-safe
    h2 = HipRaster(cam)
    h2.forward(sc["means3D"], sc["opacities"], colors=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
    assert h2.overflowed() == 0


def test_full_size_200k_512():
    """BASELINE headline size: 200k Gaussians at 512^2 (random-splat scene), full parity."""
    _need_gpu()
    sc = syn.random_splat_scene(200_000, seed=0)
    cam = syn.make_camera(512, 512)
    o, h, out = _both(sc, cam)
    _assert_forward_parity(o, h, out)
    rng = np.random.default_rng(0)
    gC = rng.normal(size=(3, 512, 512)).astype(np.float32)
    gD = rng.normal(size=(512, 512)).astype(np.float32) * 0.1
    gA = rng.normal(size=(512, 512)).astype(np.float32)
    _assert_grads(h.backward(gC, gD, gA), o.backward(gC, gD, gA), o=o)


def test_autograd_operator_matches_oracle():
    """The drop-in module (GaussianRasterizationSettings / GaussianRasterizer) end to end."""
    _need_gpu()
    import dreammesh4d_amd.diff_gaussian_rasterization as dgr

    dev = torch.device("cuda:0")
    n, H, W = 5000, 128, 160
    sc = syn.random_splat_scene(n, seed=21, log_scale_mean=math.log(0.02), log_scale_std=0.5)
    cam = syn.make_camera(H, W, elev_deg=30, azim_deg=-60)
    T = lambda a, rg=False: torch.tensor(a, device=dev).requires_grad_(rg)
    m3, op = T(sc["means3D"], True), T(sc["opacities"][:, None], True)
    col, scl, rot = T(sc["colors"], True), T(sc["scales"], True), T(sc["rotations"], True)
    m2 = torch.zeros_like(m3, requires_grad=True)
    rs = dgr.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam.tanfov, tanfovy=cam.tanfov,
                                           bg=T(np.ones(3, np.float32)), scale_modifier=1.0,
                                           viewmatrix=T(cam.viewmatrix), projmatrix=T(cam.projmatrix), sh_degree=0,
                                           campos=T(cam.campos), prefiltered=False, debug=False)
    rast = dgr.GaussianRasterizer(raster_settings=rs)
    with pytest.raises(Exception):
        rast(means3D=m3, means2D=m2, opacities=op, shs=None, colors_precomp=None, scales=scl, rotations=rot)
    color, radii, depth, alpha = rast(means3D=m3, means2D=m2, shs=None, colors_precomp=col, opacities=op, scales=scl,
                                      rotations=rot, cov3D_precomp=None)
    assert color.shape == (3, H, W) and depth.shape == (1, H, W) and alpha.shape == (1, H, W)
    assert radii.dtype == torch.int32 and radii.shape == (n,)
    rng = np.random.default_rng(3)
    gC = rng.normal(size=(3, H, W)).astype(np.float32)
    gA = rng.normal(size=(1, H, W)).astype(np.float32)
    ((color * T(gC)).sum() + (alpha * T(gA)).sum()).backward()
    o = _oracle(sc, cam, (1, 1, 1), 1.0, colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
    og = o.backward(gC, None, gA[0])
    assert np.array_equal(color.detach().cpu().numpy().view(np.uint32), o.s["out_color"].view(np.uint32))
    assert np.array_equal(radii.cpu().numpy(), o.s["radii"])
    got = {"dL_dmeans2D": m2.grad.cpu().numpy(), "dL_dopacity": op.grad.cpu().numpy()[:, 0],
           "dL_dcolors": col.grad.cpu().numpy(), "dL_dmeans3D": m3.grad.cpu().numpy(),
           "dL_dscales": scl.grad.cpu().numpy(), "dL_drots": rot.grad.cpu().numpy()}
    _assert_grads(got, og, o=o)
    vis = rast.markVisible(m3)
    assert vis.dtype == torch.bool and bool(vis.all())
    # a second backward through the same graph (retain_graph, or two losses back-propagated separately): same gradients
    c2, _, _, a2 = rast(means3D=m3, means2D=m2, opacities=op, colors_precomp=col, scales=scl, rotations=rot)
    l1, l2 = (c2 * T(gC)).sum(), (a2 * T(gA)).sum()
    for t in (m3, m2, op, col, scl, rot):
        t.grad = None
    l1.backward(retain_graph=True)
    g_first = m3.grad.clone()
    l2.backward(retain_graph=True)
    both = m3.grad.clone()
    for t in (m3, m2, op, col, scl, rot):
        t.grad = None
    (l1 + l2).backward()
    assert torch.isfinite(g_first).all() and torch.allclose(m3.grad, both, rtol=1e-4, atol=1e-6 * float(both.abs().max()))


def test_record_count_matches_the_cell_blocks():
    """R (dm4d_rasterize_num_records) = sum over visible Gaussians of the 4x4-pixel cells inside their tile rect
    that the alpha >= 1/255 ellipse bound reaches: at most 16 per duplicate; large splats (every cell of
    every tile they touch) still give exact images and gradients, also with a larger record capacity."""
    _need_gpu()
    sc = syn.random_splat_scene(300, seed=5, log_scale_mean=math.log(0.05), log_scale_std=0.3)
    cam = syn.make_camera(128, 128, azim_deg=20.0)
    o, h, out = _both(sc, cam)
    assert 0 < h.R <= 16 * h.D, (h.R, h.D)
    assert h.R > 4 * h.D      # these splats are much larger than a cell
    _assert_forward_parity(o, h, out)
    rs = np.random.RandomState(0)
    gC, gD, gA = (rs.randn(3, 128, 128).astype(np.float32), rs.randn(128, 128).astype(np.float32),
                  rs.randn(128, 128).astype(np.float32))
    og = o.backward(gC, gD, gA)
    g1 = h.backward(gC, gD, gA)
    _assert_grads(g1, og, o=o)
    g2 = h.backward(gC, gD, gA, record_capacity=h.R + 1000)
    for k in g1:
        if g1[k] is not None:
            assert np.array_equal(g1[k], g2[k], equal_nan=True), k      # deterministic, capacity-independent


@pytest.mark.parametrize("crowded", [False, True])
def test_fused_six_channel_pass_equals_two_passes(crowded):
    """C = 6 (RGB + normal in one pass) must equal the reference's two passes: same image planes,
    summed geometry gradients, per-pass colour gradients.  crowded: 5000 faint splats on a few pixels, i.e. the
    6-channel long-cell kernels (k_render_fwd_long<6>, the long-cell blocks of k_render_bwd<6, false>)."""
    _need_gpu()
    from tests.hip_raster import HipRaster

    rng = np.random.default_rng(31)
    if crowded:
        n, H, W = 5_000, 64, 64
        sc = syn.random_splat_scene(n, seed=31, log_scale_mean=math.log(0.004), log_scale_std=0.3)
        sc["means3D"] = (rng.normal(size=(n, 3)) * 0.006).astype(np.float32)
        sc["opacities"] = rng.uniform(0.004, 0.02, size=sc["opacities"].shape).astype(np.float32)
    else:
        n, H, W = 20_000, 200, 264
        sc = syn.random_splat_scene(n, seed=31, log_scale_mean=math.log(0.012), log_scale_std=0.5)
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    cam = syn.make_camera(H, W, elev_deg=25, azim_deg=100)
    bg = (1.0, 1.0, 1.0)
    o1 = _oracle(sc, cam, bg, 1.0, colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
    o2 = _oracle(sc, cam, bg, 1.0, colors_precomp=nrm, scales=sc["scales"], rotations=sc["rotations"])
    h = HipRaster(cam, bg=(1, 1, 1, 1, 1, 1))
    col6 = np.concatenate([sc["colors"], nrm], axis=1)
    color, radii, depth, alpha = h.forward(sc["means3D"], sc["opacities"], colors=col6, scales=sc["scales"],
                                           rotations=sc["rotations"])
    assert np.array_equal(color[:3].view(np.uint32), o1.s["out_color"].view(np.uint32))
    assert np.array_equal(color[3:].view(np.uint32), o2.s["out_color"].view(np.uint32))
    assert np.array_equal(depth.view(np.uint32), o1.s["out_depth"].view(np.uint32))
    assert np.array_equal(alpha.view(np.uint32), o1.s["out_alpha"].view(np.uint32))
    assert np.array_equal(radii, o1.s["radii"])
    gC = rng.normal(size=(3, H, W)).astype(np.float32)
    gN = rng.normal(size=(3, H, W)).astype(np.float32)
    gD = rng.normal(size=(H, W)).astype(np.float32) * 0.1
    gA = rng.normal(size=(H, W)).astype(np.float32)
    g1, g2 = o1.backward(gC, gD, gA), o2.backward(gN, None, None)
    g = h.backward(np.concatenate([gC, gN]), gD, gA)
    summed = {k: g1[k] + g2[k] for k in ("dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dscales", "dL_drots")}
    _assert_grads(g, summed, keys=tuple(summed), o=[o1, o2])
    _assert_grads({"a": g["dL_dcolors"][:, :3], "b": g["dL_dcolors"][:, 3:]},
                  {"a": g1["dL_dcolors"], "b": g2["dL_dcolors"]}, keys=("a", "b"))


def test_one_shot_forward_with_alloc_callback():
    """dm4d_rasterize_forward (include/dm4d.h): the entry in the shape of upstream's RasterizeGaussiansCUDA, which asks
    the CALLER for its three workspaces through a callback (upstream's resize functors).  The callback here hands out
    torch tensors; the result must be the staged path's, bit for bit, and the workspaces must serve the backward."""
    _need_gpu()
    import ctypes as C

    from dreammesh4d_amd import _lib
    from tests.hip_raster import HipRaster

    sc = syn.random_splat_scene(6000, seed=12, log_scale_mean=math.log(0.02), log_scale_std=0.5)
    cam = syn.make_camera(112, 144, elev_deg=20, azim_deg=75)
    o = _oracle(sc, cam, (1, 1, 1), 1.0, colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
    h = HipRaster(cam)
    h.forward(sc["means3D"], sc["opacities"], colors=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
    L, dev, st = h.L, h.dev, torch.cuda.current_stream(h.dev).cuda_stream
    held, calls = {}, []

    def alloc(ctx, which, nbytes):
        calls.append((which, nbytes))
        held[which] = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)
        assert held[which].data_ptr() % 16 == 0
        return held[which].data_ptr()

    cb = _lib.ALLOC_FN(alloc)
    color, depth, alpha = torch.empty(3, cam.H, cam.W, device=dev), torch.empty(cam.H, cam.W, device=dev), torch.empty(cam.H, cam.W, device=dev)
    radii = torch.empty(h.N, dtype=torch.int32, device=dev)
    D = _lib.check(L.dm4d_rasterize_forward(h.settings, h.inputs, color.data_ptr(), depth.data_ptr(), alpha.data_ptr(),
                                            radii.data_ptr(), cb, None, st), "dm4d_rasterize_forward")
    torch.cuda.synchronize()
    assert D == o.D == h.D
    assert [w for w, _ in calls] == [0, 1, 2]                       # geom, then (after the one sync) binning and image
    assert calls[0][1] == L.dm4d_raster_geom_bytes(h.N, cam.H, cam.W) and calls[1][1] == L.dm4d_raster_binning_bytes(D)
    assert np.array_equal(color.cpu().numpy().view(np.uint32), o.s["out_color"].view(np.uint32))
    assert np.array_equal(depth.cpu().numpy().view(np.uint32), o.s["out_depth"].view(np.uint32))
    assert np.array_equal(alpha.cpu().numpy().view(np.uint32), o.s["out_alpha"].view(np.uint32))
    assert np.array_equal(radii.cpu().numpy(), o.s["radii"])
    # the caller-owned workspaces feed the backward
    h.geom, h.binning, h.image, h.radii, h.cap = held[0], held[1], held[2], radii, D
    gC = np.random.default_rng(5).normal(size=(3, cam.H, cam.W)).astype(np.float32)
    _assert_grads(h.backward(gC), o.backward(gC), o=o)
    # a failing allocator is an error code, not a crash
    bad = _lib.ALLOC_FN(lambda ctx, which, nbytes: None if which == 1 else held[which].data_ptr())
    rc = L.dm4d_rasterize_forward(h.settings, h.inputs, color.data_ptr(), depth.data_ptr(), alpha.data_ptr(),
                                  radii.data_ptr(), bad, None, st)
    assert rc == -3 and b"alloc" in L.dm4d_last_error()


def test_cfg5_one_million_gaussians_1024_oracle_parity():
    """BASELINE configs[4] at the operator: 1,000,002 Gaussians at 1024^2 = 4096 tiles (44-bit keys upstream), 977
    binning workgroups.  Full forward parity (keys, values, ranges, radii, n_contrib, final_T, image bit-identical) and
    the gradient bar, against the oracle."""
    _need_gpu()
    sc = syn.random_splat_scene(1_000_002, seed=50, log_scale_mean=math.log(0.0025), log_scale_std=0.35)
    cam = syn.make_camera(1024, 1024, elev_deg=25.0, azim_deg=130.0)
    o, h, out = _both(sc, cam)
    assert o.s["ranges"].shape[0] == 4096 and o.D > 2_000_000
    _assert_forward_parity(o, h, out)
    rng = np.random.default_rng(50)
    gC = rng.normal(size=(3, 1024, 1024)).astype(np.float32)
    gA = rng.normal(size=(1024, 1024)).astype(np.float32)
    _assert_grads(h.backward(gC, None, gA), o.backward(gC, None, gA), o=o)
