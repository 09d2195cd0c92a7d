"""Pins the C rasterizer oracle (oracle/raster_oracle.c) -- CPU only.

The reference has no tests / golden vectors for the rasterizer (SURVEY.md section 4, 8c:
"parity unpinned"), so the oracle is pinned against an independent dense fp64
autograd restatement, closed-form cases and invariants.
"""
import math

import numpy as np
import pytest
import torch

from dreammesh4d_amd import synthetic as syn
from oracle import raster as orc
from tests.dense_reference import dense_rasterize


def _scene(n, H, W, seed, log_scale_mean=math.log(0.03), std=0.6, azim=30.0):
    sc = syn.random_splat_scene(n, seed=seed, log_scale_mean=log_scale_mean, log_scale_std=std)
    cam = syn.make_camera(H, W, elev_deg=15.0, azim_deg=azim)
    return sc, cam


def _oracle(sc, cam, bg=(1, 1, 1), scale_mod=1.0):
    o = orc.RasterOracle(image_height=cam.H, image_width=cam.W, tanfovx=cam.tanfov, tanfovy=cam.tanfov, bg=bg,
                         scale_modifier=scale_mod, viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix,
                         campos=cam.campos)
    o.forward(sc["means3D"], sc["opacities"], colors_precomp=sc["colors"], scales=sc["scales"],
              rotations=sc["rotations"])
    return o


def _rects(o, cam):
    gx, gy = (cam.W + 15) // 16, (cam.H + 15) // 16
    xy, r = o.s["xy"].astype(np.float32), o.s["radii"].astype(np.float32)
    f32 = np.float32
    mn = np.stack([np.clip(((xy[:, 0] - r) / f32(16)).astype(np.int64), 0, gx),
                   np.clip(((xy[:, 1] - r) / f32(16)).astype(np.int64), 0, gy)], 1)
    mx = np.stack([np.clip(((xy[:, 0] + r + f32(15)) / f32(16)).astype(np.int64), 0, gx),
                   np.clip(((xy[:, 1] + r + f32(15)) / f32(16)).astype(np.int64), 0, gy)], 1)
    return mn, mx


def test_expf_accuracy():
    xs = np.concatenate([-np.logspace(-6, 1.9, 4000), [0.0, -1e-30, -86.0, -87.0, -100.0, -1e30]])
    for x in xs:
        got = orc.expf(np.float32(x))
        want = math.exp(max(float(np.float32(x)), -86.0))      # (the contract clamps at -86: 4.5e-38, far below any alpha >= 1/255)
        assert abs(got - want) <= 2.5 * np.spacing(np.float32(want)) + 1e-45, (x, got, want)
    assert orc.expf(0.0) == 1.0


def test_sorted_list_invariants():
    sc, cam = _scene(3000, 80, 112, seed=1)
    o = _oracle(sc, cam)
    keys, vals, ranges = o.s["keys"], o.s["values"], o.s["ranges"]
    assert o.D == int(o.s["tiles_touched"].sum()) and o.D > 0
    # lexicographic (tile, depth_bits, gaussian id): what a STABLE radix sort of tile<<32|depth yields
    kv = (keys.astype(object) << 32) | vals.astype(object)
    assert all(kv[i] < kv[i + 1] for i in range(len(kv) - 1))
    dbits = o.s["depths"].view(np.uint32)
    assert np.array_equal((keys & 0xFFFFFFFF).astype(np.uint32), dbits[vals])
    tiles = (keys >> 32).astype(np.int64)
    for t in range(ranges.shape[0]):
        s, e = ranges[t]
        assert np.all(tiles[s:e] == t)
    assert int((ranges[:, 1] - ranges[:, 0]).sum()) == o.D
    # every visible Gaussian appears exactly tiles_touched times
    assert np.array_equal(np.bincount(vals, minlength=len(o.s["radii"])), o.s["tiles_touched"])
    assert np.array_equal(o.s["radii"] > 0, o.s["tiles_touched"] > 0)


def test_alpha_is_one_minus_final_T():
    sc, cam = _scene(2000, 64, 64, seed=2)
    o = _oracle(sc, cam)
    assert np.abs(o.s["out_alpha"] - (1 - o.s["final_T"])).max() < 2e-6
    assert o.s["out_alpha"].max() > 0.5


def test_permutation_invariance():
    sc, cam = _scene(1500, 64, 96, seed=3)
    o1 = _oracle(sc, cam)
    perm = np.random.default_rng(0).permutation(1500)
    sc2 = {k: v[perm] for k, v in sc.items()}
    o2 = _oracle(sc2, cam)
    assert len(np.unique(o1.s["depths"][o1.s["radii"] > 0])) == int((o1.s["radii"] > 0).sum())  # no depth ties
    assert np.array_equal(o1.s["out_color"], o2.s["out_color"])
    assert np.array_equal(o1.s["radii"][perm], o2.s["radii"])
    assert np.array_equal(perm[o2.s["values"]], o1.s["values"])


def test_single_isotropic_splat_closed_form():
    H = W = 64
    cam = syn.make_camera(H, W, elev_deg=0.0, azim_deg=0.0)
    s, op = 0.05, 0.8
    sc = {"means3D": np.zeros((1, 3), np.float32), "scales": np.full((1, 3), s, np.float32),
          "rotations": np.array([[1, 0, 0, 0]], np.float32), "opacities": np.array([op], np.float32),
          "colors": np.array([[0.2, 0.5, 0.9]], np.float32)}
    o = _oracle(sc, cam, bg=(0, 0, 0))
    focal = W / (2 * cam.tanfov)
    var = (focal * s / 3.8) ** 2 + 0.3
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    r2 = (xs - 31.5) ** 2 + (ys - 31.5) ** 2
    alpha = np.minimum(0.99, op * np.exp(-0.5 * r2 / var))
    alpha[alpha < 1 / 255] = 0
    # tile-rect truncation: radius = ceil(3 sigma)
    assert o.s["radii"][0] == math.ceil(3 * math.sqrt(var))
    inside = np.ones_like(alpha, bool)
    assert np.abs(o.s["out_alpha"] - alpha * inside).max() < 2e-5
    assert np.abs(o.s["out_color"][2] - 0.9 * alpha).max() < 2e-5
    assert np.abs(o.s["out_depth"] - 3.8 * alpha).max() < 1e-4
    assert abs(o.s["xy"][0, 0] - 31.5) < 1e-4 and abs(o.s["xy"][0, 1] - 31.5) < 1e-4


def test_culling_and_mark_visible():
    cam = syn.make_camera(32, 32, elev_deg=0.0, azim_deg=0.0)
    # one in front, one behind the camera, one closer than the 0.2 near threshold
    campos = cam.campos.astype(np.float64)
    fwd = -campos / np.linalg.norm(campos)
    means = np.stack([np.zeros(3), campos - fwd * 1.0, campos + fwd * 0.1]).astype(np.float32)
    sc = {"means3D": means, "scales": np.full((3, 3), 0.05, np.float32),
          "rotations": np.tile(np.array([[1, 0, 0, 0]], np.float32), (3, 1)),
          "opacities": np.full(3, 0.5, np.float32), "colors": np.full((3, 3), 0.5, np.float32)}
    o = _oracle(sc, cam)
    assert list(o.s["radii"] > 0) == [True, False, False]


@pytest.mark.parametrize("seed,H,W", [(11, 48, 64), (12, 40, 72)])
def test_oracle_matches_dense_autograd(seed, H, W):
    n = 160
    sc, cam = _scene(n, H, W, seed=seed, log_scale_mean=math.log(0.05), std=0.7)
    # push a few Gaussians outside the 1.3*tanfov clamp and near the image border
    sc["means3D"][:6] *= 3.0
    o = _oracle(sc, cam, bg=(0.3, 0.6, 0.9), scale_mod=1.1)
    rng = np.random.default_rng(seed)
    gC = rng.normal(size=(3, H, W)).astype(np.float32)
    gD = rng.normal(size=(H, W)).astype(np.float32)
    gA = rng.normal(size=(H, W)).astype(np.float32)
    g = o.backward(gC, gD, gA)

    t = lambda a: torch.tensor(np.asarray(a, np.float64), dtype=torch.float64)
    leaves = {k: t(sc[k]).requires_grad_(True) for k in ("means3D", "opacities", "colors", "scales", "rotations")}
    m2d = torch.zeros(n, 3, dtype=torch.float64, requires_grad=True)
    # integer / ordering state: the dense restatement's OWN float64 decisions (preprocess_fp64 takes nothing from the
    # oracle); only Gaussians within float32 rounding of a decision boundary fall back to the oracle's value
    from tests.dense_reference import preprocess_fp64
    p = preprocess_fp64(sc["means3D"], sc["scales"], sc["rotations"], cam.viewmatrix, cam.projmatrix, cam.tanfov,
                        cam.tanfov, H, W, scale_mod=1.1)
    mn_o, mx_o = _rects(o, cam)
    safe = p["margin"] > 2e-3
    assert safe.sum() >= n - 8
    assert np.array_equal(p["radii"][safe], o.s["radii"][safe])
    vis = safe & p["visible"]
    assert np.array_equal(p["rect_min"][vis], mn_o[vis]) and np.array_equal(p["rect_max"][vis], mx_o[vis])
    radii = np.where(safe, p["radii"], o.s["radii"])
    mn = np.where(safe[:, None], p["rect_min"], mn_o)
    mx = np.where(safe[:, None], p["rect_max"], mx_o)
    shown = [i for i in range(n) if radii[i] > 0]
    order_own = sorted(shown, key=lambda i: (p["depth"][i], i))
    order_orc = sorted(shown, key=lambda i: (float(o.s["depths"][i]), i))
    assert order_own == order_orc          # no float32 depth ties in this scene: the fp64 order is the oracle's
    C, D, A = dense_rasterize(leaves["means3D"], m2d, leaves["opacities"], leaves["colors"], leaves["scales"],
                              leaves["rotations"], t(cam.viewmatrix), t(cam.projmatrix), t([0.3, 0.6, 0.9]),
                              cam.tanfov, cam.tanfov, H, W, 1.1, radii, mn, mx, p["depth"])
    assert np.abs(C.detach().numpy() - o.s["out_color"]).max() < 2e-5
    assert np.abs(D.detach().numpy() - o.s["out_depth"]).max() < 1e-4
    assert np.abs(A.detach().numpy() - o.s["out_alpha"]).max() < 2e-5
    loss = (C * t(gC)).sum() + (D * t(gD)).sum() + (A * t(gA)).sum()
    loss.backward()

    def close(name, got, want, rtol=2e-3):
        want = want.numpy()
        scale = np.abs(want).max() + 1e-12
        err = np.abs(got - want).max() / scale
        assert err < rtol, (name, err)

    close("means2D", g["dL_dmeans2D"][:, :2], m2d.grad[:, :2])
    close("opacity", g["dL_dopacity"], leaves["opacities"].grad)
    close("colors", g["dL_dcolors"], leaves["colors"].grad)
    close("means3D", g["dL_dmeans3D"], leaves["means3D"].grad)
    close("scales", g["dL_dscales"], leaves["scales"].grad)
    close("rotations", g["dL_drots"], leaves["rotations"].grad)
    assert (o.s["radii"] > 0).sum() > n // 2


@pytest.mark.parametrize("n,H,W,seed,lsm,spread", [(20_000, 256, 256, 0, math.log(0.008), 1.0),
                                                    (5_000, 100, 173, 1, math.log(0.03), 1.0),
                                                    (4_000, 64, 96, 2, math.log(0.05), 3.5)])   # many off-screen / behind
def test_preprocess_decisions_match_independent_fp64(n, H, W, seed, lsm, spread):
    """Cull, radius, tile rect, tiles_touched and the depth ORDER of the oracle (float32, restating upstream) against an
    independent float64 statement of the same published algorithm (tests/dense_reference.py::preprocess_fp64, which
    takes nothing from the oracle).  They must agree for every Gaussian whose decisions are not within float32
    rounding of a boundary; the boundary cases are counted and must stay a tiny fraction."""
    from tests.dense_reference import preprocess_fp64

    sc = syn.random_splat_scene(n, seed=seed, log_scale_mean=lsm, log_scale_std=0.6)
    sc["means3D"] = (sc["means3D"] * spread).astype(np.float32)
    if spread > 1:
        sc["means3D"][: n // 10] += (syn.make_camera(H, W, azim_deg=30.0).campos * 1.2).astype(np.float32)   # behind / near the camera
    cam = syn.make_camera(H, W, elev_deg=15.0, azim_deg=30.0)
    o = _oracle(sc, cam, scale_mod=1.0)
    p = preprocess_fp64(sc["means3D"], sc["scales"], sc["rotations"], cam.viewmatrix, cam.projmatrix, cam.tanfov, cam.tanfov, H, W)
    safe = p["margin"] > 2e-3            # float32 evaluation error of 3 sigma / pixel coordinates is ~1e-4 here
    n_boundary = int((~safe).sum())
    assert n_boundary < 0.02 * n, n_boundary
    assert np.array_equal(o.s["radii"][safe], p["radii"][safe])
    assert np.array_equal(o.s["tiles_touched"][safe].astype(np.int64), p["tiles_touched"][safe])
    mn, mx = _rects(o, cam)
    vis = safe & p["visible"]
    assert np.array_equal(mn[vis], p["rect_min"][vis]) and np.array_equal(mx[vis], p["rect_max"][vis])
    # boundary cases may differ, but only by one unit of the decision
    d_r = np.abs(o.s["radii"].astype(np.int64) - p["radii"])
    both = (o.s["radii"] > 0) & p["visible"]
    assert d_r[both].max(initial=0) <= 1
    # pixel coordinates and depths themselves
    assert np.abs(o.s["xy"][both] - p["xy"][both]).max() < 2e-3
    assert np.abs(o.s["depths"][both] - p["depth"][both]).max() < 1e-5
    # depth order of the sorted list: within every tile the oracle's order is the fp64 order, except between
    # neighbours whose fp64 depths are closer than float32 resolution
    vals, ranges = o.s["values"], o.s["ranges"]
    z = p["depth"]
    inversions = tight = 0
    for s, e in ranges:
        if e - s < 2:
            continue
        zz = z[vals[s:e]]
        bad = np.nonzero(np.diff(zz) < 0)[0]
        inversions += len(bad)
        tight += int((np.abs(np.diff(zz))[bad] < 1e-6).sum())
    assert inversions == tight, (inversions, tight)
    print(f"preprocess decisions: {n_boundary} of {n} Gaussians within rounding of a boundary (excluded), "
          f"{int((d_r > 0).sum())} radius differences among them, {inversions} depth-order inversions (all below float32 resolution)")
