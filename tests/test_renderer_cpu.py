"""CPU checks of the renderer glue's closed-form pieces (projection of known points,
Depth2Normal on a plane) and of the plugin name table."""
import math
import types

import numpy as np
import torch

from dreammesh4d_amd import renderer as R, synthetic as syn


def test_cam_info_matches_the_numpy_camera_used_by_the_parity_tests():
    for el, az in ((17.0, -73.0), (-8.0, 140.0), (79.0, 5.0)):
        cam = syn.make_camera(64, 48, elev_deg=el, azim_deg=az)
        c2w = torch.tensor(cam.c2w, dtype=torch.float32)[None]
        f = torch.tensor([cam.fovy])
        wv, full, ctr = R.cam_info_gaussian(c2w, f, f)
        assert np.abs(wv[0].numpy() - cam.viewmatrix).max() < 1e-6
        assert np.abs(full[0].numpy() - cam.projmatrix).max() < 2e-6
        assert np.abs(ctr[0].numpy() - cam.campos).max() < 2e-6


def test_projection_of_known_points():
    """Row-vector convention (threestudio/utils/ops.py:398-413): a point on the optical axis at distance d projects to
    NDC (0, 0) with w = d; a point at the edge of the field of view projects to |x| = 1."""
    cam = syn.make_camera(32, 32, elev_deg=0.0, azim_deg=0.0, dist=3.0, fovy_deg=40.0)
    c2w = torch.tensor(cam.c2w, dtype=torch.float32)[None]
    f = torch.tensor([cam.fovy])
    wv, full, ctr = R.cam_info_gaussian(c2w, f, f)
    origin = torch.tensor([0.0, 0.0, 0.0, 1.0])
    ph = origin @ full[0]
    assert abs(float(ph[3]) - 3.0) < 1e-5 and abs(float(ph[0])) < 1e-5 and abs(float(ph[1])) < 1e-5
    pv = origin @ wv[0]
    assert abs(float(pv[2]) - 3.0) < 1e-5                      # view-space depth is +z in front of the camera
    # a point displaced sideways by d * tan(fov / 2) sits on the image border
    right = torch.tensor(cam.c2w[:3, 0], dtype=torch.float32)
    p = torch.cat([right * 3.0 * math.tan(0.5 * cam.fovy), torch.ones(1)])
    ph = p @ full[0]
    assert abs(abs(float(ph[0] / ph[3])) - 1.0) < 1e-5


def test_depth_to_normal_on_a_plane_is_constant():
    H, W = 12, 16
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    xyz = torch.stack([xs, ys, 0.3 * xs - 0.2 * ys], 0)[None]
    n = torch.nn.functional.normalize(R.depth_to_normal(xyz), dim=1)[0, :, 1:-1, 1:-1]      # interior (zero padding at the border)
    want = -torch.tensor([-0.3, 0.2, 1.0]) / math.sqrt(0.09 + 0.04 + 1.0)          # -cross(d/dx, d/dy), normalised
    assert (n - want[:, None, None]).abs().max() < 1e-6
    # identical to the reference's two 3x3 convolutions
    kx = torch.tensor([[0.0, 0, 0], [-1, 0, 1], [0, 0, 0]]).view(1, 1, 3, 3)
    ky = torch.tensor([[0.0, -1, 0], [0, 0, 0], [0, 1, 0]]).view(1, 1, 3, 3)
    dx = torch.nn.functional.conv2d(xyz.reshape(3, 1, H, W), kx, padding=1).reshape(1, 3, H, W)
    dy = torch.nn.functional.conv2d(xyz.reshape(3, 1, H, W), ky, padding=1).reshape(1, 3, H, W)
    assert torch.equal(R.depth_to_normal(xyz), -torch.cross(dx, dy, dim=1))


def test_rays_hit_the_projected_pixel():
    cam = syn.make_camera(40, 40, elev_deg=25.0, azim_deg=33.0)
    c2w = torch.tensor(cam.c2w, dtype=torch.float32)[None]
    focal = 0.5 * 40 / math.tan(0.5 * cam.fovy)
    ro, rd = R.rays(R.ray_directions(40, 40, focal), c2w, normalize=False)
    f = torch.tensor([cam.fovy])
    _, full, _ = R.cam_info_gaussian(c2w, f, f)
    p = torch.cat([ro[0, 7, 29] + 2.5 * rd[0, 7, 29], torch.ones(1)]) @ full[0]
    px = ((p[0] / p[3] + 1.0) * 40 - 1.0) * 0.5          # the rasterizer's ndc -> pixel map
    py = ((p[1] / p[3] + 1.0) * 40 - 1.0) * 0.5
    assert abs(float(px) - 29.0) < 1e-3 and abs(float(py) - 7.0) < 1e-3


def test_plugin_names_cover_the_reference_registry_entries_of_the_hot_path():
    from dreammesh4d_amd import plugins

    assert set(plugins.PLUGINS) == {"diff-sugar-rasterizer-temporal", "diff-sugar-rasterizer-normal", "dynamic-sugar", "sugar",
                                    "temporal-stable-zero123-guidance", "stable-zero123-guidance", "solid-color-background",
                                    "no-material", "sugar-4dgen-system", "sugar-static-system", "temporal-image-datamodule",
                                    "single-image-datamodule"}
    bgm = plugins.PLUGINS["solid-color-background"]({"color": (0.2, 0.4, 0.6)}).eval()
    out = bgm(torch.zeros(2, 3, 5, 3))
    assert out.shape == (2, 3, 5, 3) and torch.allclose(out[1, 2, 4], torch.tensor([0.2, 0.4, 0.6]))
    aug = plugins.PLUGINS["solid-color-background"]({"random_aug": True, "random_aug_prob": 1.0}).train()(torch.zeros(2, 3, 5, 3))
    assert torch.equal(aug[0, 0, 0], aug[0, 2, 4]) and not torch.equal(aug[0, 0, 0], aug[1, 0, 0])      # one colour per batch item
    mat = plugins.PLUGINS["no-material"]()
    assert torch.allclose(mat(torch.zeros(4, 3)), torch.full((4, 3), 0.5))
    reg = {}
    fake = types.SimpleNamespace(register=lambda name: (lambda cls: reg.setdefault(name, cls)))
    names = plugins.register(fake, prefix="")
    assert set(names) == set(plugins.PLUGINS) and all(reg[n] is plugins.PLUGINS[n] for n in names)
    for meth in ("batch_forward", "forward"):
        assert callable(getattr(plugins.PLUGINS["diff-sugar-rasterizer-temporal"], meth))
    for meth in ("get_timed_gs_all_single_time", "get_timed_gs_normals", "get_timed_vertex_xyz", "get_timed_vertex_rotation",
                 "get_timed_surface_mesh", "get_points_rgb", "merge_optimizer", "update_learning_rate", "update_step"):
        assert callable(getattr(plugins.PLUGINS["dynamic-sugar"], meth))
    for prop in ("get_xyz", "get_scaling", "get_rotation", "get_opacity", "get_features", "get_xyz_verts", "get_faces"):
        assert isinstance(getattr(plugins.PLUGINS["dynamic-sugar"], prop), property)
        assert isinstance(getattr(plugins.PLUGINS["sugar"], prop), property)
    assert isinstance(plugins.PLUGINS["sugar"].get_gs_normals, property)
    for meth in ("batch_forward", "forward"):
        assert callable(getattr(plugins.PLUGINS["diff-sugar-rasterizer-normal"], meth))
    for meth in ("get_points_rgb", "merge_optimizer", "update_learning_rate", "update_step", "training_setup"):
        assert callable(getattr(plugins.PLUGINS["sugar"], meth))
