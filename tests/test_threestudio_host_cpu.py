"""CPU tests of the plugin host (dreammesh4d_amd/threestudio_host.py): registry, structured config parsing, the
``${...}`` resolvers the reference's YAMLs use, Updateable, parse_optimizer -- the host logic of boundary B1
(threestudio/__init__.py:5-32, threestudio/utils/base.py:21-118, threestudio/utils/config.py:11-28,121-123,
threestudio/systems/utils.py:55-89).  The configuration fragments are typed in here (keys / values of
custom/threestudio-dreammesh4d/configs/*.yaml), not read from the reference tree."""
import dataclasses

import pytest
import torch
import torch.nn as nn

from dreammesh4d_amd import threestudio_host as host


def test_registry_has_the_reference_names_and_find_composes():
    want = {"diff-sugar-rasterizer-temporal", "diff-sugar-rasterizer-normal", "dynamic-sugar", "sugar",
            "temporal-stable-zero123-guidance", "stable-zero123-guidance", "solid-color-background", "no-material",
            "sugar-4dgen-system", "sugar-static-system", "temporal-image-datamodule", "single-image-datamodule"}
    assert set(host.__modules__) == want
    assert host.find("sugar") is host.SuGaRModel
    with pytest.raises(ValueError):
        host.register("sugar")(object)                      # "Module sugar already exists! Names of extensions conflict!"
    with pytest.raises(KeyError):
        host.find("no-such-plugin")
    mixed = host.find("solid-color-background:no-material")  # main:sub -> class deriving from (sub, main)
    assert mixed.__mro__[1] is host.NoMaterial and host.SolidColorBackground in mixed.__mro__


def test_parse_structured_defaults_overrides_unknown_and_mandatory_keys():
    C = host._DynamicSuGaRConfig
    c = host.parse_structured(C, {"num_frames": 32, "dg_node_connectivity": 4, "d_scale": False, "dist_mode": "geodisc",
                                  "n_gaussians_per_surface_triangle": 6, "spatial_lr_scale": 1})
    assert (c.num_frames, c.dg_node_connectivity, c.d_scale, c.dist_mode, c.skinning_method) == (32, 4, False, "geodisc", "hybrid")
    assert c.n_dg_nodes == 1000 and c.init_gs_opacity == 0.5 and c.init_gs_scales_s == 1.7 and c.color_clip == 2.0   # class defaults
    assert {f.name for f in dataclasses.fields(C)} >= {"sh_levels", "position_lr", "surface_mesh_to_bind_path", "deformation_lr", "grid_lr"}
    with pytest.raises(KeyError):
        host.parse_structured(C, {"num_framez": 3})
    with pytest.raises(ValueError):
        host.parse_structured(host._TemporalZero123Config, {"pretrained_model_name_or_path": "???"})


def test_resolvers_of_the_reference_yamls():
    root = {"data": {"default_camera_distance": 3.8, "video_length": 32, "default_elevation_deg": 5.0, "image_path": "load/images/a b_rgba.png"},
            "name": "x-${rmspace:${basename:${data.image_path}},_}",
            "system": {"freq": {"guidance_eval": 0}, "loss": {"lambda_sds": [0, 0.1, 0.0, 100], "lambda_x": 0.0},
                       "geometry": {"spatial_extent": "${data.default_camera_distance}"},
                       "guidance_zero123": {"num_frames": "${data.video_length}", "vram_O": "${not:${gt0:${system.freq.guidance_eval}}}",
                                            "cond_elevation_deg": "${data.default_elevation_deg}"},
                       "flags": {"a": "${cmaxgt0:${system.loss.lambda_sds}}", "b": "${cmaxgt0orcmaxgt0:${system.loss.lambda_x},${system.loss.lambda_x}}",
                                 "half": "${idiv:${data.video_length},2}", "decay": "${calc_exp_lr_decay_rate:0.1,10}"}}}
    r = host.resolve(root)
    assert r["system"]["geometry"]["spatial_extent"] == 3.8 and r["system"]["guidance_zero123"] == {"num_frames": 32, "vram_O": True, "cond_elevation_deg": 5.0}
    assert r["system"]["flags"]["a"] is True and r["system"]["flags"]["b"] is False and r["system"]["flags"]["half"] == 16
    assert abs(r["system"]["flags"]["decay"] - 0.1 ** 0.1) < 1e-12
    assert r["name"] == "x-a_b_rgba.png"
    with pytest.raises(ValueError):
        host.resolve({"a": "${b}", "b": "${a}"})


def test_updateable_recurses_and_base_classes_configure():
    calls = []

    class Leaf(host.BaseObject):
        @dataclasses.dataclass
        class Config:
            k: int = 1

        def configure(self, tag="leaf"):
            self.tag = tag

        def update_step(self, epoch, global_step, on_load_weights=False):
            calls.append((self.tag, epoch, global_step))

    class Mod(host.BaseModule):
        @dataclasses.dataclass
        class Config:
            weights: str = None
            width: int = 4

        def configure(self, leaf):
            self.lin = nn.Linear(self.cfg.width, 2)
            self.leaf = leaf

        def update_step(self, epoch, global_step, on_load_weights=False):
            calls.append(("mod", epoch, global_step))

    leaf = Leaf({"k": 3}, tag="L")
    m = Mod({"width": 6}, leaf)
    assert leaf.cfg.k == 3 and m.lin.in_features == 6 and m.leaf is leaf
    m.do_update_step(2, 17)
    assert calls == [("L", 2, 17), ("mod", 2, 17)]


def test_shims_from_cfg_and_parse_optimizer():
    bg = host.find("solid-color-background")({"color": [0.2, 0.4, 0.6], "learned": True})
    assert isinstance(bg.env_color, nn.Parameter) and bg(torch.zeros(1, 2, 2, 3)).shape == (1, 2, 2, 3)
    mat = host.find("no-material")({"n_output_dims": 0})          # the value both shipped YAMLs give ("unused")
    assert mat.n_output_dims == 0
    with pytest.raises(KeyError):
        host.find("no-material")({"n_output_dimz": 3})

    class System(nn.Module):
        def __init__(self):
            super().__init__()
            self.background, self.other = bg, nn.Linear(2, 2)

    # optimizer block of configs/sugar_dynamic_dg.yaml: Adam, lr 0.01, betas [0.9, 0.99], eps 1e-15, params: background lr 0.001
    opt = host.parse_optimizer({"name": "Adam", "args": {"lr": 0.01, "betas": [0.9, 0.99], "eps": 1e-15},
                                "params": {"background": {"lr": 0.001}}}, System())
    assert isinstance(opt, torch.optim.Adam) and len(opt.param_groups) == 1
    g = opt.param_groups[0]
    assert g["lr"] == 0.001 and g["name"] == "background" and g["params"][0] is bg.env_color and g["eps"] == 1e-15
    opt2 = host.parse_optimizer({"name": "AdamW", "args": {"lr": 0.5}}, System())
    assert sum(len(g["params"]) for g in opt2.param_groups) == 3
    with pytest.raises(NotImplementedError):
        host.parse_optimizer({"name": "Adan"}, System())


def test_prune_isolated_points_and_surface_sampling():
    import numpy as np

    from dreammesh4d_amd import synthetic as syn

    v, f = syn.uv_sphere(200, radius=0.5)
    v, f = np.asarray(v, np.float64), np.asarray(f, np.int64)
    # an isolated triangle far away: its three vertices are pruned, the sphere stays
    v2 = np.concatenate([v, [[5, 5, 5], [5, 6, 5], [6, 5, 5]]])
    f2 = np.concatenate([f, [[len(v), len(v) + 1, len(v) + 2]]])
    col = np.random.default_rng(0).random((len(v2), 3))
    pv, pf, pc = host.prune_isolated_points(v2, f2, col)
    assert len(pv) == len(v) and len(pf) == len(f) and np.array_equal(pv, v) and np.array_equal(pc, col[:len(v)])
    pts = host.sample_points_uniformly(v, f, 500, seed=1)
    assert pts.shape == (500, 3) and np.abs(np.linalg.norm(pts, axis=1) - 0.5).max() < 0.04      # on the (200-face) sphere
    assert np.array_equal(pts, host.sample_points_uniformly(v, f, 500, seed=1))


def test_stage_loss_weights_come_from_the_config_block():
    """DynamicStage.from_cfg / StaticStage.from_cfg read `system.loss` (lambda_* -> the loop's terms; C() schedules kept as
    lists), `system.freq`, `num_inter_frames`, `length_inter_frames`; a non-zero weight for a term the loop does not
    compute is an error, not a silently dropped term (C/system/sugar_4dgen.py:296-330)."""
    import torch

    from dreammesh4d_amd.dynamic_stage import DynamicStage, LAMBDA
    from dreammesh4d_amd.static_stage import StaticStage

    class _Stop(Exception):
        pass

    seen = {}

    def fake_init(self, *a, **kw):
        seen.update(kw)
        raise _Stop

    real = DynamicStage.__init__
    DynamicStage.__init__ = fake_init
    try:
        cfg = {"loss": {"lambda_rgb": 1.5, "lambda_mask": [0, 1.0, 2.0, 10], "lambda_depth": 0.0, "lambda_sds_zero123": 0.3},
               "freq": {"milestone_arap_reg": 7, "inter_frame_reg": 2}, "num_inter_frames": 5, "length_inter_frames": 0.25,
               "geometry": {"deformation_lr": 1e-3, "grid_lr": [0, 1e-2, 1e-3, 100]}}
        with pytest.raises(_Stop):
            DynamicStage.from_cfg(cfg, *[None] * 8)
        assert seen["lambdas"] == {"rgb": 1.5, "mask": [0, 1.0, 2.0, 10], "sds_zero123": 0.3}
        assert (seen["milestone_arap_reg"], seen["inter_frame_reg"], seen["num_inter_frames"], seen["length_inter_frames"]) == (7, 2, 5, 0.25)
        assert seen["deformation_lr"] == 1e-3 and seen["grid_lr"] == [0, 1e-2, 1e-3, 100]
        # round 5: the reference's optional terms (OPTIONAL_TERMS) are handed to the stage when they have a weight; a name the reference
        # does not have is still refused
        seen.clear()
        with pytest.raises(_Stop):
            DynamicStage.from_cfg({"loss": {"lambda_depth": 0.05, "lambda_rgb_tv": 0.0}}, *[None] * 8)
        assert seen["lambdas"] == {"depth": 0.05}
        with pytest.raises(NotImplementedError):
            DynamicStage.from_cfg({"loss": {"lambda_not_a_reference_term": 0.05}}, *[None] * 8)
    finally:
        DynamicStage.__init__ = real
    assert set(LAMBDA) == {"sds_zero123", "rgb", "mask", "normal_consistency", "arap_reg_key_frame", "arap_reg_inter_frame"}
    with pytest.raises(NotImplementedError):
        StaticStage.from_cfg({"loss": {"lambda_normal_smooth": 1.0}}, None, None, None, None, 8, 8)


def test_bench_partition_is_the_one_baseline_configs_3_names():
    """bench.py's headline step = BASELINE.json configs[3]'s per-GPU share (rank r takes frames {4r .. 4r+3} x 4 views plus their 4
    reference views = 20 units); the shipped YAML's own iteration (4 frames x (1 SDS + 1 reference view), configs/sugar_dynamic_dg.yaml:9-11,24) is
    the step timed beside it."""
    import json
    import os

    import bench

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg3 = json.load(open(os.path.join(root, "BASELINE.json")))["configs"][3]
    assert "32 frames" in cfg3 and "4 views" in cfg3 and "8 MI355X" in cfg3 and "200k" in cfg3
    assert bench.N_FRAMES == 32 and bench.FRAMES_PER_STEP == 32 // 8
    assert bench.VIEWS_PER_FRAME == 4 + 1 and bench.FRAMES_PER_STEP * bench.VIEWS_PER_FRAME == 20
    assert bench.VIEWS_PER_FRAME_YAML == 2 and bench.N_FACES * 6 == 200_004
    a = bench.parse.__wrapped__() if hasattr(bench.parse, "__wrapped__") else None      # (argparse reads sys.argv: defaults only)
    assert a is None or a.views_per_frame == 5
