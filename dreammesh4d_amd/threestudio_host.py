"""A minimal host for the reference's plugin API (boundary B1): what the YAML configs of
custom/threestudio-dreammesh4d bind to, without threestudio, OmegaConf or Lightning.

The reference constructs every component as ``threestudio.find(cfg.<x>_type)(cfg.<x>, *args)``
(threestudio/__init__.py:5-32): a class registered under a name, whose ``__init__(cfg, *args, **kwargs)`` parses ``cfg``
into its nested ``Config`` dataclass, sets ``self.device``, calls ``self.configure(*args, **kwargs)`` and answers
``update_step(epoch, global_step, on_load_weights)`` (threestudio/utils/base.py:21-57,70-118).  This module provides

* ``register(name)`` / ``find(name)``          -- the registry, pre-populated under the REFERENCE's names with the classes below
* ``parse_structured(Config, cfg)``           -- dataclass from a dict: unknown keys raise, as OmegaConf's structured mode does
* ``resolve(root)``                           -- the ``${a.b}`` references and the resolvers the reference registers
                                                 (threestudio/utils/config.py:11-28) on a plain nested dict
* ``Updateable`` / ``BaseObject`` / ``BaseModule``
* the eight classes of the hot path, each constructible as ``cls(cfg_dict, *args)``:
    ``dynamic-sugar`` (C/geometry/dynamic_sugar.py:42-164), ``sugar`` (C/geometry/sugar.py:33-117),
    ``diff-sugar-rasterizer-temporal`` (C/renderer/diff_sugar_rasterizer_temporal.py:56-80),
    ``diff-sugar-rasterizer-normal`` (C/renderer/diff_sugar_rasterizer_normal.py:54-78),
    ``temporal-stable-zero123-guidance`` (C/guidance/temporal_stable_zero123_guidance.py:76-172),
    ``stable-zero123-guidance`` (threestudio/models/guidance/stable_zero123_guidance.py:75-170),
    ``solid-color-background``, ``no-material``
* ``parse_optimizer(config, model)``           -- threestudio/systems/utils.py:55-89 for the optimisers torch ships

What it does NOT contain: the launcher, the systems, the data modules, the exporters.  The guidances'
``prepare_embeddings`` (guidance :174-226: the conditioning frames through the checkpoint's CLIP ViT-L/14 image tower and the VAE
encoder's posterior mean) runs from the configuration's own keys -- ``cond_video_dir`` / ``cond_image_path`` + the checkpoint -- through
``clip_vit.py`` (round 5; parity unpinned: `clip`, `kornia`, `cv2` are absent from the image); two extra ``cfg`` keys the reference's
dataclasses do not have remain as OPTIONS: ``cond_embeddings_path`` (a torch file {"c_crossattn" [L,1,768], "c_concat" [L,4,32,32]}
computed elsewhere, e.g. by the reference itself) and the deformation graph's node samples when reproducibility across runs is
wanted (``dg_node_seed``; the reference draws them with open3d's ``sample_points_uniformly``, dynamic_sugar.py:752-753).
"""
import dataclasses
import math
import os
import re
from dataclasses import dataclass, field
from typing import Any, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import renderer as _renderer
from . import shims as _shims
from . import sugar as _sugar
from . import zero123 as _z

__modules__ = {}


def register(name):
    def decorator(cls):
        if name in __modules__:
            raise ValueError(f"Module {name} already exists! Names of extensions conflict!")
        __modules__[name] = cls
        return cls

    return decorator


def find(name):
    """``threestudio.find``: a registered class, or ``main:sub1,sub2`` = a new class deriving from (sub1, sub2, main)."""
    if ":" in name:
        main_name, sub_name = name.split(":")
        name_list = sub_name.split(",") + [main_name]
        return type(f"{main_name}.{sub_name}", tuple(__modules__[n] for n in name_list), {})
    return __modules__[name]


# ------------------------------------------------------------------------------------------------ config
def _c_max(value):
    """``C_max`` of threestudio/utils/config.py:31-50: the largest value a scheduled scalar takes."""
    if isinstance(value, (int, float)):
        return value
    value = list(value)
    if len(value) >= 6:
        value = [value[0], value[1], max([value[2]] + [value[i] for i in range(4, len(value), 2)]), value[3]]
    if len(value) == 3:
        value = [0] + value
    if len(value) != 4:
        raise TypeError("Scalar specification only supports a 3/4-list or a piecewise list")
    return max(value[1], value[2])


_RESOLVERS = {
    "calc_exp_lr_decay_rate": lambda factor, n: factor ** (1.0 / n), "add": lambda a, b: a + b, "sub": lambda a, b: a - b,
    "mul": lambda a, b: a * b, "div": lambda a, b: a / b, "idiv": lambda a, b: a // b, "basename": lambda p: os.path.basename(p),
    "rmspace": lambda s, sub: s.replace(" ", sub), "tuple2": lambda s: [float(s), float(s)], "gt0": lambda s: s > 0,
    "cmaxgt0": lambda s: _c_max(s) > 0, "not": lambda s: not s, "cmaxgt0orcmaxgt0": lambda a, b: _c_max(a) > 0 or _c_max(b) > 0,
}
_MISSING = "???"


def _lookup(root, path):
    node = root
    for key in path.split("."):
        node = node[int(key)] if isinstance(node, (list, tuple)) else node[key]
    return node


def _split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
            continue
        depth += ch == "{"
        depth -= ch == "}"
        cur += ch
    return out + [cur]


def _literal(s):
    s = s.strip()
    for conv in (int, float):
        try:
            return conv(s)
        except ValueError:
            pass
    return {"true": True, "false": False, "null": None}.get(s.lower(), s)


def _resolve_value(v, root, stack=()):
    if isinstance(v, dict):
        return {k: _resolve_value(x, root, stack) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_resolve_value(x, root, stack) for x in v]
    if not isinstance(v, str) or "${" not in v:
        return v

    def expr(e):
        e = e.strip()
        if ":" in e and re.match(r"^[A-Za-z_][A-Za-z0-9_]*:", e):
            fn, args = e.split(":", 1)
            if fn not in _RESOLVERS:
                raise KeyError(f"unknown resolver {fn!r} in {v!r}")
            return _RESOLVERS[fn](*[_resolve_value(a.strip(), root, stack) if "${" in a else _literal(a) for a in _split_args(args)])
        if e in stack:
            raise ValueError(f"circular reference {e!r}")
        return _resolve_value(_lookup(root, e), root, stack + (e,))

    m = re.fullmatch(r"\$\{(.*)\}", v.strip(), flags=re.S)
    if m and v.strip().count("${") - v.strip().count("}") <= 0 and _balanced(m.group(1)):
        return expr(m.group(1))                      # the whole string is one interpolation: keep the value's type
    # string interpolation
    out, i = "", 0
    while i < len(v):
        if v.startswith("${", i):
            j, depth = i + 2, 1
            while depth:
                depth += v.startswith("${", j)
                depth -= v[j] == "}"
                j += 1
            out += str(expr(v[i + 2:j - 1]))
            i = j
        else:
            out += v[i]
            i += 1
    return out


def _balanced(s):
    depth = 0
    for i, ch in enumerate(s):
        if s.startswith("${", i):
            depth += 1
        elif ch == "}":
            depth -= 1
            if depth < 0:
                return False
    return depth == 0


def resolve(root):
    """A nested dict with every ``${...}`` replaced (OmegaConf.resolve for the forms the reference's YAMLs use)."""
    return _resolve_value(root, root)


def parse_structured(fields, cfg=None):
    """``parse_structured(self.Config, cfg)`` (threestudio/utils/config.py:121-123): the dataclass with its defaults
    overridden by ``cfg``; a key the dataclass does not declare is an error; ``???`` is a missing mandatory value."""
    cfg = {} if cfg is None else (dataclasses.asdict(cfg) if dataclasses.is_dataclass(cfg) else dict(cfg))
    names = {f.name for f in dataclasses.fields(fields)}
    unknown = [k for k in cfg if k not in names]
    if unknown:
        raise KeyError(f"{fields.__qualname__}: unknown configuration key(s) {unknown}")
    out = fields(**cfg)
    for f in dataclasses.fields(fields):
        if getattr(out, f.name) == _MISSING:
            raise ValueError(f"{fields.__qualname__}.{f.name} is a mandatory value (???) that was not set")
    return out


def get_device():
    return torch.device("cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")) if torch.cuda.is_available() else "cpu")


class Updateable:
    def do_update_step(self, epoch, global_step, on_load_weights=False):
        for attr in self.__dir__():
            if attr.startswith("_"):
                continue
            try:
                module = getattr(self, attr)
            except Exception:
                continue
            if isinstance(module, Updateable) and module is not self:
                module.do_update_step(epoch, global_step, on_load_weights=on_load_weights)
        self.update_step(epoch, global_step, on_load_weights=on_load_weights)

    def update_step(self, epoch, global_step, on_load_weights=False):
        pass


class BaseObject(Updateable):
    @dataclass
    class Config:
        pass

    def __init__(self, cfg=None, *args, **kwargs):
        super().__init__()
        self.cfg = parse_structured(self.Config, cfg)
        self.device = get_device()
        self.configure(*args, **kwargs)

    def configure(self, *args, **kwargs):
        pass


class BaseModule(nn.Module, Updateable):
    @dataclass
    class Config:
        weights: Optional[str] = None

    def __init__(self, cfg=None, *args, **kwargs):
        nn.Module.__init__(self)
        self.cfg = parse_structured(self.Config, cfg)
        self.configure(*args, **kwargs)
        if self.cfg.weights is not None:               # "path/to/ckpt:module_name" (base.py:104-113)
            from .wire_formats import load_module_weights

            path, module_name = self.cfg.weights.split(":")
            state, epoch, step = load_module_weights(path, module_name=module_name, map_location="cpu")
            self.load_state_dict(state, strict=False)
            self.do_update_step(epoch, step, on_load_weights=True)

    def configure(self, *args, **kwargs):
        pass


# ------------------------------------------------------------------------------------------------ shims
@register("solid-color-background")
class SolidColorBackground(_shims.SolidColorBackground, Updateable):
    @dataclass
    class Config:
        weights: Optional[str] = None
        n_output_dims: int = 3
        color: Tuple = (1.0, 1.0, 1.0)
        learned: bool = False
        random_aug: bool = False
        random_aug_prob: float = 0.5

    def __init__(self, cfg=None):
        self.cfg = c = parse_structured(self.Config, cfg)
        super().__init__(c.n_output_dims, tuple(c.color), c.learned, c.random_aug, c.random_aug_prob)


@register("no-material")
class NoMaterial(_shims.NoMaterial, Updateable):
    @dataclass
    class Config:
        weights: Optional[str] = None
        n_output_dims: int = 3
        color_activation: str = "sigmoid"
        input_feature_dims: Optional[int] = None
        mlp_network_config: Optional[dict] = None
        requires_normal: bool = False

    def __init__(self, cfg=None):
        self.cfg = c = parse_structured(self.Config, cfg)
        if c.input_feature_dims is not None or c.mlp_network_config is not None:
            raise NotImplementedError("no-material with a feature network (tiny-cuda-nn) is not on the hot path")
        super().__init__(c.n_output_dims, c.color_activation, c.requires_normal)


# ------------------------------------------------------------------------------------------------ geometry
def sample_points_uniformly(verts, faces, n, seed=0):
    """Area-weighted uniform samples on the mesh surface (what open3d's ``sample_points_uniformly`` draws; its own
    random stream is not reproduced -- the node set is random in the reference too)."""
    rng = np.random.default_rng(seed)
    v, f = np.asarray(verts, np.float64), np.asarray(faces, np.int64)
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    tri = rng.choice(len(f), size=n, p=area / area.sum())
    r1, r2 = np.sqrt(rng.random(n)), rng.random(n)
    return ((1 - r1)[:, None] * a[tri] + (r1 * (1 - r2))[:, None] * b[tri] + (r1 * r2)[:, None] * c[tri]).astype(np.float32)


def prune_isolated_points(verts, faces, colors):
    """``SuGaRModel.prune_isolated_points`` (sugar.py:119-161): keep the first connected component (over the one-ring
    graph, searched from vertex 0, 1, ...) that holds more than 75 % of the vertices, drop the faces that lose a corner."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import connected_components

    V = len(verts)
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    g = sp.coo_matrix((np.ones(len(e)), (e[:, 0], e[:, 1])), shape=(V, V))
    _, lab = connected_components(g, directed=False)
    size = np.bincount(lab)
    big = [c for c in np.unique(lab) if size[c] > math.ceil(V * 0.75)]
    if not big:
        raise AssertionError("no connected component holds more than 75 % of the vertices")
    first = min(int(np.nonzero(lab == c)[0][0]) for c in big)
    keep = lab == lab[first]
    new = -np.ones(V, np.int64)
    new[keep] = np.arange(int(keep.sum()))
    nf = new[faces]
    return verts[keep], nf[(nf >= 0).all(axis=1)], colors[keep]


def _load_mesh(cfg, o3d_mesh=None):
    from .wire_formats import read_ply

    if o3d_mesh is not None:             # anything with .vertices / .triangles / .vertex_colors (the reference passes an open3d mesh)
        verts, faces = np.asarray(o3d_mesh.vertices, np.float64), np.asarray(o3d_mesh.triangles, np.int64)
        colors = np.asarray(getattr(o3d_mesh, "vertex_colors", np.zeros((0, 3))), np.float64)
    else:
        if not cfg.surface_mesh_to_bind_path:
            raise ValueError("surface_mesh_to_bind_path is empty: the mesh-bound geometry needs the refined mesh of the static stage")
        m = read_ply(cfg.surface_mesh_to_bind_path)
        verts, faces, colors = m["verts"], m["faces"], (m["colors"] if m["colors"] is not None else np.zeros((0, 3)))
    if len(colors) == 0:
        colors = np.ones_like(verts) * 0.5
    return prune_isolated_points(verts, faces, colors)


@dataclass
class _SuGaRConfig:                                   # SuGaRModel.Config (sugar.py:35-72), BaseGeometry.Config
    weights: Optional[str] = None
    sh_levels: int = 1
    position_lr: Any = 0.001
    feature_lr: Any = 0.01
    opacity_lr: Any = 0.05
    scaling_lr: Any = 0.005
    rotation_lr: Any = 0.005
    learnable_positions: bool = False
    triangle_scale: float = 1.0
    n_gaussians_per_surface_triangle: int = 1
    keep_track_of_knn: bool = False
    knn_to_track: int = 16
    beta_mode: str = "average"
    primitive_types: str = "diamond"
    surface_mesh_to_bind_path: str = ""
    learn_surface_mesh_positions: bool = True
    learn_surface_mesh_opacity: bool = True
    learn_surface_mesh_scales: bool = True
    freeze_gaussians: bool = False
    spatial_lr_scale: float = 10.0
    spatial_extent: float = 3.5
    color_clip: Any = 2.0
    gs_color_inherit_vertices: bool = True
    init_gs_opacity: float = 0.5
    geometry_convert_from: str = ""
    square_size_in_texture: int = 10
    pred_normal: bool = False
    init_gs_scales_s: float = 1.7


@register("sugar")
class SuGaRModel(_sugar.SuGaR, Updateable):
    Config = _SuGaRConfig

    def __init__(self, cfg=None, o3d_mesh=None):
        self.cfg = c = parse_structured(self.Config, cfg)
        verts, faces, colors = _load_mesh(c, o3d_mesh)
        super().__init__(verts, faces, n_gaussians_per_surface_triangle=c.n_gaussians_per_surface_triangle,
                         spatial_extent=c.spatial_extent, vertex_colors=colors if c.gs_color_inherit_vertices else None,
                         learn_positions=c.learn_surface_mesh_positions, learn_opacities=c.learn_surface_mesh_opacity,
                         learn_scales=c.learn_surface_mesh_scales, freeze_gaussians=c.freeze_gaussians,
                         position_lr=c.position_lr, feature_lr=c.feature_lr, opacity_lr=c.opacity_lr, scaling_lr=c.scaling_lr,
                         rotation_lr=c.rotation_lr, spatial_lr_scale=c.spatial_lr_scale, init_gs_opacity=c.init_gs_opacity,
                         init_gs_scales_s=c.init_gs_scales_s, color_clip=c.color_clip, device=get_device())
        if c.weights is not None:
            from .wire_formats import load_module_weights

            path, module_name = c.weights.split(":")
            state, epoch, step = load_module_weights(path, module_name=module_name, map_location="cpu")
            self.load_state_dict(state, strict=False)
            self.do_update_step(epoch, step, on_load_weights=True)


@dataclass
class _DynamicSuGaRConfig(_SuGaRConfig):              # DynamicSuGaRModel.Config (dynamic_sugar.py:44-74)
    num_frames: int = 14
    static_learnable: bool = False
    use_deform_graph: bool = True
    dynamic_mode: str = "deformation"
    n_dg_nodes: int = 1000
    dg_node_connectivity: int = 8
    dg_trans_lr: Any = 0.001
    dg_rot_lr: Any = 0.001
    dg_scale_lr: Any = 0.001
    vert_trans_lr: Any = 0.001
    vert_rot_lr: Any = 0.001
    vert_scale_lr: Any = 0.001
    deformation_lr: Any = 0.001
    grid_lr: Any = 0.001
    d_xyz: bool = True
    d_rotation: bool = True
    d_opacity: bool = False
    d_scale: bool = True
    dist_mode: str = "eucdisc"
    skinning_method: str = "hybrid"
    dg_node_seed: int = 0                             # (not a reference key: seed of the node samples, see the module docstring)


@register("dynamic-sugar")
class DynamicSuGaRModel(_sugar.DynamicSuGaR, Updateable):
    Config = _DynamicSuGaRConfig

    def __init__(self, cfg=None, o3d_mesh=None):
        from .graph_build import build_deformation_graph

        self.cfg = c = parse_structured(self.Config, cfg)
        if c.dynamic_mode != "deformation" or not c.use_deform_graph:
            raise NotImplementedError("only dynamic_mode: deformation with use_deform_graph: true (the shipped configuration; "
                                      "`discrete` + hybrid cannot even be constructed in the reference, SURVEY.md Appendix A)")
        dev = get_device()
        verts, faces, colors = _load_mesh(c, o3d_mesh)
        nodes = sample_points_uniformly(verts, faces, c.n_dg_nodes, seed=c.dg_node_seed)
        idx, w = build_deformation_graph(verts, faces, nodes, c.dg_node_connectivity, c.dist_mode, device=dev)
        super().__init__(verts, faces, nodes, idx, w, n_gaussians_per_surface_triangle=c.n_gaussians_per_surface_triangle,
                         skinning_method=c.skinning_method, spatial_extent=c.spatial_extent,
                         vertex_colors=colors if c.gs_color_inherit_vertices else None, deformation_lr=c.deformation_lr,
                         grid_lr=c.grid_lr, d_scale=c.d_scale, init_gs_opacity=c.init_gs_opacity,
                         init_gs_scales_s=c.init_gs_scales_s, learn_opacities=c.learn_surface_mesh_opacity, device=dev)
        self.num_frames, self.dynamic_mode = c.num_frames, c.dynamic_mode
        if c.static_learnable:
            # dynamic_sugar.py:79-87: the static SuGaR tensors keep requires_grad (they receive gradients -- the FULL blend
            # backward, dL/dopacity / dL/d rgb / dL/dscales reduced and recorded too -- but training_setup_dynamic puts only
            # the deformation network into the optimiser, :167-235).  Gradients reach the appearance tensors (_scales,
            # all_densities, _sh_coordinates_dc); dL/d(static vertices) and dL/d(static in-plane rotation) THROUGH the
            # skinning are not produced by the HIP skinning kernels (they take the static mesh as a constant).
            self.set_static_learnable(True)
        if c.weights is not None:
            from .wire_formats import load_geometry

            path, _ = c.weights.split(":")
            load_geometry(self, path, strict=False)


# ------------------------------------------------------------------------------------------------ renderers
@dataclass
class _RasterizerConfig:                              # Rasterizer.Config -> Renderer.Config (renderers/base.py:17-19)
    weights: Optional[str] = None
    radius: float = 1.0
    debug: bool = False
    invert_bg_prob: float = 1.0
    back_ground_color: Tuple = (1, 1, 1)


class _RendererPlugin(Updateable):
    Config = _RasterizerConfig

    def _init_plugin(self, cfg, geometry, material, background):
        self.cfg = parse_structured(self.Config, cfg)
        self.configure(geometry, material, background)

    def configure(self, geometry, material, background):
        # non-owning references, as threestudio's Renderer keeps them (renderers/base.py:28-35)
        self.sub_modules = {"geometry": geometry, "material": material, "background": background}

    @property
    def material(self):
        return self.sub_modules["material"]

    @property
    def background(self):
        return self.sub_modules["background"]


@register("diff-sugar-rasterizer-temporal")
class DiffGaussian(_renderer.DiffGaussianTemporal, _RendererPlugin):
    def __init__(self, cfg=None, geometry=None, material=None, background=None):
        self._init_plugin(cfg, geometry, material, background)
        _renderer.DiffGaussianTemporal.__init__(self, geometry, back_ground_color=tuple(float(x) for x in self.cfg.back_ground_color))


@register("diff-sugar-rasterizer-normal")
class DiffSuGaR(_renderer.DiffSuGaRNormal, _RendererPlugin):
    def __init__(self, cfg=None, geometry=None, material=None, background=None):
        self._init_plugin(cfg, geometry, material, background)
        _renderer.DiffSuGaRNormal.__init__(self, geometry, back_ground_color=tuple(float(x) for x in self.cfg.back_ground_color),
                                           invert_bg_prob=self.cfg.invert_bg_prob)


# ------------------------------------------------------------------------------------------------ guidance
def _zero123_from_config(pretrained_config, pretrained_model_name_or_path, device):
    """``load_model_from_config`` (guidance :53-73): the LatentDiffusion hyper-parameters from the model YAML
    (load/zero123/sd-objaverse-finetune-c_concat-256.yaml), the weights from the checkpoint's ``state_dict``."""
    import yaml

    with open(pretrained_config) as fh:
        conf = yaml.safe_load(fh)
    p = conf["model"]["params"]
    up, dd = p["unet_config"]["params"], p["first_stage_config"]["params"]["ddconfig"]
    unet_kwargs = dict(in_channels=up["in_channels"], out_channels=up["out_channels"], model_channels=up["model_channels"],
                       attention_resolutions=tuple(up["attention_resolutions"]), num_res_blocks=up["num_res_blocks"],
                       channel_mult=tuple(up["channel_mult"]), num_heads=up["num_heads"], context_dim=up["context_dim"])
    vae_kwargs = dict(ch=dd["ch"], ch_mult=tuple(dd["ch_mult"]), num_res_blocks=dd["num_res_blocks"], in_channels=dd["in_channels"],
                      z_channels=dd["z_channels"])
    with torch.device(device):
        model = _z.Zero123(unet_kwargs=unet_kwargs, vae_kwargs=vae_kwargs, scale_factor=p.get("scale_factor", 0.18215),
                           timesteps=p["timesteps"], linear_start=p["linear_start"], linear_end=p["linear_end"])
    if not os.path.exists(pretrained_model_name_or_path):
        raise FileNotFoundError(f"{pretrained_model_name_or_path}: the Zero123 checkpoint (load/zero123/download.sh) is not there")
    sd = torch.load(pretrained_model_name_or_path, map_location="cpu")
    sd = sd.get("state_dict", sd)
    missing = _z.load_zero123_state_dict(model, sd)
    if missing:
        raise KeyError(f"checkpoint lacks {len(missing)} tensors of the UNet / VAE encoder / cc_projection, e.g. {missing[:3]}")
    # the checkpoint's CLIP image tower (`cond_stage_model.model.visual.*`): only `prepare_embeddings` needs it, once, at set-up time --
    # kept as the plain tensors until then (and dropped afterwards), never moved to the device with the model
    model.__dict__["_clip_visual_sd"] = {k: v for k, v in sd.items() if k.startswith("cond_stage_model.model.visual.")}
    return model


@dataclass
class _TemporalZero123Config:                         # TemporalStableZero123Guidance.Config (guidance :79-103)
    pretrained_model_name_or_path: str = "load/zero123/stable-zero123.ckpt"
    pretrained_config: str = "load/zero123/sd-objaverse-finetune-c_concat-256.yaml"
    vram_O: bool = True
    num_frames: int = 14
    cond_video_dir: str = "load/videos/anya"
    cond_elevation_deg: float = 0.0
    cond_azimuth_deg: float = 0.0
    cond_camera_distance: float = 1.2
    guidance_scale: float = 5.0
    grad_clip: Optional[Any] = None
    half_precision_weights: bool = True
    min_step_percent: float = 0.02
    max_step_percent: float = 0.98
    chunk_size: Optional[int] = None
    cond_embeddings_path: Optional[str] = None        # (not a reference key, see the module docstring)


def _load_embeddings(c, n, model=None, device=None, clip_tower=None):
    """The conditioning embeddings of the guidance: ``prepare_embeddings_video(cfg.cond_video_dir)`` / ``prepare_embeddings(
    cfg.cond_image_path)`` as the reference computes them at ``configure`` time (guidance :166,174-226) from the configuration's OWN keys
    -- the frames through the checkpoint's CLIP ViT-L/14 image tower and the VAE encoder's posterior mean (clip_vit.py; parity
    unpinned: `clip` / `kornia` / `cv2` are absent) -- or, when the non-reference key ``cond_embeddings_path`` is set, from that file."""
    if not c.cond_embeddings_path:
        from . import clip_vit

        if hasattr(c, "cond_video_dir"):
            paths = [clip_vit.video_frame_path(c.cond_video_dir, i) for i in range(n)]
        else:
            paths = [c.cond_image_path]
        lost = [p for p in paths if not os.path.exists(p)]
        if lost:
            raise FileNotFoundError(f"conditioning frame(s) not found: {lost[:3]} (cond_video_dir / cond_image_path of the configuration; "
                                    "or pass precomputed embeddings as cond_embeddings_path)")
        if clip_tower is None:
            vis = getattr(model, "_clip_visual_sd", None)
            if not vis:
                raise KeyError("the Zero123 model carries no CLIP image tower (`cond_stage_model.model.visual.*` of the checkpoint): pass "
                               "`clip_tower=` or the conditioning embeddings as cond_embeddings_path")
            clip_tower = clip_vit.CLIPVisionTower.from_state_dict(vis)
        wd = torch.float16 if c.half_precision_weights else torch.float32
        clip_tower = clip_tower.to(device=device, dtype=wd).eval()      # (its LayerNorms compute in float32 whatever their storage, guidance :117-134)
        p0 = next(model.first_stage_model.parameters())
        vae_back = None
        if p0.device != torch.device(device) or p0.dtype != wd:
            vae_back = (p0.device, p0.dtype)
            model.first_stage_model.to(device=device, dtype=wd)
        _, cc, ct = clip_vit.prepare_embeddings(model, clip_tower, paths, device, wd)
        if vae_back is not None:
            model.first_stage_model.to(device=vae_back[0], dtype=vae_back[1])
        model.__dict__.pop("_clip_visual_sd", None)
        return cc.cpu(), ct.cpu()
    e = torch.load(c.cond_embeddings_path, map_location="cpu")
    cc, ct = e["c_crossattn"], e["c_concat"]
    if cc.shape[0] < n or ct.shape[0] < n:
        raise ValueError(f"{c.cond_embeddings_path}: {cc.shape[0]} frames of embeddings, the configuration names {n}")
    return cc[:n], ct[:n]


@register("temporal-stable-zero123-guidance")
class TemporalStableZero123Guidance(_z.TemporalStableZero123Guidance, Updateable):
    Config = _TemporalZero123Config
    _frames_key = "num_frames"

    def __init__(self, cfg=None, model=None, clip_tower=None):
        """``model``: an already built ``zero123.Zero123`` (tests, random weights); otherwise the checkpoint is loaded.  ``clip_tower``:
        a ``clip_vit.CLIPVisionTower`` for ``prepare_embeddings`` when ``model`` was not loaded from a checkpoint that carries one."""
        c = parse_structured(self.Config, cfg)
        dev = get_device()
        if model is None:
            model = _zero123_from_config(c.pretrained_config, c.pretrained_model_name_or_path, dev)
        cc, ct = _load_embeddings(c, getattr(c, self._frames_key) if self._frames_key else 1, model=model, device=dev, clip_tower=clip_tower)
        grad_clip = c.grad_clip
        _z.TemporalStableZero123Guidance.__init__(
            self, model, cc, ct, cond_elevation_deg=c.cond_elevation_deg, cond_azimuth_deg=c.cond_azimuth_deg,
            guidance_scale=c.guidance_scale, min_step_percent=_scalar0(c.min_step_percent), max_step_percent=_scalar0(c.max_step_percent),
            grad_clip=None if grad_clip is None else _scalar0(grad_clip), half_precision_weights=c.half_precision_weights)
        self.cfg = c
        self.to(dev)

    def update_step(self, epoch, global_step, on_load_weights=False):
        """guidance :376-388: the scheduled grad clip and the min / max timestep fractions."""
        from .schedule import C

        c = self.cfg
        _z.TemporalStableZero123Guidance.update_step(
            self, epoch, global_step, min_step_percent=C(c.min_step_percent, epoch, global_step),
            max_step_percent=C(c.max_step_percent, epoch, global_step),
            grad_clip=None if c.grad_clip is None else C(c.grad_clip, epoch, global_step))


def _scalar0(v):
    from .schedule import C

    return C(v, 0, 0)


@dataclass
class _StaticZero123Config:                           # StableZero123Guidance.Config (stable_zero123_guidance.py:78-100)
    pretrained_model_name_or_path: str = "load/zero123/stable-zero123.ckpt"
    pretrained_config: str = "load/zero123/sd-objaverse-finetune-c_concat-256.yaml"
    vram_O: bool = True
    cond_image_path: str = "load/images/hamburger_rgba.png"
    cond_elevation_deg: float = 0.0
    cond_azimuth_deg: float = 0.0
    cond_camera_distance: float = 1.2
    guidance_scale: float = 5.0
    grad_clip: Optional[Any] = None
    half_precision_weights: bool = False
    min_step_percent: float = 0.02
    max_step_percent: float = 0.98
    cond_embeddings_path: Optional[str] = None


@register("stable-zero123-guidance")
class StableZero123Guidance(TemporalStableZero123Guidance):
    Config = _StaticZero123Config
    _frames_key = None

    def __call__(self, rgb, elevation, azimuth, camera_distances, rgb_as_latents=False, **kwargs):
        fi = torch.zeros(rgb.shape[0], dtype=torch.long, device=rgb.device)     # one conditioning image
        return _z.TemporalStableZero123Guidance.__call__(self, rgb, elevation, azimuth, camera_distances, frame_indices=fi,
                                                         rgb_as_latents=rgb_as_latents, **kwargs)


# ------------------------------------------------------------------------------------------------ optimiser
def _get_parameters(model, name):
    module = model
    for part in name.split("."):
        module = getattr(module, part)
    if isinstance(module, nn.Module):
        return module.parameters()
    if isinstance(module, nn.Parameter):
        return [module]
    return []


def parse_optimizer(config, model):
    """``parse_optimizer`` (threestudio/systems/utils.py:55-89): ``config.params`` = {dotted module name: group args}
    (else all parameters), ``config.name`` an optimiser of torch.optim, ``config.args`` its arguments."""
    params = config.get("params")
    if params is not None:
        groups = [{"params": list(_get_parameters(model, name)), "name": name, **dict(args)} for name, args in params.items()]
    else:
        groups = list(model.parameters())
    name = config["name"]
    if name in ("FusedAdam", "Adan"):
        raise NotImplementedError(f"optimizer {name} is an external package the shipped configurations do not use")
    return getattr(torch.optim, name)(groups, **dict(config.get("args", {})))


# ------------------------------------------------------------------------------------------------ datamodules, systems
# Thin configuration adapters, NOT Lightning: what `launch.py` does with `data_type` / `system_type` -- find(type)(cfg) -- gives
# an object that holds the parsed block and drives `DynamicStage` / `StaticStage` (the training_step bodies of
# C/system/sugar_4dgen.py:397-429 and C/system/sugar_static.py:110-340).  Image files are outside the path: the frames are
# handed over as arrays (the reference reads `<video_frames_dir>/*.png` with cv2, C/data/temporal_image.py:166-214).
@dataclass
class _TemporalDataConfig:                            # TemporalRandomImageDataModuleConfig (C/data/temporal_image.py:33-70): the keys the path reads
    height: Any = 512
    width: Any = 512
    default_elevation_deg: float = 5.0
    default_azimuth_deg: float = 0.0
    default_camera_distance: float = 3.8
    default_fovy_deg: float = 20.0
    video_length: int = 32
    num_frames: int = 4                               # frames sampled per iteration (yaml:11)
    norm_timestamp: bool = True
    video_frames_dir: Optional[str] = None
    random_camera: Any = None                         # {batch_size, elevation_range, azimuth_range, ...} (yaml:21-48)
    batch_size: int = 1
    requires_depth: bool = False
    requires_normal: bool = False
    use_random_camera: bool = True
    rays_d_normalize: bool = False


@register("temporal-image-datamodule")
class TemporalImageDataModule:
    Config = _TemporalDataConfig

    def __init__(self, cfg=None, frames=None, masks=None):
        """frames [L,H,W,3] in [0,1], masks [L,H,W,1] (float or bool): the video of `video_frames_dir`, already decoded."""
        self.cfg = c = parse_structured(self.Config, {k: v for k, v in dict(cfg or {}).items() if k in {f.name for f in dataclasses.fields(self.Config)}})
        self.height, self.width = _scalar0(c.height), _scalar0(c.width)
        L = int(c.video_length)
        self.frames, self.masks = frames, masks
        if frames is not None and int(frames.shape[0]) != L:
            raise ValueError(f"video_length {L} but {int(frames.shape[0])} frames")
        # timestamps exclude 0 and 1 (C/data/temporal_image.py:155-158)
        self.timestamps = torch.linspace(0, 1, L + 2)[1:-1] if c.norm_timestamp else torch.arange(L, dtype=torch.float32)
        rc = dict(c.random_camera or {})
        self.random_views_per_frame = int(rc.get("batch_size", 1))

    def ref_camera(self):
        from . import synthetic as syn

        c = self.cfg
        return syn.make_camera(int(self.height), int(self.width), elev_deg=c.default_elevation_deg, azim_deg=c.default_azimuth_deg,
                               dist=c.default_camera_distance, fovy_deg=c.default_fovy_deg)


@register("single-image-datamodule")
class SingleImageDataModule(TemporalImageDataModule):
    """C/../threestudio/data/image.py: one reference image = a video of length 1."""

    def __init__(self, cfg=None, image=None, mask=None):
        cfg = dict(cfg or {})
        cfg["video_length"] = 1
        super().__init__(cfg, None if image is None else image.reshape(1, *image.shape[-3:]),
                         None if mask is None else mask.reshape(1, *mask.shape[-3:]))


class _SystemBase:
    """BaseLift3DSystem.configure (threestudio/systems/base.py:262-282): the plugins of a `system:` block by name."""

    def _plugins(self, cfg, guidance_key, model):
        self.cfg = cfg
        self.geometry = find(cfg["geometry_type"])(cfg["geometry"])
        self.material = find(cfg["material_type"])(cfg.get("material"))
        self.background = find(cfg["background_type"])(cfg.get("background"))
        self.renderer = find(cfg["renderer_type"])(cfg.get("renderer"), geometry=self.geometry, material=self.material,
                                                   background=self.background)
        gcfg = cfg.get(guidance_key)
        self.guidance = find(cfg[guidance_key + "_type"])(gcfg, model=model) if gcfg is not None and model is not None else None

    def do_update_step(self, epoch, global_step):
        for m in (self.geometry, self.renderer, self.guidance):
            if m is not None and hasattr(m, "do_update_step"):
                m.do_update_step(epoch, global_step)


@register("sugar-4dgen-system")
class SuGaR4DGen(_SystemBase):
    """C/system/sugar_4dgen.py:28 -- `system_type: sugar-4dgen-system`.  __init__(cfg, data, model): `cfg` the resolved
    `system:` block, `data` a TemporalImageDataModule holding the frames, `model` the Zero123 network (the checkpoint of
    `guidance_zero123.pretrained_model_name_or_path` is not in the tree; None = no SDS term).  training_step() is one
    iteration of `DynamicStage`, with `system.loss`, `system.freq`, `num_inter_frames`, `length_inter_frames` and the
    geometry's learning rates taken from the block."""

    def __init__(self, cfg, data, model=None, **stage_kw):
        from . import mesh_reg, views
        from .dynamic_stage import DynamicStage

        self._plugins(cfg, "guidance_zero123", model)
        g = self.geometry
        dev = g.device
        H, W = int(data.height), int(data.width)
        cam = data.ref_camera()
        self.view_renderer = views.ViewRenderer(g.graph, g.topo, H, W, cam.tanfov, method=g.skinning_method)
        static = {"q_static": g.static_quaternions, "scales": g.get_scaling, "opacities": g.get_opacity, "rgb": g.get_points_rgb()}
        loss = dict(cfg.get("loss", {}))
        faces = g.get_faces.detach().cpu().numpy()
        verts = g.get_xyz_verts.detach().cpu().numpy()
        nc = mesh_reg.MeshNormalConsistency(faces, g.n_verts, dev) if _nonzero(loss.get("lambda_normal_consistency")) else None
        arap = mesh_reg.ARAPCoach(verts, faces, dev) if _nonzero(loss.get("lambda_arap_reg_key_frame")) or \
            _nonzero(loss.get("lambda_arap_reg_inter_frame")) else None
        self.stage = DynamicStage.from_cfg(cfg, self.view_renderer, g._deformation, g._deform_graph_node_xyz, static,
                                           data.timestamps.to(dev), data.frames.to(dev), data.masks.to(dev).float(), cam,
                                           guidance=self.guidance, frames_per_step=int(data.cfg.num_frames),
                                           random_views_per_frame=data.random_views_per_frame, normal_consistency=nc, arap=arap,
                                           **stage_kw)

    def training_step(self):
        self.do_update_step(0, self.stage.global_step)
        return self.stage.iteration()


@register("sugar-static-system")
class SuGaRStatic(_SystemBase):
    """C/system/sugar_static.py -- `system_type: sugar-static-system` (stage "sugar"): one StaticStage iteration per
    training_step, loss weights from `system.loss`."""

    def __init__(self, cfg, data, model=None, **stage_kw):
        from . import mesh_reg
        from .static_stage import StaticStage

        self._plugins(cfg, "guidance", model)
        g = self.geometry
        dev = g.device
        H, W = int(data.height), int(data.width)
        loss = dict(cfg.get("loss", {}))
        faces = g.get_faces.detach().cpu().numpy()
        nc = mesh_reg.MeshNormalConsistency(faces, g.n_verts, dev) if _nonzero(loss.get("lambda_normal_consistency")) else None
        lap = mesh_reg.MeshLaplacianSmoothing(faces, g.n_verts, dev) if _nonzero(loss.get("lambda_laplacian_smoothing")) else None
        rc = dict(getattr(data.cfg, "random_camera", None) or {})
        self.stage = StaticStage.from_cfg(cfg, g, self.renderer, data.frames.to(dev), data.masks.to(dev).float(), H, W,
                                          guidance=self.guidance, random_views=int(rc.get("batch_size", 4)),
                                          normal_consistency=nc, laplacian_smoothing=lap, **stage_kw)

    def training_step(self):
        self.do_update_step(0, self.stage.global_step)
        return self.stage.iteration()


def _nonzero(v):
    if v is None:
        return False
    if isinstance(v, (list, tuple)):
        return any(float(x) != 0.0 for x in v[1:3])
    return float(v) != 0.0
