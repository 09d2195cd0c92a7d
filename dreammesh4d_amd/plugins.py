"""threestudio plugin names of the hot path (boundary B1).

The reference binds its YAML configs to classes registered under these names
(``@threestudio.register(name)``, threestudio/__init__.py:5-32):

    diff-sugar-rasterizer-temporal      C/renderer/diff_sugar_rasterizer_temporal.py:56
    diff-sugar-rasterizer-normal        C/renderer/diff_sugar_rasterizer_normal.py:54
    dynamic-sugar                       C/geometry/dynamic_sugar.py:42
    sugar                               C/geometry/sugar.py:33
    temporal-stable-zero123-guidance    C/guidance/temporal_stable_zero123_guidance.py:76
    stable-zero123-guidance             threestudio/models/guidance/stable_zero123_guidance.py:75
    solid-color-background              threestudio/models/background/solid_color_background.py:14
    no-material                         threestudio/models/materials/no_material.py:16

``PLUGINS`` maps every name to the class of this package that implements its hot-path surface;
``register(threestudio_module)`` enters them into a threestudio registry (when threestudio is importable,
``register()`` with no argument imports it).  threestudio's launcher / trainer / config parsing are out of
scope, so the classes take plain constructor arguments instead of a ``cfg`` dataclass: a
maintainer wires ``cfg`` fields to them in a three-line subclass.
"""
from .renderer import DiffGaussianTemporal, DiffSuGaRNormal
from .shims import NoMaterial, SolidColorBackground
from .sugar import DynamicSuGaR, SuGaR
from .zero123 import StableZero123Guidance, TemporalStableZero123Guidance

PLUGINS = {
    "diff-sugar-rasterizer-temporal": DiffGaussianTemporal,
    "diff-sugar-rasterizer-normal": DiffSuGaRNormal,
    "dynamic-sugar": DynamicSuGaR,
    "sugar": SuGaR,
    "temporal-stable-zero123-guidance": TemporalStableZero123Guidance,
    "stable-zero123-guidance": StableZero123Guidance,
    "solid-color-background": SolidColorBackground,
    "no-material": NoMaterial,
}


def register(threestudio=None, prefix="dm4d-"):
    """Registers the classes as ``<prefix><name>`` (default prefix so that they can coexist with the reference's
    own CUDA-backed plugins; pass prefix="" to take the reference's names over).  Returns the registered names."""
    if threestudio is None:
        import threestudio  # noqa: F811  (only when the caller has it)
    names = []
    for name, cls in PLUGINS.items():
        threestudio.register(prefix + name)(cls)
        names.append(prefix + name)
    return names
