"""threestudio plugin names of the hot path (boundary B1).

The reference binds its YAML configs to classes registered under these names
(``@threestudio.register(name)``, threestudio/__init__.py:5-32) and constructs them as ``cls(cfg, *args)``:

    diff-sugar-rasterizer-temporal      C/renderer/diff_sugar_rasterizer_temporal.py:56
    diff-sugar-rasterizer-normal        C/renderer/diff_sugar_rasterizer_normal.py:54
    dynamic-sugar                       C/geometry/dynamic_sugar.py:42
    sugar                               C/geometry/sugar.py:33
    temporal-stable-zero123-guidance    C/guidance/temporal_stable_zero123_guidance.py:76
    stable-zero123-guidance             threestudio/models/guidance/stable_zero123_guidance.py:75
    solid-color-background              threestudio/models/background/solid_color_background.py:14
    no-material                         threestudio/models/materials/no_material.py:16
    sugar-4dgen-system                  C/system/sugar_4dgen.py:28          (thin cfg adapter over DynamicStage, not Lightning)
    sugar-static-system                 C/system/sugar_static.py            (thin cfg adapter over StaticStage)
    temporal-image-datamodule           C/data/temporal_image.py:483        (frames handed over as arrays)
    single-image-datamodule             threestudio/data/image.py

``PLUGINS`` maps every name to its cfg-constructed class (dreammesh4d_amd/threestudio_host.py: the ``Config``
dataclasses of the reference, ``configure``, ``update_step``); ``threestudio_host.find(name)`` is the registry without
threestudio; ``register(threestudio_module)`` enters the classes into a real threestudio registry under the
REFERENCE's names (prefix "" -- they replace the CUDA-backed plugins; pass a prefix to let both coexist).
"""
from . import threestudio_host as host

PLUGINS = dict(host.__modules__)


def register(threestudio=None, prefix=""):
    """Registers the classes as ``<prefix><name>`` in `threestudio` (imported when not given).  Returns the names."""
    if threestudio is None:
        import threestudio  # noqa: F811  (only when the caller has it)
    names = []
    for name, cls in PLUGINS.items():
        threestudio.register(prefix + name)(cls)
        names.append(prefix + name)
    return names
