"""dreammesh4d_amd -- MI355X-native hot path for DreamMesh4D's dynamic stage.

See DESIGN.md / INTEGRATION.md.  The compute path is libdm4d_hip.so (hand-written HIP for gfx950,
C ABI declared in include/dm4d.h); this package is the Python host that mirrors the reference's
operator / plugin interface for that path.  There is no CPU fallback.
"""
import sys

__version__ = "0.1.0"


def install_compat():
    """Register the drop-in operator modules under the names the reference imports
    (`diff_gaussian_rasterization`, `simple_knn`, `simple_knn._C`) so
    custom/threestudio-dreammesh4d/{renderer,geometry}/*.py import them unmodified."""
    from . import diff_gaussian_rasterization as dgr
    from . import simple_knn as sknn

    sys.modules["diff_gaussian_rasterization"] = dgr
    sys.modules["simple_knn"] = sknn
    sys.modules["simple_knn._C"] = sknn._C
    return dgr, sknn
