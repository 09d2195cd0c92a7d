"""Drop-in for the ``simple_knn`` package (DSaurus fork) imported at
custom/threestudio-dreammesh4d/geometry/sugar.py:11, dynamic_sugar.py:10, gaussian_base.py:25."""
from . import _C  # noqa: F401
