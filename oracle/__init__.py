"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference algorithms on DreamMesh4D's dynamic-stage
hot path (rasterizer, simple-knn, skinning, face->Gaussian).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product (``dreammesh4d_amd/``) never does and fails loudly
when its HIP library is missing.

Parity status (see DESIGN.md section "Oracle"):
  * rasterizer, simple-knn : PARITY UNPINNED -- algorithm lives in un-vendored,
    un-pinned third-party CUDA packages; the reference holds no tests/vectors.
  * skinning / face->Gaussian : PARITY UNPINNED for the pypose/pytorch3d
    conventions (packages absent); pinned by closed-form identities.
  * HexPlane deformation network : pinned by golden vectors generated from the
    reference's own ``geometry/deformation.py`` (tests/golden/, script committed).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    """Compile liboracle (gcc).  Building the checker is not using it."""
    so = os.path.join(_HERE, "libdm4d_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("raster_oracle.c", "knn_oracle.c", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libdm4d_oracle.so"])
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libdm4d_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
    return _LIB
