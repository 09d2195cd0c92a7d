"""CPU-side checks of the C-ABI library: it loads without a GPU and exports every symbol
include/dm4d.h declares (no compute calls here)."""
import ctypes

import pytest

from dreammesh4d_amd import _lib


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    declared = _lib.declared_symbols()
    assert len(declared) >= 15
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    # the ctypes signature table covers the same set
    assert sorted(_lib._SIGNATURES) == declared
    assert L.dm4d_version() >= 100


def test_workspace_size_queries_are_monotone_and_aligned():
    L = _lib.lib()
    a = L.dm4d_raster_geom_bytes(1000, 256, 256)
    b = L.dm4d_raster_geom_bytes(200_004, 512, 512)
    assert 0 < a < b and a % 256 == 0 and b % 256 == 0
    assert L.dm4d_raster_binning_bytes(0) > 0
    assert L.dm4d_raster_binning_bytes(10) <= L.dm4d_raster_binning_bytes(1_000_000)
    assert L.dm4d_raster_grad_bytes(1_000_000, 3) >= 1_000_000 * 40     # 10 floats per (Gaussian, cell) record
    assert L.dm4d_raster_grad_bytes(1_000_000, 6) > L.dm4d_raster_grad_bytes(1_000_000, 3)
    assert L.dm4d_raster_image_bytes(512, 512) >= 512 * 512 * 8


def test_argument_validation_without_a_device():
    L = _lib.lib()
    s = _lib.RasterSettings(0, 0, 1.0, 1.0, 1.0, 0, 0, 0, None, None, None, None)
    i = _lib.RasterInputs(0, 0, 3, None, None, None, None, None, None, None)
    rc = L.dm4d_rasterize_prepare(ctypes.byref(s), ctypes.byref(i), None, None, 0, None)
    assert rc == -1 and b"image size" in L.dm4d_last_error()
    with pytest.raises(_lib.Dm4dError):
        _lib.check(rc, "prepare")


def test_product_has_no_cpu_fallback():
    import torch

    import dreammesh4d_amd.diff_gaussian_rasterization as dgr

    z = torch.zeros(4, 3)
    rs = dgr.GaussianRasterizationSettings(16, 16, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                           torch.zeros(3), False, False)
    with pytest.raises(RuntimeError, match="no CPU path"):
        dgr.GaussianRasterizer(rs)(means3D=z, means2D=z, opacities=torch.ones(4, 1), colors_precomp=z,
                                   scales=z + 1, rotations=torch.zeros(4, 4))
