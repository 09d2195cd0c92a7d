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


def test_round4_operators_have_no_cpu_path_either():
    """The batched static renderer, the static image head and the step object refuse CPU tensors (no fallback), and the new C entry
    points reject bad arguments before touching a device."""
    import ctypes as C

    import torch

    from dreammesh4d_amd import _lib, gviews, static_head

    with pytest.raises(_lib.Dm4dError, match="no CPU fallback"):
        gviews.GaussianViews(100, 64, 64, 0.2, "cpu")
    z = torch.zeros(2, 6, 8, 8)
    with pytest.raises(ValueError, match="float32 HIP tensors"):
        static_head.static_head(z, z[:, :1], z[:, :1], torch.zeros(2, dtype=torch.int32), torch.zeros(2, dtype=torch.int32),
                                torch.zeros(1, 8, 8, 3), torch.zeros(1, 8, 8, 1), torch.zeros(1, dtype=torch.int64), 1, 1)
    L = _lib.lib()
    dummy = (C.c_float * 64)()
    p = C.cast(dummy, C.c_void_p)
    assert L.dm4d_static_head_forward(2, 7, 8, p, p, p, p, p, p, p, p, 1, 1, p, p, None) == -4            # odd height: DM4D_ERR_UNSUPPORTED
    assert b"H and W must be even" in L.dm4d_last_error()
    assert L.dm4d_sugar_attributes_forward(10, 5, 30, p, p, p, p, p, p, p, 1e-6, 1.0, p, p, p, p, p, None) == -1      # G = 5
    assert L.dm4d_sds_prepare(0, 32, 32, 0.18, p, p, p, p, p, p, p, p, p, p, p, p, p, p, p, p, None) == -1
    gv = _lib.GViewsStruct()
    gv.B, gv.N, gv.image_height, gv.image_width = 0, 10, 64, 64
    assert L.dm4d_gviews_forward(C.byref(gv), None) == -1 and b"bad batch size" in L.dm4d_last_error()
    assert L.dm4d_step_forward(None, p, p, p, None, None) == -1 and b"not a step object" in L.dm4d_last_error()
    args = _lib.AdamwArgs()
    seg = _lib.GradSegments()
    assert L.dm4d_adamw_message(C.byref(seg), C.byref(args), 1.0, None) == -1 and b"null state" in L.dm4d_last_error()


def test_argument_validation_is_host_side():
    """Entry points reject bad arguments before touching the device (so this runs without a GPU): the rasterizer's limits,
    the GroupNorm / pointwise operators' shape rules; the message is available through dm4d_last_error."""
    import ctypes as C

    from dreammesh4d_amd import _lib

    L = _lib.lib()
    dummy = (C.c_float * 64)()
    p = C.cast(dummy, C.c_void_p)
    s = _lib.RasterSettings(64, 64, 0.2, 0.2, 1.0, 0, 0, 0, p, p, p, p)
    too_many = _lib.RasterInputs((1 << 25) + 1, 0, 3, p, None, p, p, p, p, None)
    assert L.dm4d_rasterize_prepare(C.byref(s), C.byref(too_many), p, p, 64, None) == -4            # DM4D_ERR_UNSUPPORTED
    assert b"at most 33554432 Gaussians" in L.dm4d_last_error()
    bad_size = _lib.RasterSettings(0, 64, 0.2, 0.2, 1.0, 0, 0, 0, p, p, p, p)
    ok_in = _lib.RasterInputs(4, 0, 3, p, None, p, p, p, p, None)
    assert L.dm4d_rasterize_prepare(C.byref(bad_size), C.byref(ok_in), p, p, 64, None) == -1        # DM4D_ERR_INVALID
    both = _lib.RasterInputs(4, 1, 3, p, p, p, p, p, p, None)                                        # shs AND colors_precomp
    assert L.dm4d_rasterize_prepare(C.byref(s), C.byref(both), p, p, 64, None) == -1
    assert L.dm4d_rasterize_prepare(C.byref(s), C.byref(ok_in), p, p, 64, None) == -3               # workspace too small: DM4D_ERR_CAPACITY
    assert b"geom workspace too small" in L.dm4d_last_error()
    # GroupNorm / pointwise operators
    assert L.dm4d_groupnorm_nhwc_forward(1, 4, 6, 4, 1, p, None, 0, p, p, 1e-5, 0, p, p, p, 1, None) == -1      # C % G
    assert L.dm4d_groupnorm_nhwc_forward(1, 4, 8, 4, 1, p, p, 3, p, p, 1e-5, 0, p, p, p, 1, None) == -1         # add_stride
    assert L.dm4d_groupnorm_nhwc_backward(1, 4, 8, 4, 1, p, None, 0, p, p, p, 0, None, p, p, 1, None) == -1     # dy missing
    assert L.dm4d_add_bias_nhwc(4, 6, 1, p, p, p, p, None) == -1 and L.dm4d_geglu(4, 6, 0, p, p, None) == -1
    assert L.dm4d_add_bias_nhwc(0, 8, 1, None, None, None, None, None) == 0 and L.dm4d_geglu(0, 8, 1, None, None, None) == 0


def test_conv_plan_scratch_sizes_follow_the_split_rule():
    """dm4d_conv3x3_scratch_bytes is host-only: the float32 partial sums of the split-K launches.  The rule (csrc/conv_mfma.hip::
    conv_plan): split only until every CU has one workgroup and never below 30 k-tiles per workgroup; tiles of 128 x 128 for the
    implicit GEMM, of 256 x 128 for the direct kernel, which takes W >= 64 and the narrower power-of-two images from 14 GFLOP."""
    from dreammesh4d_amd import _lib

    L = _lib.lib()
    part = lambda splits, N, H, W, Co: splits * N * H * W * Co * 4 + 256
    assert L.dm4d_conv3x3_scratch_bytes(8, 4, 4, 1280, 1280) == part(12, 8, 4, 4, 1280)       # implicit GEMM (W < 8): 10 tiles, 360 k-tiles: 360 / 30
    assert L.dm4d_conv3x3_scratch_bytes(8, 4, 4, 2560, 1280) == part(24, 8, 4, 4, 1280)
    assert L.dm4d_conv3x3_scratch_bytes(8, 8, 8, 640, 1280) == part(6, 8, 8, 8, 1280)         # implicit GEMM (7.5 GFLOP): 40 tiles: 256 / 40
    assert L.dm4d_conv3x3_scratch_bytes(8, 16, 16, 320, 640) == part(3, 8, 16, 16, 640)       # implicit GEMM: 80 tiles, 90 k-tiles
    assert L.dm4d_conv3x3_scratch_bytes(8, 8, 8, 1280, 1280) == part(12, 8, 8, 8, 1280)       # direct (15.1 GFLOP): 20 tiles of 256 x 128, 360 / 30
    assert L.dm4d_conv3x3_scratch_bytes(8, 16, 16, 640, 640) == part(6, 8, 16, 16, 640)       # direct: 40 tiles
    assert L.dm4d_conv3x3_scratch_bytes(8, 32, 32, 320, 320) == part(2, 8, 32, 32, 320)       # direct: 96 tiles: 256 / 96
    assert L.dm4d_conv3x3_scratch_bytes(4, 256, 256, 128, 128) == 256                          # direct, 1024 tiles: no split
    assert L.dm4d_conv3x3_scratch_bytes(0, 4, 4, 32, 32) == 256


def test_blocked_triangular_inverse_of_the_dense_graph_solver():
    """graph_build._tri_inverse (block recursion, all GEMM) against torch's inverse, on the CPU (pure torch): sizes below and
    above the recursion's block size, not a multiple of its 256-row alignment."""
    import torch

    from dreammesh4d_amd.graph_build import _tri_inverse

    g = torch.Generator().manual_seed(4)
    for n, nb in ((37, 1024), (700, 128), (1030, 256)):
        A = torch.randn(n, n, generator=g, dtype=torch.float64)
        Lc = torch.linalg.cholesky(A @ A.T + n * torch.eye(n, dtype=torch.float64))
        Li = _tri_inverse(Lc, nb=nb)
        assert float((Li @ Lc - torch.eye(n, dtype=torch.float64)).abs().max()) < 1e-12
        assert float(torch.triu(Li, 1).abs().max()) == 0.0


def test_round3_entry_points_validate_on_the_host():
    """dm4d_linear_f16, dm4d_conv3x3_s2_dgrad_nhwc_f16, dm4d_quat_to_matrix_* reject what they do not take before any launch (no GPU
    needed): the error codes of include/dm4d.h and a message in dm4d_last_error; an empty problem is DM4D_OK."""
    import ctypes as C

    L = _lib.lib()
    buf = (C.c_float * 256)()
    p = C.cast(buf, C.c_void_p)                                   # (16-byte aligned: ctypes arrays of this size are)
    assert p.value % 16 == 0
    # linear: K % 32, N % 8, GEGLU's N % 128 / no residual, act range, null, empty
    assert L.dm4d_linear_f16(4, 48, 8, p, p, None, None, p, 0, None, None) == -4 and b"multiple of 32" in L.dm4d_last_error()
    assert L.dm4d_linear_f16(4, 64, 12, p, p, None, None, p, 0, None, None) == -4
    assert L.dm4d_linear_f16(4, 64, 64, p, p, None, None, p, 1, None, None) == -4 and b"GEGLU" in L.dm4d_last_error()
    assert L.dm4d_linear_f16(4, 64, 128, p, p, None, p, p, 1, None, None) == -4
    assert L.dm4d_linear_f16(4, 64, 64, p, p, None, None, p, 2, None, None) == -1
    assert L.dm4d_linear_f16(4, 64, 64, None, p, None, None, p, 0, None, None) == -1
    assert L.dm4d_linear_f16(4, 64, 64, C.c_void_p(p.value + 2), p, None, None, p, 0, None, None) == -1 and b"aligned" in L.dm4d_last_error()
    assert L.dm4d_linear_f16(0, 64, 64, None, None, None, None, None, 0, None, None) == 0
    assert L.dm4d_linear_scratch_bytes(8192, 320, 320) == 256                                   # 192 tiles of 128 x 128 / 768 of 64 x 64: no split
    assert L.dm4d_linear_scratch_bytes(512, 5120, 1280) == 5 * 512 * 1280 * 4 + 256            # 40 tiles, 160 k-tiles: 160 / 30
    # stride-2 data gradient: even sizes, channel multiples, the four filters
    w4 = (C.c_void_p * 4)(p, p, p, p)
    assert L.dm4d_conv3x3_s2_dgrad_nhwc_f16(1, 15, 16, 32, 32, p, w4, p, None) == -4 and b"even" in L.dm4d_last_error()
    assert L.dm4d_conv3x3_s2_dgrad_nhwc_f16(1, 16, 16, 24, 32, p, w4, p, None) == -4
    assert L.dm4d_conv3x3_s2_dgrad_nhwc_f16(1, 16, 16, 32, 32, p, (C.c_void_p * 4)(p, p, None, p), p, None) == -1
    assert L.dm4d_conv3x3_s2_dgrad_nhwc_f16(0, 16, 16, 32, 32, None, None, None, None) == 0
    # quaternion -> matrix
    assert L.dm4d_quat_to_matrix_forward(-1, p, p, None) == -1
    assert L.dm4d_quat_to_matrix_forward(4, None, p, None) == -1
    assert L.dm4d_quat_to_matrix_backward_pypose(4, p, p, None, None) == -1
    assert L.dm4d_quat_to_matrix_forward(0, None, None, None) == 0 and L.dm4d_quat_to_matrix_backward_pypose(0, None, None, None, None) == 0


def test_bench_gpus_flag_is_honoured_or_refused():
    """`python bench.py --gpus N` with N > 1 and no launcher starts its own ranks (tests/test_dynamic_stage_gpu.py); where fewer HIP
    devices than ranks are visible -- here: none -- it must refuse loudly instead of rendering on one device and printing n_gpus = 1
    (round 4's behaviour: `--gpus` was parsed and never read)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DM4D_BENCH_BACKEND")}
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode != 0 and "HIP device" in (r.stdout + r.stderr)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "0"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode != 0
