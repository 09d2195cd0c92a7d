"""ctypes front-end of oracle/raster_oracle.c (TEST INFRASTRUCTURE ONLY).

Mirrors the operator boundary the reference binds at
custom/threestudio-dreammesh4d/renderer/diff_sugar_rasterizer_temporal.py:129-178
(GaussianRasterizationSettings + GaussianRasterizer call), with numpy arrays.
"""
import ctypes as C
import math

import numpy as np

from . import lib

_f = C.POINTER(C.c_float)


class _In(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("W", C.c_int32), ("H", C.c_int32), ("sh_degree", C.c_int32), ("prefiltered", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("bg", _f), ("means3D", _f), ("colors_precomp", _f), ("shs", _f), ("opacities", _f), ("scales", _f),
        ("rotations", _f), ("cov3D_precomp", _f), ("viewmatrix", _f), ("projmatrix", _f), ("campos", _f),
    ]


class _State(C.Structure):
    _fields_ = [
        ("out_color", _f), ("out_depth", _f), ("out_alpha", _f), ("radii", C.POINTER(C.c_int32)),
        ("xy", _f), ("depths", _f), ("conic_opacity", _f), ("rgb", _f), ("cov3D", _f),
        ("clamped", C.POINTER(C.c_uint8)), ("tiles_touched", C.POINTER(C.c_uint32)),
        ("keys", C.POINTER(C.c_uint64)), ("values", C.POINTER(C.c_uint32)), ("cap", C.c_int64),
        ("ranges", C.POINTER(C.c_uint32)), ("n_contrib", C.POINTER(C.c_uint32)), ("final_T", _f),
    ]


class _Grads(C.Structure):
    _fields_ = [(n, _f) for n in ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_ddepths",
                                  "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drots")]


def _p(a, t=_f):
    return None if a is None else a.ctypes.data_as(t)


def _c(a, dtype=np.float32):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


class RasterOracle:
    """One forward (and optionally backward) of the CPU oracle; keeps all state as numpy arrays."""

    def __init__(self, *, image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix, projmatrix,
                 sh_degree=0, campos=None, prefiltered=False):
        self.H, self.W = int(image_height), int(image_width)
        self.tanfovx, self.tanfovy = float(tanfovx), float(tanfovy)
        self.bg = _c(bg).reshape(3)
        self.scale_modifier = float(scale_modifier)
        self.view = _c(viewmatrix).reshape(16)
        self.proj = _c(projmatrix).reshape(16)
        self.campos = _c(campos if campos is not None else np.zeros(3)).reshape(3)
        self.sh_degree = int(sh_degree)
        self.prefiltered = bool(prefiltered)

    def forward(self, means3D, opacities, *, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        L = lib()
        L.dm4d_oracle_rasterize_forward.restype = C.c_int64
        self.means3D = _c(means3D).reshape(-1, 3)
        N = self.N = self.means3D.shape[0]
        self.opac = _c(opacities).reshape(N)
        self.shs = _c(shs)
        self.colors = _c(colors_precomp)
        self.scales = _c(scales)
        self.rots = _c(rotations)
        self.cov3Dp = _c(cov3D_precomp)
        assert (self.shs is None) != (self.colors is None)
        assert (self.cov3Dp is None) != (self.scales is None)
        H, W = self.H, self.W
        gx, gy = (W + 15) // 16, (H + 15) // 16
        inp = self._in = _In(N, W, H, self.sh_degree, int(self.prefiltered), self.tanfovx, self.tanfovy,
                             self.scale_modifier, _p(self.bg), _p(self.means3D), _p(self.colors), _p(self.shs),
                             _p(self.opac), _p(self.scales), _p(self.rots), _p(self.cov3Dp), _p(self.view),
                             _p(self.proj), _p(self.campos))
        s = self.s = {
            "out_color": np.zeros((3, H, W), np.float32), "out_depth": np.zeros((H, W), np.float32),
            "out_alpha": np.zeros((H, W), np.float32), "radii": np.zeros(N, np.int32),
            "xy": np.zeros((N, 2), np.float32), "depths": np.zeros(N, np.float32),
            "conic_opacity": np.zeros((N, 4), np.float32), "rgb": np.zeros((N, 3), np.float32),
            "cov3D": np.zeros((N, 6), np.float32), "clamped": np.zeros((N, 3), np.uint8),
            "tiles_touched": np.zeros(N, np.uint32), "ranges": np.zeros((gx * gy, 2), np.uint32),
            "n_contrib": np.zeros((H, W), np.uint32), "final_T": np.zeros((H, W), np.float32),
        }

        def mk(cap):
            s["keys"] = np.zeros(max(cap, 1), np.uint64)
            s["values"] = np.zeros(max(cap, 1), np.uint32)
            return _State(_p(s["out_color"]), _p(s["out_depth"]), _p(s["out_alpha"]),
                          _p(s["radii"], C.POINTER(C.c_int32)), _p(s["xy"]), _p(s["depths"]), _p(s["conic_opacity"]),
                          _p(s["rgb"]), _p(s["cov3D"]), _p(s["clamped"], C.POINTER(C.c_uint8)),
                          _p(s["tiles_touched"], C.POINTER(C.c_uint32)), _p(s["keys"], C.POINTER(C.c_uint64)),
                          _p(s["values"], C.POINTER(C.c_uint32)), cap, _p(s["ranges"], C.POINTER(C.c_uint32)),
                          _p(s["n_contrib"], C.POINTER(C.c_uint32)), _p(s["final_T"]))

        st = mk(0)
        D = L.dm4d_oracle_rasterize_forward(C.byref(inp), C.byref(st))
        if D < 0:
            raise RuntimeError(f"oracle forward failed ({D})")
        st = mk(int(D))
        D2 = L.dm4d_oracle_rasterize_forward(C.byref(inp), C.byref(st))
        assert D2 == D
        self._st = st
        self.D = int(D)
        s["keys"] = s["keys"][:self.D]
        s["values"] = s["values"][:self.D]
        return s["out_color"], s["radii"], s["out_depth"], s["out_alpha"]

    def backward(self, dL_dcolor, dL_ddepth=None, dL_dalpha=None):
        L = lib()
        N = self.N
        g = self.g = {
            "dL_dmeans2D": np.zeros((N, 3), np.float32), "dL_dconic": np.zeros((N, 3), np.float32),
            "dL_dopacity": np.zeros(N, np.float32), "dL_dcolors": np.zeros((N, 3), np.float32),
            "dL_ddepths": np.zeros(N, np.float32), "dL_dmeans3D": np.zeros((N, 3), np.float32),
            "dL_dcov3D": np.zeros((N, 6), np.float32),
            "dL_dsh": np.zeros((N, 1, 3), np.float32) if self.shs is not None else None,
            "dL_dscales": np.zeros((N, 3), np.float32) if self.scales is not None else None,
            "dL_drots": np.zeros((N, 4), np.float32) if self.rots is not None else None,
        }
        gs = _Grads(*[_p(g[n]) for n, _ in _Grads._fields_])
        dc = _c(dL_dcolor).reshape(3, self.H, self.W)
        dd = _c(dL_ddepth)
        da = _c(dL_dalpha)
        rc = L.dm4d_oracle_rasterize_backward(C.byref(self._in), C.byref(self._st), C.c_int64(self.D), _p(dc), _p(dd),
                                              _p(da), C.byref(gs))
        if rc != 0:
            raise RuntimeError("oracle backward failed")
        return g


    def preprocess_backward(self, rel_noise=0.0, seed=0):
        """Stage 2 alone (conic / mean2D / depth / colour gradients -> parameter gradients) from the blend-level
        gradients of the last backward(), each multiplied by (1 + rel_noise * N(0,1)) first.  Returns a new dict;
        self.g is untouched."""
        L = lib()
        rng = np.random.default_rng(seed)
        g = {k: (None if v is None else v.copy()) for k, v in self.g.items()}
        if rel_noise:
            for k in ("dL_dmeans2D", "dL_dconic", "dL_dcolors", "dL_ddepths"):
                g[k] = (g[k] * (1.0 + rel_noise * rng.standard_normal(g[k].shape))).astype(np.float32)
        gs = _Grads(*[_p(g[n]) for n, _ in _Grads._fields_])
        if L.dm4d_oracle_preprocess_backward(C.byref(self._in), C.byref(self._st), C.byref(gs)) != 0:
            raise RuntimeError("oracle preprocess backward failed")
        return g


def expf(x):
    L = lib()
    L.dm4d_oracle_expf.restype = C.c_float
    L.dm4d_oracle_expf.argtypes = [C.c_float]
    return L.dm4d_oracle_expf(float(x))


def dist2_knn3(points, brute=False):
    L = lib()
    pts = _c(points).reshape(-1, 3)
    out = np.zeros(pts.shape[0], np.float32)
    fn = L.dm4d_oracle_dist2_brute if brute else L.dm4d_oracle_dist2_knn3
    fn(C.c_int(pts.shape[0]), _p(pts), _p(out))
    return out


def camera_matrices(c2w, fovy, znear=0.1, zfar=100.0):
    """numpy restatement of threestudio/utils/ops.py:359-413 (get_cam_info_gaussian,
    get_projection_matrix_gaussian, convert_pose).  Returns (world_view_transform^T-convention,
    full_proj_transform, camera_center), float32."""
    c2w = np.asarray(c2w, np.float64).copy()
    if c2w.shape == (3, 4):
        c2w = np.vstack([c2w, [0, 0, 0, 1]])
    flip = np.diag([1.0, -1.0, -1.0, 1.0])
    c2w = c2w @ flip
    w2c = np.linalg.inv(c2w)
    wvt = w2c.T.astype(np.float32)
    t = math.tan(fovy / 2)
    top, right = t * znear, t * znear
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    proj = P.T.copy()
    full = (wvt @ proj).astype(np.float32)
    cam = np.linalg.inv(wvt.astype(np.float64))[3, :3].astype(np.float32)
    return wvt, full, cam
