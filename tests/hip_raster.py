"""Test helper: drive the rasterizer through the C ABI (include/dm4d.h) stage by stage and read
back every piece of state the parity tests compare with the oracle."""
import ctypes as C

import numpy as np
import torch

from dreammesh4d_amd import _lib


def _p(t):
    return None if t is None else t.data_ptr()


class HipRaster:
    def __init__(self, cam, bg=(1, 1, 1), scale_mod=1.0, device="cuda:0"):
        self.dev = torch.device(device)
        self.cam = cam
        t = lambda a: torch.tensor(np.asarray(a, np.float32), device=self.dev).contiguous()
        self.bg, self.view, self.proj, self.campos = t(bg), t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos)
        self.scale_mod = scale_mod
        self.L = _lib.lib()

    def forward(self, means3D, opacities, colors=None, shs=None, scales=None, rotations=None, cov3D=None,
                capacity=None):
        L, dev, cam = self.L, self.dev, self.cam
        t = lambda a: None if a is None else torch.tensor(np.asarray(a, np.float32), device=dev).contiguous()
        self.t = dict(m=t(means3D), o=t(opacities), c=t(colors), sh=t(shs), s=t(scales), r=t(rotations), cov=t(cov3D))
        N = self.N = int(self.t["m"].shape[0])
        self.C = 3 if colors is None else int(np.asarray(colors).shape[-1])
        H, W = cam.H, cam.W
        M = 0 if shs is None else int(self.t["sh"].shape[1])
        self.settings = _lib.RasterSettings(H, W, cam.tanfov, cam.tanfov, self.scale_mod, 0, 0, 0, _p(self.bg),
                                            _p(self.view), _p(self.proj), _p(self.campos))
        self.inputs = _lib.RasterInputs(N, M, self.C, _p(self.t["m"]) if N else None, _p(self.t["sh"]), _p(self.t["c"]),
                                        _p(self.t["o"]) if N else None, _p(self.t["s"]), _p(self.t["r"]),
                                        _p(self.t["cov"]))
        st = torch.cuda.current_stream(dev).cuda_stream
        self.color = torch.empty(self.C, H, W, device=dev)
        self.depth = torch.empty(H, W, device=dev)
        self.alpha = torch.empty(H, W, device=dev)
        self.radii = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
        gb = L.dm4d_raster_geom_bytes(N, H, W)
        self.geom = torch.empty(gb, dtype=torch.uint8, device=dev)
        _lib.check(L.dm4d_rasterize_prepare(self.settings, self.inputs, _p(self.radii), _p(self.geom), gb, st), "prepare")
        self.D = _lib.check(L.dm4d_rasterize_num_rendered(_p(self.geom), st), "num_rendered")
        self.R = _lib.check(L.dm4d_rasterize_num_records(_p(self.geom), st), "num_records")
        self.cap = self.D if capacity is None else capacity
        self.binning = torch.empty(L.dm4d_raster_binning_bytes(self.cap), dtype=torch.uint8, device=dev)
        self.image = torch.empty(L.dm4d_raster_image_bytes(H, W), dtype=torch.uint8, device=dev)
        _lib.check(L.dm4d_rasterize_render(self.settings, self.inputs, _p(self.radii), _p(self.geom), _p(self.binning),
                                           self.cap, _p(self.image), _p(self.color), _p(self.depth), _p(self.alpha),
                                           st), "render")
        torch.cuda.synchronize(dev)
        return self.color.cpu().numpy(), self.radii[:N].cpu().numpy(), self.depth.cpu().numpy(), self.alpha.cpu().numpy()

    def overflowed(self):
        return self.L.dm4d_rasterize_overflowed(_p(self.geom), torch.cuda.current_stream(self.dev).cuda_stream)

    def state(self):
        L, N, H, W = self.L, self.N, self.cam.H, self.cam.W
        st = torch.cuda.current_stream(self.dev).cuda_stream
        T = ((W + 15) // 16) * ((H + 15) // 16)
        s = {"xy": np.zeros((max(N, 1), 2), np.float32), "depths": np.zeros(max(N, 1), np.float32),
             "conic_opacity": np.zeros((max(N, 1), 4), np.float32), "tiles_touched": np.zeros(max(N, 1), np.uint32),
             "keys": np.zeros(max(self.D, 1), np.uint64), "values": np.zeros(max(self.D, 1), np.uint32),
             "ranges": np.zeros((T, 2), np.uint32), "n_contrib": np.zeros((H, W), np.uint32),
             "final_T": np.zeros((H, W), np.float32)}
        f = lambda a, ty: a.ctypes.data_as(ty)
        _lib.check(L.dm4d_raster_read_geom(_p(self.geom), N, H, W, f(s["xy"], _lib.c_f), f(s["depths"], _lib.c_f),
                                           f(s["conic_opacity"], _lib.c_f), f(s["tiles_touched"], _lib.c_u32), st))
        _lib.check(L.dm4d_raster_read_sorted(_p(self.geom), _p(self.binning), N, H, W, min(self.D, self.cap),
                                             f(s["keys"], _lib.c_u64), f(s["values"], _lib.c_u32),
                                             f(s["ranges"], _lib.c_u32), st))
        _lib.check(L.dm4d_raster_read_image_state(_p(self.geom), _p(self.binning), _p(self.image), N, H, W, self.cap,
                                                  f(s["n_contrib"], _lib.c_u32), f(s["final_T"], _lib.c_f), st))
        for k in ("xy", "depths", "conic_opacity", "tiles_touched"):
            s[k] = s[k][:N]
        s["keys"], s["values"] = s["keys"][:self.D], s["values"][:self.D]
        return s

    def backward(self, gC, gD=None, gA=None, record_capacity=None):
        L, dev, N = self.L, self.dev, self.N
        t = lambda a: None if a is None else torch.tensor(np.asarray(a, np.float32), device=dev).contiguous()
        gC, gD, gA = t(gC), t(gD), t(gA)
        z = lambda *s: torch.full(s, float("nan"), device=dev)
        has_sr = self.t["s"] is not None
        M = 0 if self.t["sh"] is None else int(self.t["sh"].shape[1])
        o = {"dL_dmeans2D": z(N, 3), "dL_dmeans3D": z(N, 3), "dL_dopacity": z(N), "dL_dcolors": z(N, self.C),
             "dL_dsh": z(N, M, 3) if M else None, "dL_dscales": z(N, 3) if has_sr else None,
             "dL_drots": z(N, 4) if has_sr else None, "dL_dcov3D": z(N, 6)}
        rcap = self.R if record_capacity is None else record_capacity
        grad = torch.empty(L.dm4d_raster_grad_bytes(rcap, self.C), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(L.dm4d_rasterize_backward(self.settings, self.inputs, _p(self.radii), _p(self.geom),
                                             _p(self.binning), self.cap, _p(self.image), _p(grad), rcap, _p(gC), _p(gD),
                                             _p(gA), _p(o["dL_dmeans2D"]), _p(o["dL_dmeans3D"]), _p(o["dL_dopacity"]),
                                             _p(o["dL_dcolors"]), _p(o["dL_dsh"]), _p(o["dL_dscales"]),
                                             _p(o["dL_drots"]), _p(o["dL_dcov3D"]), st), "backward")
        torch.cuda.synchronize(dev)
        return {k: (None if v is None else v.cpu().numpy()) for k, v in o.items()}
