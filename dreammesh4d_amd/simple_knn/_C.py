"""``simple_knn._C`` mirror: ``distCUDA2(points[N,3] float32 on the GPU) -> Tensor[N]``
(call site custom/threestudio-dreammesh4d/geometry/gaussian_base.py:435-438), computed by
libdm4d_hip.so (csrc/knn.hip).  No CPU path."""
import torch

from .. import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if points.device.type != "cuda":
        raise RuntimeError("simple_knn.distCUDA2 (dm4d): points must live on a HIP device; there is no CPU path")
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError("points must be [N, 3]")
    L = _lib.lib()
    p = points.detach().to(torch.float32).contiguous()
    out = torch.empty(p.shape[0], dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        _lib.check(L.dm4d_dist2_knn3(p.shape[0], p.data_ptr() if p.numel() else None, out.data_ptr() if p.numel() else None,
                                     torch.cuda.current_stream(p.device).cuda_stream), "dm4d_dist2_knn3")
    return out
