"""``simple_knn._C`` mirror: ``distCUDA2(points[N,3] float32 on the GPU) -> Tensor[N]``
(call site custom/threestudio-dreammesh4d/geometry/gaussian_base.py:435-438), computed by
libdm4d_hip.so (csrc/knn.hip).  No CPU path."""
import torch

from .. import _lib

BRUTE_FORCE_MAX = 8192


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if points.device.type != "cuda":
        raise RuntimeError("simple_knn.distCUDA2 (dm4d): points must live on a HIP device; there is no CPU path")
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError("points must be [N, 3]")
    L = _lib.lib()
    p = points.detach().to(torch.float32).contiguous()
    out = torch.empty(p.shape[0], dtype=torch.float32, device=p.device)
    n = int(p.shape[0])
    with torch.cuda.device(p.device):
        st = torch.cuda.current_stream(p.device).cuda_stream
        if n <= BRUTE_FORCE_MAX:       # exhaustive LDS-tiled search: no scratch, fastest for small clouds
            _lib.check(L.dm4d_dist2_knn3(n, p.data_ptr() if n else None, out.data_ptr() if n else None, st), "dm4d_dist2_knn3")
        else:                          # Morton-ordered 1024-point boxes (upstream's structure): same values, O(N) boxes visited
            nbytes = L.dm4d_knn_scratch_bytes(n)
            scratch = torch.empty(nbytes, dtype=torch.uint8, device=p.device)
            _lib.check(L.dm4d_dist2_knn3_ws(n, p.data_ptr(), out.data_ptr(), scratch.data_ptr(), nbytes, st), "dm4d_dist2_knn3_ws")
    return out
