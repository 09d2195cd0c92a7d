"""Multi-GPU layer of the hot path: one process per GPU, `torch.distributed`
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference relies on implicit Lightning DDP (launch.py:114,228-235: every rank seeds differently,
samples its own 4 frames x views, parameters replicated) and would in fact fail on its
trainable-but-unused parameters.  Here the two pieces are explicit:

* `shard_frames`   -- (frame, view) units are independent: rank r takes frames {4r .. 4r+3} of the
                      timeline each iteration; no data-path collective.
* `GradAllReducer` -- the ONE exchange step: a single all-reduce (sum, x 1/world) of one flat
                      float32 buffer holding every trainable gradient; parameters that received no
                      gradient on a rank contribute zeros (DDP would raise on them).  At the shipped
                      dynamic-stage configuration the buffer is 143 MB (35.76 M floats), i.e. one large
                      message -- what a point-to-point xGMI ring wants -- instead of DDP's 25 MB buckets.
"""
from typing import Iterable, List

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_frames(n_frames: int, rank_: int, world_: int, frames_per_rank: int = 4, iteration: int = 0) -> List[int]:
    """Frames rendered by `rank_` in `iteration`: consecutive blocks of `frames_per_rank`, rotated by
    the iteration so every rank visits the whole timeline (cfg 4: 32 frames / 8 ranks = 4 each)."""
    start = (rank_ + iteration * world_) * frames_per_rank
    return [(start + i) % n_frames for i in range(frames_per_rank)]


class GradAllReducer:
    """One all-reduce per step over one flat float32 buffer.

    `touched` (optional): {parameter: LongTensor of element indices in STORAGE order (`storage_flat`)}.  For those parameters only the listed
    elements are exchanged; every rank must pass the SAME index sets and the gradient must be zero elsewhere on
    every rank.  That is the case for the HexPlane grids: the graph nodes are static and identical on all ranks, so
    the spatial planes only ever receive gradient at the texels the nodes touch (`touched_from_plan`) -- 1.0 M of
    their 33.4 M elements at the shipped configuration -- which shrinks the message from 143 MB to ~13 MB (spatial
    texels + the dense time planes + the MLP): the exchange stops being the per-link-bound term of a step
    (one xGMI ring moves 143 MB in ~1 ms; a step is ~2 ms).  The result is identical to the
    dense all-reduce (mean of zeros is zero)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], touched=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, torch.float32
        touched = touched or {}
        self.index = [touched.get(p) for p in self.params]
        self.offsets, n = [], 0
        for p, ix in zip(self.params, self.index):
            self.offsets.append(n)
            n += p.numel() if ix is None else int(ix.numel())
            if ix is not None and (ix.dtype != torch.long or ix.device != p.device):
                raise ValueError("touched indices must be int64 tensors on the parameter's device")
        self.flat = torch.zeros(n, dtype=dt, device=dev)
        self.dense_elements = sum(p.numel() for p in self.params)

    @property
    def nbytes(self):
        return self.flat.numel() * 4

    def __call__(self):
        """all-reduce (mean over ranks) of every gradient, in place.  No-op for a single process."""
        w = world()
        if w == 1:
            return
        self.pack()
        if self.flat.is_cuda and dist.get_backend() == "gloo":
            host = self.flat.cpu()      # functional rehearsals over gloo: stage through the host
            dist.all_reduce(host)
            self.flat.copy_(host)
        else:
            dist.all_reduce(self.flat)  # one message; RCCL picks ring / direct on the xGMI mesh
        self.unpack(1.0 / w)

    # On a HIP device packing and unpacking are ONE launch each (csrc/gradpack.hip, C ABI dm4d_grad_pack / _unpack)
    # instead of ~80 copy / index kernels; on CPU tensors (the gloo tests) the same thing with torch ops.
    def _segments(self, for_unpack):
        import ctypes as C

        from . import _lib

        if len(self.params) > _lib.MAX_GRAD_SEGMENTS:
            raise ValueError(f"more than {_lib.MAX_GRAD_SEGMENTS} gradient tensors")
        seg = _lib.GradSegments()
        seg.n_segments = len(self.params)
        for k, (p, o, ix) in enumerate(zip(self.params, self.offsets, self.index)):
            if p.grad is None and for_unpack:
                p.grad = torch.zeros_like(p, memory_format=torch.preserve_format)
            g = p.grad
            if g is not None and (g.dtype != torch.float32 or g.stride() != p.stride()):
                raise ValueError("gradients must be float32 with the parameter's strides")
            storage_flat(p)                                     # dense (contiguous or channels_last), else raises
            seg.grad[k] = None if g is None else g.data_ptr()
            seg.index[k] = None if ix is None else ix.data_ptr()
            seg.count[k] = p.numel() if ix is None else ix.numel()
            seg.offset[k] = o
        return seg

    def pack(self):
        flat = self.flat
        if flat.is_cuda:
            import ctypes as C

            from . import _lib

            seg = self._segments(False)
            with torch.cuda.device(flat.device):
                _lib.check(_lib.lib().dm4d_grad_pack(C.byref(seg), flat.data_ptr(),
                                                     torch.cuda.current_stream(flat.device).cuda_stream), "dm4d_grad_pack")
            return
        for p, o, ix in zip(self.params, self.offsets, self.index):
            n = p.numel() if ix is None else ix.numel()
            seg = flat[o:o + n]
            if p.grad is None:
                seg.zero_()
            elif ix is None:
                seg.copy_(storage_flat(p.grad))
            else:
                torch.index_select(storage_flat(p.grad), 0, ix, out=seg)

    def unpack(self, scale):
        flat = self.flat
        if flat.is_cuda:
            import ctypes as C

            from . import _lib

            seg = self._segments(True)
            with torch.cuda.device(flat.device):
                _lib.check(_lib.lib().dm4d_grad_unpack(C.byref(seg), flat.data_ptr(), float(scale),
                                                       torch.cuda.current_stream(flat.device).cuda_stream), "dm4d_grad_unpack")
            return
        flat.mul_(scale)
        for p, o, ix in zip(self.params, self.offsets, self.index):
            n = p.numel() if ix is None else ix.numel()
            seg = flat[o:o + n]
            if p.grad is None:
                p.grad = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif p.grad.stride() != p.stride():
                raise ValueError("gradients must have the parameter's strides")
            if ix is None:
                storage_flat(p.grad).copy_(seg)
            else:
                storage_flat(p.grad).index_copy_(0, ix, seg)


def storage_flat(t):
    """1-D view of a dense tensor in STORAGE order (contiguous and channels_last tensors alike)."""
    if t.is_contiguous():
        return t.view(-1)
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return t.as_strided((t.numel(),), (1,), t.storage_offset())
    raise ValueError("parameters / gradients must be dense (contiguous or channels_last)")


def touched_from_plan(field, plan):
    """{spatial plane parameter: STORAGE indices of the elements that can receive gradient} from a `hexplane.HexPlan`
    (its `sp` lists are the touched texels per (scale, plane); element = channel * H * W + texel for a contiguous
    plane, texel * C + channel for a channels_last one).  The time planes are exchanged densely (which rows a step
    touches depends on the rank's timestamps)."""
    sc, pl, tx = (plan.sp[k].to(torch.long) for k in ("scale", "plane", "texel"))
    out = {}
    for s, planes in enumerate(field.grids):
        for p, par in enumerate(planes):
            m = (sc == s) & (pl == p)
            if not bool(m.any()):
                continue
            C, HW = int(par.shape[1]), int(par.shape[2]) * int(par.shape[3])
            t = tx[m]
            ch = torch.arange(C, device=t.device, dtype=torch.long)
            if par.is_contiguous():
                out[par] = (ch[:, None] * HW + t[None, :]).reshape(-1)
            else:
                storage_flat(par)                               # raises unless channels_last
                out[par] = (t[:, None] * C + ch[None, :]).reshape(-1)
    return out


def all_reduce_max(t):
    """In-place MAX over the ranks of a small tensor (the overflow flag of a step: if ANY rank's forward overflowed, every
    rank skips the optimiser step, so the replicas stay identical).  No-op for a single process."""
    if world() == 1:
        return t
    if t.is_cuda and dist.get_backend() == "gloo":
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MAX)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t
