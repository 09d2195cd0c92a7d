"""The scalar arithmetic of an iteration's loss as ONE launch each way (csrc/imagehead.hip, include/dm4d.h: dm4d_weighted_sum*,
dm4d_partial_sums).

`loss = lambda_rgb * loss_rgb + lambda_mask * loss_mask + lambda_sds * loss_sds + ...` (C/system/sugar_4dgen.py:296-330,
sugar_static.py:246-340) on 0-dim device tensors is one torch operator per multiply and per add, and as many again in the
backward: ~25 launches of 5 us for a dozen flops.  `weighted_sum` evaluates the same float32 expression left to right (the same
roundings) in one launch and its backward (grad * lambda_i for every term) in another; `partial_sums` adds a head's per-workgroup
partial sums and applies its normalisation matrix.  CPU tensors (the host-logic tests) take the torch expression."""
import ctypes as C

import torch

from . import _lib


def partial_sums(partial, matrix):
    """partial [n, k] float32 on the HIP device, matrix [k][m] (host numbers, k, m <= 8) -> out [m] = (sum over n) @ matrix."""
    n, k = int(partial.shape[0]), int(partial.shape[1])
    m = len(matrix[0])
    if not (partial.is_cuda and partial.dtype == torch.float32 and partial.is_contiguous() and len(matrix) == k):
        raise ValueError("partial_sums: a contiguous float32 [n, k] HIP tensor and a [k][m] matrix")
    out = torch.empty(m, dtype=torch.float32, device=partial.device)
    flat = (C.c_float * (k * m))(*[float(v) for row in matrix for v in row])
    with torch.cuda.device(partial.device):
        _lib.check(_lib.lib().dm4d_partial_sums(n, k, m, partial.data_ptr(), flat, out.data_ptr(),
                                                torch.cuda.current_stream(partial.device).cuda_stream), "dm4d_partial_sums")
    return out


class _WeightedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, *terms):
        # weights[i]: a number for a 0-dim term, a tuple of numbers for a 1-D term (one per element)
        dev = terms[0].device
        ptrs, w, seg = [], [], []
        for wi, t in zip(weights, terms):
            t = t.detach()
            if t.dim() == 0:
                seg.append((len(w), None))
                ptrs.append(t.data_ptr())
                w.append(wi)
            else:
                seg.append((len(w), len(wi)))
                ptrs += [t.data_ptr() + 4 * j for j in range(len(wi))]
                w += list(wi)
        n = len(w)
        cp = (C.c_void_p * n)(*ptrs)
        ctx.w, ctx.n, ctx.seg = (C.c_float * n)(*w), n, seg
        out = torch.empty((), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().dm4d_weighted_sum(n, cp, ctx.w, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "dm4d_weighted_sum")
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.detach().to(torch.float32).contiguous()
        out = torch.empty(ctx.n, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().dm4d_weighted_sum_backward(ctx.n, g.data_ptr(), ctx.w, out.data_ptr(),
                                                             torch.cuda.current_stream(g.device).cuda_stream), "dm4d_weighted_sum_backward")
        return (None,) + tuple(out[a] if k is None else out[a:a + k] for a, k in ctx.seg)      # (views: no launches)


def _fusable(w, t):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32):
        return False
    if t.dim() == 0:
        return not isinstance(w, (tuple, list))
    return t.dim() == 1 and t.is_contiguous() and isinstance(w, (tuple, list)) and len(w) == t.shape[0]


def weighted_sum(pairs):
    """pairs: [(lambda_i, term_i)] -> sum_i lambda_i * term_i, accumulated left to right from 0.0 in float32.  term_i: a 0-dim
    tensor, or a 1-D tensor with a tuple of weights (one per element: a head's vector of terms, without unbinding it)."""
    pairs = [((tuple(float(v) for v in w) if isinstance(w, (tuple, list)) else float(w)), t) for w, t in pairs]
    n = sum(len(w) if isinstance(w, tuple) else 1 for w, _ in pairs)
    if pairs and n <= 16 and all(_fusable(w, t) for w, t in pairs):
        return _WeightedSum.apply(tuple(w for w, _ in pairs), *[t for _, t in pairs])
    loss = 0.0
    for w, t in pairs:
        if isinstance(w, tuple):
            for wj, tj in zip(w, t.unbind(0)):
                loss = loss + wj * tj
        else:
            loss = loss + w * t
    return loss
