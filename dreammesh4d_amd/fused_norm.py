"""GroupNorm (+ SiLU) and two fused elementwise operators of the Zero123 SDS step for channels-last activations: the
torch-side binding of csrc/groupnorm.hip (``dm4d_groupnorm_nhwc_forward`` / ``_backward``) and csrc/pointwise.hip
(``dm4d_add_bias_nhwc``, ``dm4d_geglu``), include/dm4d.h.

``group_norm(module, x, silu=False, add=None)`` evaluates ``act(module(x + add[:, :, None, None]))`` for a
``torch.nn.GroupNorm`` -- the "GroupNorm32, SiLU" pairs of the reference's ResBlocks
(extern/ldm_zero123/modules/diffusionmodules/openaimodel.py:214-275), the VAE encoder's "Normalize, nonlinearity"
(diffusionmodules/model.py) and the plain norms in front of the attention blocks.  On a HIP device with a dense
channels-last input the HIP kernels run (two launches, the activation read twice and written once); anything else --
CPU tensors of the golden-vector tests, NCHW tensors -- takes torch's own operators, literally as the reference writes
them.  gamma / beta are treated as frozen: the guidance model is not trained (guidance/...zero123_guidance.py:90-91),
so backward returns dL/dx only.
"""
import torch
import torch.nn.functional as F

from . import _lib

_DTYPES = {torch.float16: 0, torch.float32: 1}      # DM4D_GN_F16, DM4D_GN_F32
MAX_SPLITS = 128                                    # DM4D_GN_MAX_SPLITS

# A tensor on a HIP device that does NOT take the HIP operator (a layout regression upstream: NCHW activations, an odd
# channel count, a dtype mismatch) silently costs the step its fused kernels.  Every such call is counted here, by operator
# and reason; `expect_fused()` turns them into an error (the guidance step runs under it, tests assert the count is 0).
FALLBACKS = {}


def _fallback(op, x, reason):
    if x.is_cuda:
        key = (op, reason)
        FALLBACKS[key] = FALLBACKS.get(key, 0) + 1


def fallback_count():
    return sum(FALLBACKS.values())


class expect_fused:
    """Context: device tensors inside are expected to take the HIP operators.  Leaving it with new fallbacks warns (once per
    process, with the operators and reasons); under DM4D_STRICT_FUSED=1 -- the GPU tests -- it raises."""
    _warned = False

    def __enter__(self):
        self.before = dict(FALLBACKS)
        return self

    def __exit__(self, et, ev, tb):
        if et is not None:
            return False
        new = {k: v - self.before.get(k, 0) for k, v in FALLBACKS.items() if v != self.before.get(k, 0)}
        if new:
            import os
            import warnings

            msg = (f"fused_norm: {sum(new.values())} call(s) on device tensors fell back to torch operators: {new} (the HIP operators "
                   "need dense channels_last activations with C % 8 == 0 (fp16) / C % 4 == 0 (fp32) and frozen affine parameters)")
            if os.environ.get("DM4D_STRICT_FUSED", "0") == "1":
                raise RuntimeError(msg)
            if not expect_fused._warned:
                warnings.warn(msg)
                expect_fused._warned = True
        return False


def _splits(hw):
    """Workgroups per sample: slabs of >= 16 positions, at most DM4D_GN_MAX_SPLITS."""
    return max(1, min(MAX_SPLITS, hw // 16))


def is_channels_last(x):
    return x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)


class _GroupNormNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, add, groups, eps, silu, skip=False):
        N, C, H, W = x.shape
        L = _lib.lib()
        y = torch.empty_like(x)                                    # same (channels-last) strides
        S = _splits(H * W)
        stats = torch.empty(N, groups, 2, device=x.device, dtype=torch.float32)
        scratch = torch.empty(N, S, groups, 2, device=x.device, dtype=torch.float32)
        add_stride = 0 if add is None or add.dim() == 1 else int(add.stride(0))     # [C]: the same for every sample; [N, C]: rows may be strided
        with torch.cuda.device(x.device):
            _lib.check(L.dm4d_groupnorm_nhwc_forward(N, H * W, C, groups, _DTYPES[x.dtype], x.data_ptr(),
                                                 0 if add is None else add.data_ptr(), add_stride, weight.data_ptr(), bias.data_ptr(),
                                                 eps, int(silu), y.data_ptr(), stats.data_ptr(), scratch.data_ptr(), S,
                                                 torch.cuda.current_stream(x.device).cuda_stream), "groupnorm forward")
        ctx.save_for_backward(x, weight, bias, stats, add)
        ctx.cfg = (groups, bool(silu), S, add_stride)
        if skip:
            # second output: x itself (a view), for the branch that goes AROUND the norm (a ResnetBlock's residual connection): the
            # gradient it brings back is added where dx is written, not by autograd's accumulation launch
            ctx.set_materialize_grads(False)
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, d_skip=None):
        x, weight, bias, stats, add = ctx.saved_tensors
        groups, silu, S, add_stride = ctx.cfg
        N, C, H, W = x.shape
        if dy is None:                                             # only the branch around the norm was used
            return d_skip, None, None, None, None, None, None, None
        if not is_channels_last(dy):
            dy = dy.contiguous(memory_format=torch.channels_last)
        if d_skip is not None and not (is_channels_last(d_skip) and d_skip.dtype == x.dtype):
            d_skip = d_skip.to(x.dtype).contiguous(memory_format=torch.channels_last)
        L = _lib.lib()
        dx = torch.empty_like(x)
        scratch = torch.empty(N, S, groups, 2, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(L.dm4d_groupnorm_nhwc_backward_add(N, H * W, C, groups, _DTYPES[x.dtype], x.data_ptr(),
                                                      0 if add is None else add.data_ptr(), add_stride, weight.data_ptr(), bias.data_ptr(),
                                                      stats.data_ptr(), int(silu), dy.data_ptr(), 0 if d_skip is None else d_skip.data_ptr(),
                                                      dx.data_ptr(), scratch.data_ptr(), S,
                                                      torch.cuda.current_stream(x.device).cuda_stream), "groupnorm backward")
        return dx, None, None, None, None, None, None, None


def fused_ok(module, x):
    w = module.weight
    return (x.is_cuda and is_channels_last(x) and x.dtype in _DTYPES and w is not None and module.bias is not None
            and w.dtype == x.dtype and x.shape[1] % (8 if x.dtype == torch.float16 else 4) == 0 and x.shape[0] > 0
            and not (torch.is_grad_enabled() and (w.requires_grad or module.bias.requires_grad)))


def group_norm(module, x, silu=False, add=None, float32=False, skip=False):
    """act(GroupNorm(x + add)), add [N, C] (per sample and channel) or [C] (per channel).  `float32`: the reference's
    GroupNorm32 (statistics and affine map evaluated in float32 around half-precision storage) -- what the HIP kernels do
    for every input.  `skip`: returns (y, x') with x' = x for the caller's branch AROUND the norm (a residual connection): on the
    HIP operator the gradient x' brings back is added inside the backward kernel; elsewhere x' is x."""
    if add is not None:
        C = x.shape[1]
        if add.dim() == 2 and add.shape[0] == 1 and x.shape[0] != 1:
            add = add.reshape(-1)                                        # [1, C]: one row for every sample
        if tuple(add.shape) not in ((C,), (x.shape[0], C)):
            raise ValueError(f"group_norm: add must be [{C}] or [{x.shape[0]}, {C}], got {tuple(add.shape)}")
    if fused_ok(module, x):
        if add is not None:
            if torch.is_grad_enabled() and add.requires_grad:        # the operator treats `add` as a constant
                x, add = x + (add if add.dim() == 1 else add[:, :, None, None]).type(x.dtype).view(-1, x.shape[1], 1, 1), None
            else:
                add = add.detach().to(x.dtype)
                # rows of a wider matrix are taken as they are (the UNet projects the timestep embedding for all its ResBlocks
                # in one GEMM and hands each block a column slice): the C ABI takes the sample stride
                if not (add.dim() == 2 and add.stride(1) == 1 and add.stride(0) >= add.shape[1] and add.stride(0) % 8 == 0
                        and add.data_ptr() % 16 == 0) and not add.is_contiguous():
                    add = add.contiguous()
        if skip and torch.is_grad_enabled() and x.requires_grad:
            return _GroupNormNHWC.apply(x, module.weight, module.bias, add, module.num_groups, module.eps, silu, True)
        y = _GroupNormNHWC.apply(x, module.weight, module.bias, add, module.num_groups, module.eps, silu)
        return (y, x) if skip else y
    _fallback("group_norm", x, "layout" if not is_channels_last(x) else "dtype/channels/trainable affine")
    x0 = x
    if add is not None:
        x = x + add.type(x.dtype).view(-1, x.shape[1], 1, 1)
    if float32 and not (x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and module.weight.dtype == x.dtype):
        y = F.group_norm(x.float(), module.num_groups, module.weight.float(), module.bias.float(), module.eps).type(x.dtype)
    else:
        # (half tensors on a device: the library kernel already accumulates in float32 and rounds once)
        y = F.group_norm(x, module.num_groups, module.weight, module.bias, module.eps)
    y = F.silu(y) if silu else y
    return (y, x0) if skip else y


# ----------------------------------------------------------------------------- a + b + bias[c], GEGLU
class _AddBias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, bias):
        N, C, H, W = a.shape
        y = torch.empty_like(a)
        with torch.cuda.device(a.device):
            _lib.check(_lib.lib().dm4d_add_bias_nhwc(N * H * W, C, _DTYPES[a.dtype], a.data_ptr(), b.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                                 torch.cuda.current_stream(a.device).cuda_stream), "add_bias")
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy, None            # (the bias is a frozen parameter)


def add_bias(a, b, bias):
    """a + b + bias[None, :, None, None]: the end of a ResBlock (skip + convolution output + the convolution's bias)."""
    if (a.is_cuda and is_channels_last(a) and is_channels_last(b) and a.dtype in _DTYPES and b.dtype == a.dtype and bias.dtype == a.dtype
            and a.shape == b.shape and a.shape[1] % (8 if a.dtype == torch.float16 else 4) == 0 and a.numel() > 0
            and not (torch.is_grad_enabled() and bias.requires_grad)):
        return _AddBias.apply(a, b, bias)
    _fallback("add_bias", a, "layout" if not (is_channels_last(a) and is_channels_last(b)) else "dtype/channels/trainable bias")
    return a + (b + bias.view(1, -1, 1, 1))


def geglu(proj):
    """x * gelu(gate) for proj = [x | gate] on the last axis (extern/ldm_zero123/modules/attention.py:48-56)."""
    D = proj.shape[-1] // 2
    if (proj.is_cuda and proj.is_contiguous() and proj.dtype in _DTYPES and D % (8 if proj.dtype == torch.float16 else 4) == 0
            and proj.numel() > 0 and not (torch.is_grad_enabled() and proj.requires_grad)):
        y = torch.empty(proj.shape[:-1] + (D,), device=proj.device, dtype=proj.dtype)
        with torch.cuda.device(proj.device):
            _lib.check(_lib.lib().dm4d_geglu(proj.numel() // (2 * D), D, _DTYPES[proj.dtype], proj.data_ptr(), y.data_ptr(),
                                         torch.cuda.current_stream(proj.device).cuda_stream), "geglu")
        return y
    _fallback("geglu", proj, "layout" if not proj.is_contiguous() else "dtype/width/requires_grad")
    x, gate = proj.chunk(2, dim=-1)
    return x * F.gelu(gate)


def add_layer_norm(norm, x, tok=None, bias2=None, want_sum=True):
    """(LayerNorm(s), s + bias2) for s = x + tok, x [B, L, C] float16 contiguous on a HIP device, tok [B, 1, C] or None (a row
    per sample), ``norm`` an nn.LayerNorm over C with affine parameters, bias2 [C] or None -- ONE launch
    (``dm4d_add_layernorm_f16``) for what the transformer block writes as up to two adds and a LayerNorm.  No autograd: the
    caller (zero123.BasicTransformerBlock) uses it under no_grad with frozen parameters only."""
    B, Lq, C = x.shape
    if not (x.is_cuda and x.dtype == torch.float16 and x.is_contiguous() and C % 8 == 0 and C <= 2048
            and norm.weight.dtype == torch.float16 and tuple(norm.normalized_shape) == (C,)):
        raise ValueError("add_layer_norm: needs a contiguous float16 [B, L, C] HIP tensor, C % 8 == 0, C <= 2048")
    if tok is not None:
        tok = tok.reshape(B, C)
        if not tok.is_contiguous() or tok.dtype != torch.float16:
            tok = tok.to(torch.float16).contiguous()
    n = torch.empty_like(x)
    xb = torch.empty_like(x) if want_sum else None
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().dm4d_add_layernorm_f16(B * Lq, C, Lq, x.data_ptr(), 0 if tok is None else tok.data_ptr(), norm.weight.data_ptr(),
                                                     norm.bias.data_ptr(), float(norm.eps), 0 if bias2 is None else bias2.data_ptr(), n.data_ptr(),
                                                     0 if xb is None else xb.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream),
                   "add_layernorm")
    return n, xb
