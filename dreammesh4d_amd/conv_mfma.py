"""3x3 / stride 1 / padding 1 convolution of NHWC float16 activations on the hand-written MFMA implicit-GEMM kernel
(csrc/conv_mfma.hip, C ABI ``dm4d_conv3x3_nhwc_f16``): the convolutions of the Zero123 UNet (forward only: the guidance model
runs under ``torch.no_grad``, extern/ldm_zero123/modules/diffusionmodules/openaimodel.py:214-275,429-842).

``conv3x3(x, w_ohwi, bias, residual=None)``: x [N,C,H,W] in torch.channels_last memory format (storage [N][H][W][C]),
w_ohwi = ``pack_weight(conv.weight)`` ([C_out][3][3][C_in] contiguous: what a channels_last weight tensor already is in
memory), optional bias [C_out] and residual (same shape / format as the output; the ResBlock's skip connection rides in the
epilogue).  `conv3x3` itself carries no autograd (the UNet runs under no_grad); `conv3x3_frozen` is the autograd form for
frozen parameters (the VAE encoder the rendered image is differentiated through, stable_zero123_guidance.py:153-160): its data
gradient is the same kernel on the flipped, transposed filter."""
import ctypes

import torch

from . import _lib

FLOPS = [0]        # multiply-add flops launched through this module (tools/zero123_profile.py: torch's flop counter cannot see them)

def supported(x, w):
    """Shapes / layouts the kernel takes (everything else stays on the library path)."""
    return (x.is_cuda and x.dtype == torch.float16 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
            and w.dim() == 4 and tuple(w.shape[2:]) == (3, 3) and w.dtype == torch.float16 and w.shape[1] == x.shape[1]
            and x.shape[1] % 32 == 0 and w.shape[0] % 32 == 0 and x.shape[0] > 0)


def pack_weight(w):
    """[C_out, C_in, 3, 3] -> [C_out, 3, 3, C_in] contiguous (a no-copy view for a channels_last weight)."""
    return w.detach().permute(0, 2, 3, 1).contiguous()


def conv3x3(x, w_ohwi, bias=None, residual=None, stride=1, pad=1):
    """stride 1 / pad 1 (the default), or stride 2 with pad 1 (the UNet's Downsample) / pad 0 (the VAE encoder's Downsample:
    F.pad(x, (0, 1, 0, 1)) + an unpadded convolution, without the padded copy): include/dm4d.h, dm4d_conv3x3_strided_nhwc_f16."""
    N, Ci, Hin, Win = x.shape
    Co = int(w_ohwi.shape[0])
    if not (x.is_cuda and x.dtype == torch.float16 and x.is_contiguous(memory_format=torch.channels_last) and Ci % 32 == 0 and Co % 32 == 0):
        raise ValueError("conv3x3: x must be a channels_last float16 HIP tensor with C_in and C_out multiples of 32 (see supported())")
    if tuple(w_ohwi.shape) != (Co, 3, 3, Ci) or not w_ohwi.is_contiguous():
        raise ValueError(f"conv3x3: weight must be [C_out,3,3,{Ci}] contiguous (pack_weight), got {tuple(w_ohwi.shape)}")
    if (stride, pad) not in ((1, 1), (2, 1), (2, 0)):
        raise ValueError("conv3x3: stride 1 / pad 1, or stride 2 / pad 0 or 1")
    H, W = (Hin, Win) if stride == 1 else (((Hin + 1) // 2, (Win + 1) // 2) if pad else (Hin // 2, Win // 2))
    if residual is not None and (tuple(residual.shape) != (N, Co, H, W) or not residual.is_contiguous(memory_format=torch.channels_last)
                                 or residual.dtype != torch.float16):
        raise ValueError("conv3x3: residual must be a channels_last float16 tensor of the output's shape")
    FLOPS[0] += 2 * N * H * W * Ci * Co * 9
    L = _lib.lib()
    y = torch.empty((N, Co, H, W), device=x.device, dtype=torch.float16, memory_format=torch.channels_last)
    # split-K partial sums of the small problems: from the caching allocator per call (stream-ordered, and a hipGraph capture
    # gets it from the graph's own pool)
    buf = torch.empty(L.dm4d_conv3x3_strided_scratch_bytes(N, Hin, Win, Ci, Co, stride), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.dm4d_conv3x3_strided_nhwc_f16(N, Hin, Win, Ci, Co, stride, pad, x.data_ptr(), w_ohwi.data_ptr(),
                                                   0 if bias is None else bias.data_ptr(), 0 if residual is None else residual.data_ptr(),
                                                   y.data_ptr(), buf.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream),
                   "dm4d_conv3x3_strided_nhwc_f16")
    return y


def pack_geglu(w, bias=None):
    """Rows of a GEGLU projection ([2 D, K]: D value rows, then D gate rows; attention.py:48-56) interleaved in blocks of 64 --
    rows 128 j .. 128 j + 63 = value rows 64 j .., rows 128 j + 64 .. = gate rows 64 j .. -- so that a 128-wide output tile of
    `linear(..., act="geglu")` holds the values AND the gates of 64 output columns (include/dm4d.h, dm4d_linear_f16)."""
    D = w.shape[0] // 2
    if w.shape[0] != 2 * D or D % 64:
        raise ValueError("pack_geglu: [2 D, K] with D a multiple of 64")
    idx = torch.arange(2 * D, device=w.device).view(2, D // 64, 64).transpose(0, 1).reshape(-1)
    return w.detach()[idx].contiguous(), (None if bias is None else bias.detach()[idx].contiguous())


def linear_supported(x, w):
    """float16 rows on a HIP device, K a multiple of 32, N of 8 (csrc/conv_mfma.hip: the one-tap implicit GEMM)."""
    return (x.is_cuda and x.dtype == torch.float16 and w.dtype == torch.float16 and w.dim() == 2 and x.shape[-1] == w.shape[1]
            and w.shape[1] % 32 == 0 and w.shape[0] % 8 == 0)


def linear(x, w, bias=None, residual=None, act=None):
    """act(x w^T + bias) (+ residual) on the MFMA kernel: x [..., K] float16 contiguous, w [N, K] contiguous (an nn.Linear weight as
    it lies), residual [..., N]; act None or "geglu" (w / bias packed by pack_geglu; result [..., N / 2]).  No autograd (frozen
    parameters under no_grad: zero123.py); include/dm4d.h, dm4d_linear_f16."""
    K, N = int(x.shape[-1]), int(w.shape[0])
    if not (linear_supported(x, w) and x.is_contiguous() and w.is_contiguous()):
        raise ValueError("linear: x [..., K] / w [N, K] contiguous float16 HIP tensors, K % 32 == 0, N % 8 == 0 (see linear_supported())")
    if act not in (None, "geglu"):
        raise ValueError("linear: act is None or 'geglu'")
    M = x.numel() // K
    No = N // 2 if act else N
    if residual is not None and (tuple(residual.shape) != tuple(x.shape[:-1]) + (No,) or not residual.is_contiguous() or residual.dtype != torch.float16):
        raise ValueError("linear: residual must be a contiguous float16 tensor of the output's shape")
    FLOPS[0] += 2 * M * K * N
    L = _lib.lib()
    y = torch.empty(tuple(x.shape[:-1]) + (No,), device=x.device, dtype=torch.float16)
    nbytes = L.dm4d_linear_scratch_bytes(M, K, N)
    buf = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes > 256 and not act else None
    with torch.cuda.device(x.device):
        _lib.check(L.dm4d_linear_f16(M, K, N, x.data_ptr(), w.data_ptr(), 0 if bias is None else bias.data_ptr(),
                                     0 if residual is None else residual.data_ptr(), y.data_ptr(), 1 if act else 0,
                                     0 if buf is None else buf.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream), "dm4d_linear_f16")
    return y


def narrow_supported(x, w):
    """A 3x3 convolution whose channel counts are NOT multiples of 32 (the ends of both networks: 8 -> 320 and 320 -> 4 in the UNet,
    3 -> 128 and 512 -> 8 in the VAE encoder): run on the same kernel with the channels zero-padded to the next multiple of 32."""
    return (x.is_cuda and x.dtype == torch.float16 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
            and w.dim() == 4 and tuple(w.shape[2:]) == (3, 3) and w.dtype == torch.float16 and w.shape[1] == x.shape[1]
            and (x.shape[1] % 32 != 0 or w.shape[0] % 32 != 0) and x.shape[0] > 0)


def pad_weight(w, bias=None):
    """(w [C_out', C_in', 3, 3] zero-padded to multiples of 32, bias [C_out'] or None): the padded taps / filters contribute exact
    zeros, so conv(pad_channels(x), pad_weight(w))[:, :C_out] is the convolution."""
    Co, Ci = int(w.shape[0]), int(w.shape[1])
    Cop, Cip = -(-Co // 32) * 32, -(-Ci // 32) * 32
    wp = torch.zeros((Cop, Cip, 3, 3), dtype=w.dtype, device=w.device)
    wp[:Co, :Ci] = w.detach()
    bp = None
    if bias is not None:
        bp = torch.zeros(Cop, dtype=bias.dtype, device=bias.device)
        bp[:Co] = bias.detach()
    return wp, bp


def pad_channels(x, C):
    """x [N, C', H, W] channels_last -> [N, C, H, W] channels_last with zeros in the new channels (differentiable: a slice assignment)."""
    if x.shape[1] == C:
        return x
    # (allocated channels-last and zeroed in place: `zeros(...).contiguous(channels_last)` was a fill AND a copy of the padded tensor --
    # 21 us for the VAE encoder's 4 x 256^2 x 32 input)
    y = torch.empty((x.shape[0], C, x.shape[2], x.shape[3]), dtype=x.dtype, device=x.device, memory_format=torch.channels_last).zero_()
    y[:, :x.shape[1]] = x
    return y


def pack_weight_transposed(w):
    """The filter of the DATA GRADIENT: dL/dx = conv3x3(dL/dy, w') with w'[ci][ky][kx][co] = w[co][2 - ky][2 - kx][ci]
    (stride 1, padding 1): [C_in, 3, 3, C_out] contiguous."""
    return w.detach().flip(2, 3).permute(1, 2, 3, 0).contiguous()


class _Conv3x3Frozen(torch.autograd.Function):
    """conv3x3 with FROZEN filter and bias (the guidance's VAE encoder: gradients flow to the image only): forward and data
    gradient both on the MFMA kernel; `residual` passes its gradient through."""

    @staticmethod
    def forward(ctx, x, w_ohwi, w_t, bias, residual):
        ctx.save_for_backward(w_t)
        ctx.has_res = residual is not None
        return conv3x3(x, w_ohwi, bias, residual)

    @staticmethod
    def backward(ctx, dy):
        (w_t,) = ctx.saved_tensors
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        dx = conv3x3(dy, w_t) if ctx.needs_input_grad[0] else None
        return dx, None, None, None, (dy if ctx.has_res else None)


def conv3x3_frozen(x, w_ohwi, w_t, bias=None, residual=None):
    return _Conv3x3Frozen.apply(x, w_ohwi, w_t, bias, residual)


class _ConvFirstFrozen(torch.autograd.Function):
    """The VAE encoder's first convolution (image, <= 4 channels -> 128 feature channels) with frozen parameters: forward on the
    library, data gradient on ``dm4d_conv3x3_c128_small_nhwc_f16`` (a 128 -> 3 channel convolution with the flipped, transposed
    filter: memory bound, where the library's grouped-convolution kernel took 0.5 ms of the 13 ms SDS step)."""

    @staticmethod
    def forward(ctx, x, w, b, w_t, w_pad=None, b_pad=None):
        ctx.save_for_backward(w_t)
        if w_pad is not None:        # on the MFMA kernel with the image's channels zero-padded to 32 (pad_weight: [128, 3, 3, 32])
            return conv3x3(pad_channels(x.detach(), int(w_pad.shape[3])), w_pad, b_pad)
        return torch.nn.functional.conv2d(x, w, b, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        (w_t,) = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None, None
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        N, C, H, W = dy.shape
        Ci = int(w_t.shape[0])
        FLOPS[0] += 2 * N * H * W * Ci * C * 9
        dx = torch.empty((N, Ci, H, W), device=dy.device, dtype=torch.float16, memory_format=torch.channels_last)
        with torch.cuda.device(dy.device):
            _lib.check(_lib.lib().dm4d_conv3x3_c128_small_nhwc_f16(N, H, W, Ci, dy.data_ptr(), w_t.data_ptr(), dx.data_ptr(),
                                                                   torch.cuda.current_stream(dy.device).cuda_stream), "dm4d_conv3x3_c128_small_nhwc_f16")
        return dx, None, None, None, None, None


def first_conv_supported(x, w):
    return (x.is_cuda and x.dtype == torch.float16 and w.dtype == torch.float16 and x.dim() == 4 and tuple(w.shape[2:]) == (3, 3)
            and w.shape[0] == 128 and 1 <= w.shape[1] <= 4 and x.shape[1] == w.shape[1])


def conv3x3_first_frozen(x, w, b, w_t, w_pad=None, b_pad=None):
    """x [N, C<=4, H, W] -> [N, 128, H, W]; w_t = pack_weight_transposed(w) ([C, 3, 3, 128]); w_pad / b_pad: pack_weight(pad_weight(w, b))
    for the forward on the MFMA kernel (None: the library's)."""
    return _ConvFirstFrozen.apply(x, w, b, w_t, w_pad, b_pad)


def pack_weight_s2_dgrad(w):
    """The four filters of the stride-2 / pad-0 data gradient (include/dm4d.h, dm4d_conv3x3_s2_dgrad_nhwc_f16): for the input
    pixels of parity (py, px) the taps (ky, kx) with ky in (2, 0) if py == 0 else (1,), kx likewise, as [C_in, KH, KW, C_out]."""
    out = []
    for py in (0, 1):
        for px in (0, 1):
            ky, kx = ([2, 0] if py == 0 else [1]), ([2, 0] if px == 0 else [1])
            out.append(w.detach()[:, :, ky][:, :, :, kx].permute(1, 2, 3, 0).contiguous())
    return out


def conv3x3_s2_dgrad(dy, w_cls, in_shape):
    """dL/dx [N, C_in, H, W] (channels_last) of the stride-2 / pad-0 Downsample convolution from dy [N, C_out, H/2, W/2]."""
    N, Ci, H, W = in_shape
    Co = int(dy.shape[1])
    if not (dy.is_cuda and dy.dtype == torch.float16 and dy.is_contiguous(memory_format=torch.channels_last) and H % 2 == 0 and W % 2 == 0
            and tuple(dy.shape) == (N, Co, H // 2, W // 2) and Ci % 32 == 0 and Co % 32 == 0):
        raise ValueError("conv3x3_s2_dgrad: channels_last float16 dy of an even-sized input, C_in and C_out multiples of 32")
    FLOPS[0] += 2 * N * (H // 2) * (W // 2) * Ci * Co * 9
    dx = torch.empty((N, Ci, H, W), device=dy.device, dtype=torch.float16, memory_format=torch.channels_last)
    ptrs = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in w_cls])
    with torch.cuda.device(dy.device):
        _lib.check(_lib.lib().dm4d_conv3x3_s2_dgrad_nhwc_f16(N, H, W, Ci, Co, dy.data_ptr(), ptrs, dx.data_ptr(),
                                                              torch.cuda.current_stream(dy.device).cuda_stream), "dm4d_conv3x3_s2_dgrad_nhwc_f16")
    return dx


class _Conv3x3Stride2Frozen(torch.autograd.Function):
    """Stride-2 convolution with frozen parameters (the VAE encoder's Downsample: pad 0 + one zero behind each axis): forward on the
    MFMA kernel without materialising the padded input; the data gradient as four stride-1 convolutions of dy, one per parity
    class of the input pixels, on the same kernel (pad 0, even sizes, channel counts in multiples of 32); anything else on the
    library's transposed convolution of the padded shape, cropped."""

    @staticmethod
    def forward(ctx, x, w, w_ohwi, bias, pad, w_cls=None):
        ctx.save_for_backward(w)
        ctx.in_shape, ctx.pad, ctx.w_cls = tuple(x.shape), pad, w_cls
        return conv3x3(x, w_ohwi, bias, None, stride=2, pad=pad)

    @staticmethod
    def backward(ctx, dy):
        (w,) = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None, None
        N, C, H, W = ctx.in_shape
        if (ctx.pad == 0 and ctx.w_cls is not None and H % 2 == 0 and W % 2 == 0 and C % 32 == 0 and dy.shape[1] % 32 == 0
                and dy.dtype == torch.float16):
            return conv3x3_s2_dgrad(dy.contiguous(memory_format=torch.channels_last), ctx.w_cls, ctx.in_shape), None, None, None, None, None
        if ctx.pad:
            dx = torch.nn.grad.conv2d_input((N, C, H, W), w, dy, stride=2, padding=1)
        else:
            dx = torch.nn.grad.conv2d_input((N, C, H + 1, W + 1), w, dy, stride=2, padding=0)[:, :, :H, :W]
        return dx, None, None, None, None, None


def conv3x3_stride2_frozen(x, w, w_ohwi, bias=None, pad=0, w_cls=None):
    """w_cls: pack_weight_s2_dgrad(w) for the MFMA data gradient (None: the library's)."""
    return _Conv3x3Stride2Frozen.apply(x, w, w_ohwi, bias, pad, w_cls)


def attention_supported(qkv):
    """qkv [B, L, 3, heads, D] float16 on a HIP device (the fused q / k / v projection), D in (40, 64, 80, 160), L % 64 == 0."""
    return (qkv.is_cuda and qkv.dtype == torch.float16 and qkv.dim() == 5 and qkv.shape[2] == 3 and qkv.is_contiguous()
            and int(qkv.shape[4]) in (40, 64, 80, 160) and qkv.shape[1] % 64 == 0)


def attention_qkv(qkv, scale=None):
    """softmax(q k^T * scale) v for qkv [B, L, 3, heads, D] (q = qkv[:, :, 0] ...) -> [B, L, heads * D]: the UNet's self-attention on the
    hand-written MFMA kernel (csrc/attention.hip, include/dm4d.h: dm4d_attention_f16).  No autograd."""
    if not attention_supported(qkv):
        raise ValueError("attention_qkv: contiguous float16 [B, L, 3, heads, D] on a HIP device, D in (40, 64, 80, 160), L % 64 == 0")
    B, L, _, H, D = (int(v) for v in qkv.shape)
    FLOPS[0] += 4 * B * H * L * L * D
    out = torch.empty(B, L, H * D, dtype=torch.float16, device=qkv.device)
    base, es = qkv.data_ptr(), 2
    with torch.cuda.device(qkv.device):
        _lib.check(_lib.lib().dm4d_attention_f16(B, L, H, D, base, base + H * D * es, base + 2 * H * D * es, L * 3 * H * D, 3 * H * D,
                                                 out.data_ptr(), float(D ** -0.5 if scale is None else scale),
                                                 torch.cuda.current_stream(qkv.device).cuda_stream), "dm4d_attention_f16")
    return out
