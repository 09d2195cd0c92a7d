"""Zero123 SDS gradient step of the dynamic stage.

Host-side mirror of
  custom/threestudio-dreammesh4d/guidance/temporal_stable_zero123_guidance.py:228-236 (encode_images),
  :250-297 (get_cond), :299-374 (__call__: the SDS step), :376-386 (update_step)
  threestudio/models/guidance/stable_zero123_guidance.py:277-... (static twin, no frame_indices)
and of the pieces of the vendored LDM they call:
  extern/ldm_zero123/modules/diffusionmodules/openaimodel.py:429-842 (UNetModel, in 8 / out 4 channels,
      model_channels 320, mult (1,2,4,4), attention at ds 1,2,4, 8 heads, context 768),
  extern/ldm_zero123/modules/attention.py:150-190 (CrossAttention), :193-... (BasicTransformerBlock, SpatialTransformer),
  extern/ldm_zero123/modules/diffusionmodules/model.py (AutoencoderKL Encoder, ch 128, mult (1,2,4,4)),
  extern/ldm_zero123/models/diffusion/ddpm.py:653 (cc_projection), :1953-1956 (hybrid conditioning).

This is where MFMA belongs on this path: dense conv / GEMM through PyTorch-ROCm (MIOpen / hipBLASLt), fp16
weights, attention through F.scaled_dot_product_attention instead of einsum + softmax.  On a HIP device the
activations are kept channels-last (NHWC) end to end -- MIOpen's implicit-GEMM convolutions are NHWC kernels and wrap
NCHW tensors in transposes, 11 % of the step in profiles/r02_zero123.md -- the 1x1 convolutions around the attention
blocks become plain GEMMs on the token view of the same memory, and every GroupNorm (+ SiLU, + the ResBlock's
timestep-embedding add) is ONE hand-written HIP operator (fused_norm.py, csrc/groupnorm.hip).

Module and parameter names follow the LDM checkpoint layout (`model.diffusion_model.*`,
`first_stage_model.encoder.*`, `first_stage_model.quant_conv.*`, `cc_projection.*`) so
`stable_zero123.ckpt` loads with `load_zero123_state_dict` (weights are not in the reference tree,
load/zero123/download.sh).  Architecture parity is pinned by tests/golden/zero123_small.npz: a
reduced-width UNet and encoder of the REFERENCE code, filled by a name-seeded recipe both sides share.
"""
import contextlib
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import conv_mfma
from .fused_norm import add_bias, add_layer_norm, geglu, group_norm, is_channels_last


# cross-attention over a one-token context evaluated in closed form (CrossAttention.single_token); False: the general path
SINGLE_TOKEN_SHORTCUT = True


# ----------------------------------------------------------------------------- building blocks
class GroupNorm32(nn.GroupNorm):
    """GroupNorm evaluated in fp32 (extern/ldm_zero123/modules/diffusionmodules/util.py:242-244).  Half tensors on a device:
    both the HIP operator and the library kernel accumulate the statistics and evaluate the affine map in float32 and round
    once on output -- within 1 half-precision ulp of casting to float32 around it (measured), without the cast launches.
    float32 modules (the golden-vector tests) take the reference's literal path."""

    def forward(self, x):
        return group_norm(self, x, float32=True)


def _to_nhwc(x):
    return x if is_channels_last(x) else x.contiguous(memory_format=torch.channels_last)


def _fold_bias(conv, x):
    """On a HIP device with NHWC activations a 3x3 convolution's bias is folded into the operator that follows it (the
    GroupNorm's `add`, the residual add): MIOpen's NHWC convolutions add it in a separate kernel otherwise."""
    return x.is_cuda and is_channels_last(x) and conv.bias is not None and not (torch.is_grad_enabled() and conv.bias.requires_grad)


def _conv_nobias(conv, x):
    return F.conv2d(x, conv.weight, None, conv.stride, conv.padding)


VAE_ATTENTION_BMM = os.environ.get("DM4D_VAE_ATTN_BMM", "1") != "0"                  # (A/B switch: _VaeAttn as bmm + softmax + bmm)
# The transformer blocks' linear layers on the hand-written MFMA kernel (conv_mfma.linear: the one-tap implicit GEMM with bias /
# residual / GEGLU in its epilogue) where it beats the library GEMM + the extra launches: measured per shape (tools/linear_shapes.py),
# that is the 32 x 32 level (M = 8192 rows: q/k/v 18.7 -> 14 us, attention output + residual 13.7 -> 10.3, GEGLU projection 52.5 ->
# 39.5, proj_out + residual 13.7 -> ~9); at M <= 2048 hipBLASLt's kernels are as fast or faster and stay.
MFMA_LINEAR_MIN_ROWS = int(os.environ.get("DM4D_MFMA_LINEAR_MIN_ROWS", "4096"))      # (A/B switch: a huge value = library everywhere)
# ... and the projections with a RESIDUAL (attention output, proj_out) from 512 rows on: the fused epilogue replaces the library GEMM + an
# add launch (or its prepared C operand): 2048 x 640 -> 640: 7.9 us against 12.0, 512 x 1280 -> 1280: 10.5 against 11.4 + the add
MFMA_LINEAR_RES_MIN_ROWS = int(os.environ.get("DM4D_MFMA_LINEAR_RES_MIN_ROWS", "512"))
# ... the feed-forward's second projection (+ the residual the LayerNorm kernel prepared) and the 1x1 convolutions (proj_in, a ResBlock's
# skip connection) at the 32 x 32 level too: in the step (tools/sds_ab.py, two alternating rounds) 9.94 / 9.95 ms with the library, 9.95 /
# 9.97 and 9.92 / 9.94 with these two on the MFMA kernel from 8192 rows on (13 library launches fewer, the same loss to the last
# digit), but 10.02-10.03 and 9.98-10.00 from 512 rows on (like q/k/v + GEGLU from 512 rows on: 10.02-10.03): at M <= 2048 hipBLASLt's
# small tiles stay the faster choice
MFMA_FF2_MIN_ROWS = int(os.environ.get("DM4D_MFMA_FF2_MIN_ROWS", "4096"))                 # (A/B switches)
MFMA_CONV1X1_MIN_ROWS = int(os.environ.get("DM4D_MFMA_CONV1X1_MIN_ROWS", "4096"))
MFMA_ATTENTION = os.environ.get("DM4D_MFMA_ATTENTION", "1") != "0"            # (A/B switch: the self-attention on csrc/attention.hip)
FUSE_QKV = os.environ.get("DM4D_FUSE_QKV", "1") != "0"                              # (A/B switch: CrossAttention's one-GEMM q, k, v)
FUSE_ADD_LAYERNORM = os.environ.get("DM4D_FUSE_ADD_LN", "1") != "0"                 # (A/B switch for BasicTransformerBlock._fused_no_grad)
BATCH_SMALL_GEMMS = os.environ.get("DM4D_BATCH_SMALL_GEMMS", "1") != "0"      # (A/B switch for UNetModel._batched_small_gemms)
_USE_MFMA_CONV_NARROW = os.environ.get("DM4D_MFMA_CONV_NARROW", "1") != "0"  # (A/B switch: the 3- / 4- / 8-channel convolutions on zero-padded channels)
_USE_MFMA_S2_DGRAD = os.environ.get("DM4D_MFMA_S2_DGRAD", "1") != "0"       # (A/B switch: the stride-2 data gradients of the VAE encoder)
_USE_MFMA_CONV_S2 = os.environ.get("DM4D_MFMA_CONV_S2", "1") != "0"            # (A/B switch: the stride-2 Downsample convolutions)
_USE_MFMA_CONV = os.environ.get("DM4D_MFMA_CONV", "1") != "0"      # (A/B switch: "0" keeps every convolution on the library)


def _library_fallback(op, x, reason):
    """Count a call that took the library path although it ran on a device tensor (fused_norm.FALLBACKS: the same counter the
    GroupNorm / add / GEGLU operators use; `fused_norm.expect_fused()` reports new entries when the guidance step ends and
    raises under DM4D_STRICT_FUSED=1)."""
    from . import fused_norm

    if x.dtype == torch.float16:       # (a float32 model is a configuration, not a regression: the hand-written kernels are the fp16 path)
        fused_norm._fallback(op, x, reason)


def _why_not_mfma_conv(conv, x, frozen):
    w = conv.weight
    if not _USE_MFMA_CONV:
        return "DM4D_MFMA_CONV=0"
    if not frozen:
        return "trainable parameters"
    if x.dtype != torch.float16 or w.dtype != torch.float16:
        return f"dtype {x.dtype}"
    if not is_channels_last(x):
        return "not channels_last"
    return f"shape C_in {x.shape[1]} C_out {w.shape[0]} groups {conv.groups}"


def _conv3x3(conv, x, bias=True, residual=None):
    """A 3x3 / stride 1 / padding 1 convolution (+ bias, + residual) with FROZEN parameters (the guidance model is not
    trained).  On a HIP device with channels-last float16 activations: the hand-written MFMA implicit-GEMM kernel
    (csrc/conv_mfma.hip), bias and residual in its epilogue -- without autograd in the UNet, with the data gradient on the same
    kernel (transposed, flipped filter) in the VAE encoder.  Anything else: the library convolution, literally as the
    reference writes it."""
    from . import conv_mfma

    w = conv.weight
    frozen = not (w.requires_grad or (conv.bias is not None and conv.bias.requires_grad))
    if (_USE_MFMA_CONV and _USE_MFMA_CONV_NARROW and frozen and residual is None and bias and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv_mfma.narrow_supported(x, w)):
        # 3 / 4 / 8 channels on one side (the ends of both networks): the same kernel on zero-padded channels
        key = (w.data_ptr(), w._version, None if conv.bias is None else conv.bias._version)
        packed = getattr(conv, "_dm4d_narrow", None)
        if packed is None or packed[0] != key:
            wp, bp = conv_mfma.pad_weight(w, conv.bias)
            packed = conv._dm4d_narrow = (key, conv_mfma.pack_weight(wp), bp, conv_mfma.pack_weight_transposed(wp))
        xp = conv_mfma.pad_channels(x, int(packed[1].shape[3]))
        if torch.is_grad_enabled() and x.requires_grad:
            y = conv_mfma.conv3x3_frozen(xp, packed[1], packed[3], packed[2], None)
        else:
            y = conv_mfma.conv3x3(xp, packed[1], packed[2])
        return y[:, :w.shape[0]]
    if (_USE_MFMA_CONV and frozen and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
            and conv_mfma.supported(x, w) and (residual is None or (residual.dtype == x.dtype and residual.shape[1] == w.shape[0]))):
        need_grad = torch.is_grad_enabled() and (x.requires_grad or (residual is not None and residual.requires_grad))
        key = (w.data_ptr(), w._version)
        packed = getattr(conv, "_dm4d_ohwi", None)
        if packed is None or packed[0] != key:
            packed = conv._dm4d_ohwi = [key, conv_mfma.pack_weight(w), None]
        b = conv.bias if bias else None
        res = None if residual is None else _to_nhwc(residual)
        if not need_grad:
            return conv_mfma.conv3x3(x, packed[1], b, res)
        if w.shape[1] % 32 == 0 and w.shape[0] % 32 == 0:          # the data gradient swaps the channel counts
            if packed[2] is None:
                packed[2] = conv_mfma.pack_weight_transposed(w)
            return conv_mfma.conv3x3_frozen(x, packed[1], packed[2], b, res)
    # the library branch: on a device tensor inside the guidance step this is a REGRESSION (4.4 ms of convolutions silently back
    # on MIOpen): counted like the GroupNorm fallbacks, an error under fused_norm.expect_fused() + DM4D_STRICT_FUSED=1
    _library_fallback("conv3x3", x, _why_not_mfma_conv(conv, x, frozen))
    if residual is not None and bias and _fold_bias(conv, x):
        return add_bias(residual, _conv_nobias(conv, x), conv.bias)      # (one fused kernel for skip + bias)
    y = F.conv2d(x, w, conv.bias if bias else None, conv.stride, conv.padding)
    return y if residual is None else residual + y


def _conv3x3_stride2(conv, x, pad):
    """The two Downsample convolutions (stride 2; pad 1: the UNet's, pad 0 + one zero behind each axis: the VAE encoder's) with
    frozen parameters on a HIP device: the MFMA implicit-GEMM kernel reads the input with the stride and zero-fills what lies
    outside, so the VAE's padded copy (a fill + a copy of up to 67 MB) is never made.  Anything else: the library path, as the
    reference writes it."""
    from . import conv_mfma

    w = conv.weight
    frozen = not (w.requires_grad or (conv.bias is not None and conv.bias.requires_grad))
    if _USE_MFMA_CONV and _USE_MFMA_CONV_S2 and frozen and conv.groups == 1 and conv.dilation == (1, 1) and conv_mfma.supported(x, w):
        key = (w.data_ptr(), w._version)
        packed = getattr(conv, "_dm4d_ohwi", None)
        if packed is None or packed[0] != key:
            packed = conv._dm4d_ohwi = [key, conv_mfma.pack_weight(w), None]
        if torch.is_grad_enabled() and x.requires_grad:
            if packed[2] is None and _USE_MFMA_S2_DGRAD and pad == 0:
                packed[2] = conv_mfma.pack_weight_s2_dgrad(w)
            return conv_mfma.conv3x3_stride2_frozen(x, w, packed[1], conv.bias, pad, packed[2] if pad == 0 else None)
        return conv_mfma.conv3x3(x, packed[1], conv.bias, None, stride=2, pad=pad)
    _library_fallback("conv3x3_stride2", x, _why_not_mfma_conv(conv, x, frozen))
    return conv(x) if pad else conv(F.pad(x, (0, 1, 0, 1)))


def _conv1x1(conv, x):
    """A 1x1 convolution; on a channels-last tensor a GEMM over the [B, H*W, C] token view of the same memory."""
    if x.is_cuda and is_channels_last(x):
        from . import conv_mfma

        B, Cc, Hh, Ww = x.shape
        tok, w = x.permute(0, 2, 3, 1).reshape(B, Hh * Ww, Cc), conv.weight.flatten(1)
        if (B * Hh * Ww >= MFMA_CONV1X1_MIN_ROWS and not torch.is_grad_enabled() and not conv.weight.requires_grad and tok.is_contiguous()
                and w.is_contiguous() and conv_mfma.linear_supported(tok, w)):
            y = conv_mfma.linear(tok, w, conv.bias)
        else:
            y = F.linear(tok, w, conv.bias)
        return y.view(B, Hh, Ww, -1).permute(0, 3, 1, 2)
    return conv(x)


_FREQS = {}


def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    key = (t.device, half, max_period)
    freqs = _FREQS.get(key)          # a constant of (dim, max_period): four launches per step otherwise
    if freqs is None:
        freqs = _FREQS[key] = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class Upsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return _conv3x3(self.conv, F.interpolate(x, scale_factor=2, mode="nearest"))


class Downsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.op = nn.Conv2d(ch, ch, 3, stride=2, padding=1)

    def forward(self, x):
        return _conv3x3_stride2(self.op, x, 1)


class ResBlock(nn.Module):
    def __init__(self, ch, emb_ch, out_ch):
        super().__init__()
        self.in_layers = nn.Sequential(GroupNorm32(32, ch), nn.SiLU(), nn.Conv2d(ch, out_ch, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_ch, out_ch))
        self.out_layers = nn.Sequential(GroupNorm32(32, out_ch), nn.SiLU(), nn.Dropout(0.0),
                                        nn.Conv2d(out_ch, out_ch, 3, padding=1))
        nn.init.zeros_(self.out_layers[3].weight)
        nn.init.zeros_(self.out_layers[3].bias)
        self.skip_connection = nn.Identity() if ch == out_ch else nn.Conv2d(ch, out_ch, 1)

    def forward(self, x, emb):
        conv1, conv2 = self.in_layers[2], self.out_layers[3]
        h = group_norm(self.in_layers[0], x, silu=True, float32=True)
        skip = x if isinstance(self.skip_connection, nn.Identity) else _conv1x1(self.skip_connection, x)
        # h + emb[:, :, None, None], GroupNorm, SiLU (openaimodel.py:259-275) in one operator; Dropout(0) is the identity
        if _fold_bias(conv1, h):
            add = self.__dict__.get("_emb_add")      # emb_layers(emb) + conv1.bias of ALL ResBlocks from one GEMM (UNetModel._batched_small_gemms)
            if add is None:
                add = self.emb_layers(emb) + conv1.bias
            h = group_norm(self.out_layers[0], _conv3x3(conv1, h, bias=False), silu=True, add=add, float32=True)
            return _conv3x3(conv2, h, bias=True, residual=skip)        # skip + bias in the convolution's epilogue
        h = group_norm(self.out_layers[0], conv1(h), silu=True, add=self.emb_layers(emb), float32=True)
        return skip + conv2(h)


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        context_dim = query_dim if context_dim is None else context_dim
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(0.0))

    def single_token(self, context):
        """Cross-attention over ONE context token (Zero123's conditioning is a single CLIP + pose embedding per sample,
        ddpm.py:1953-1956): the softmax over a single key is exactly 1, so every query's output is to_out(to_v(context)),
        whatever the queries are -- [B, 1, query_dim], to be broadcast over the positions.  The same numbers as the general
        path without its query projection, attention and output GEMM over all positions."""
        tok = self.__dict__.get("_tok")            # to_out(to_v(context)) of ALL cross-attentions from 1 + 3 GEMMs (UNetModel._batched_small_gemms)
        if tok is not None:
            return tok[:, None, :]
        v = self.__dict__.get("_v_token")
        return self.to_out(self.to_v(context) if v is None else v)

    def _qkv_weight(self):
        """[to_q; to_k; to_v] as one matrix (self-attention under no_grad with frozen parameters: one GEMM instead of three
        launch-bound ones at M = B x L, N = K = 320 .. 1280)."""
        ws = (self.to_q.weight, self.to_k.weight, self.to_v.weight)
        key = tuple((w.data_ptr(), w._version) for w in ws)
        c = self.__dict__.get("_qkv_cache")
        if c is None or c[0] != key:
            with torch.no_grad():
                c = (key, torch.cat(ws).contiguous())
            self.__dict__["_qkv_cache"] = c
        return c[1]

    def forward(self, x, context=None):
        ctx = x if context is None else context
        B, L, _ = x.shape
        if context is not None and ctx.shape[1] == 1 and SINGLE_TOKEN_SHORTCUT:
            return self.single_token(ctx).expand(B, L, -1)
        return self.to_out(self.attend(x, ctx))

    def attend(self, x, ctx):
        """softmax(q k^T / sqrt(d)) v with the heads merged again: everything of the layer before `to_out`."""
        B, L, _ = x.shape
        h = self.heads
        if (FUSE_QKV and x is ctx and x.is_cuda and not torch.is_grad_enabled() and self.to_q.weight.shape == self.to_k.weight.shape
                and not any(w.requires_grad for w in (self.to_q.weight, self.to_k.weight, self.to_v.weight))):
            wqkv = self._qkv_weight()
            if B * L >= MFMA_LINEAR_MIN_ROWS and x.dtype == torch.float16 and x.is_contiguous() and conv_mfma.linear_supported(x, wqkv):
                qkv = conv_mfma.linear(x, wqkv).view(B, L, 3, h, -1)
            else:
                if B * L >= MFMA_LINEAR_MIN_ROWS and x.dtype == torch.float16:     # (below the threshold the library GEMM is the CHOICE)
                    _library_fallback("linear_qkv", x, f"contiguous {x.is_contiguous()} K {x.shape[-1]} N {wqkv.shape[0]}")
                qkv = F.linear(x, wqkv).view(B, L, 3, h, -1)                     # the same products; q, k, v are strided views
            if MFMA_ATTENTION and conv_mfma.attention_supported(qkv):
                return conv_mfma.attention_qkv(qkv)          # csrc/attention.hip: 78 -> 33 us at 1024 tokens, 18 -> 11 at 256, 8.5 -> 7 at 64
            # (other head sizes, or fewer than 64 tokens -- the UNet's 4 x 4 level -- are outside the kernel's domain: a property of
            # the model, not a regression)
            if x.dtype == torch.float16 and int(qkv.shape[4]) in (40, 64, 80, 160) and L >= 64:
                _library_fallback("attention", x, f"MFMA_ATTENTION {MFMA_ATTENTION} qkv {tuple(qkv.shape)}")
            o = F.scaled_dot_product_attention(qkv[:, :, 0].transpose(1, 2), qkv[:, :, 1].transpose(1, 2), qkv[:, :, 2].transpose(1, 2))
            return o.transpose(1, 2).reshape(B, L, -1)
        q = self.to_q(x).view(B, L, h, -1).transpose(1, 2)
        k = self.to_k(ctx).view(B, ctx.shape[1], h, -1).transpose(1, 2)
        v = self.to_v(ctx).view(B, ctx.shape[1], h, -1).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v)
        return o.transpose(1, 2).reshape(B, L, -1)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        return geglu(self.proj(x))


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim))

    def forward(self, x):
        return self.net(x)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, context_dim):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, heads, dim_head)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, context_dim, heads, dim_head)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)

    def _fused_no_grad(self, x, context):
        """The block without gradients on a HIP device, float16, one context token: the same sums as `forward`, but the residual
        adds ride in the GEMMs -- the epilogue of the MFMA kernel (conv_mfma.linear: bias + residual, GEGLU) where that is the
        faster one (MFMA_LINEAR_*_MIN_ROWS), else the library's `addmm` with C operand = residual + that GEMM's bias -- and the
        LayerNorms come from the kernel that also prepares that operand (fused_norm.add_layer_norm): per block 2 launches
        instead of 2 LayerNorms + 3 adds."""
        B, L, Cc = x.shape
        out1, ff1, ff2 = self.attn1.to_out[0], self.ff.net[0].proj, self.ff.net[2]
        ok = conv_mfma.linear_supported(x, out1.weight)
        own_res = ok and B * L >= MFMA_LINEAR_RES_MIN_ROWS                                 # attention output + residual on the MFMA kernel
        own_big = ok and B * L >= MFMA_LINEAR_MIN_ROWS and ff1.out_features % 256 == 0     # ... and the GEGLU projection (q/k/v: attend)
        tok = self.attn2.single_token(context)                                            # [B, 1, C]: the whole cross-attention
        if own_res:    # the MFMA kernel: bias and residual in the epilogue (no prepared C operand)
            n1, _ = add_layer_norm(self.norm1, x, None, None, want_sum=False)
            o = self.attn1.attend(n1, n1)
            x1 = conv_mfma.linear(o, out1.weight, out1.bias, residual=x)                  # attn1(norm1(x)) + x
            n3, x2b = add_layer_norm(self.norm3, x1, tok, ff2.bias)                       # norm3(x1 + tok) | x1 + tok + b_ff
            if own_big:                                                                   # GEGLU in the projection's epilogue
                key = (ff1.weight.data_ptr(), ff1.weight._version, ff1.bias._version)
                pk = self.__dict__.get("_geglu_packed")
                if pk is None or pk[0] != key:
                    pk = self.__dict__["_geglu_packed"] = (key,) + conv_mfma.pack_geglu(ff1.weight, ff1.bias)
                g = conv_mfma.linear(n3, pk[1], pk[2], act="geglu")
            else:
                g = self.ff.net[0](n3)
            if B * L >= MFMA_FF2_MIN_ROWS and g.is_contiguous() and conv_mfma.linear_supported(g, ff2.weight):
                return conv_mfma.linear(g, ff2.weight, None, residual=x2b)                    # (x2b already carries ff2's bias)
            return x2b.view(-1, Cc).addmm_(g.view(B * L, -1), ff2.weight.t()).view(B, L, Cc)      # ff(norm3(x2)) + x2
        n1, xb = add_layer_norm(self.norm1, x, None, out1.bias)                           # norm1(x) | x + b_out
        o = self.attn1.attend(n1, n1)
        # (in place: xb / x2b are this function's own temporaries, and torch.addmm would first COPY its C operand to the result)
        x1 = xb.view(-1, Cc).addmm_(o.view(B * L, -1), out1.weight.t()).view(B, L, Cc)            # attn1(norm1(x)) + x
        n3, x2b = add_layer_norm(self.norm3, x1, tok, ff2.bias)                           # norm3(x1 + tok) | x1 + tok + b_ff
        g = self.ff.net[0](n3)
        return x2b.view(-1, Cc).addmm_(g.view(B * L, -1), ff2.weight.t()).view(B, L, Cc)          # ff(norm3(x2)) + x2

    def forward(self, x, context):
        if (FUSE_ADD_LAYERNORM and SINGLE_TOKEN_SHORTCUT and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float16
                and x.is_contiguous() and context is not None and context.shape[1] == 1 and self.norm1.weight.dtype == torch.float16
                and x.shape[-1] % 8 == 0 and x.shape[-1] <= 2048):
            return self._fused_no_grad(x, context)
        x = self.attn1(self.norm1(x)) + x
        if context is not None and context.shape[1] == 1 and SINGLE_TOKEN_SHORTCUT:
            x = x + self.attn2.single_token(context)          # (norm2 only feeds the queries, which do not matter here)
        else:
            x = self.attn2(self.norm2(x), context) + x
        return self.ff(self.norm3(x)) + x


class SpatialTransformer(nn.Module):
    def __init__(self, ch, heads, dim_head, context_dim, depth=1):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(32, ch, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(ch, inner, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, dim_head, context_dim)
                                                 for _ in range(depth)])
        self.proj_out = nn.Conv2d(inner, ch, 1)
        nn.init.zeros_(self.proj_out.weight)
        nn.init.zeros_(self.proj_out.bias)

    def forward(self, x, context):
        B, Cc, Hh, Ww = x.shape
        h = _conv1x1(self.proj_in, group_norm(self.norm, x))
        h = h.flatten(2).transpose(1, 2)                      # b (h w) c  (a view of a channels-last tensor)
        for blk in self.transformer_blocks:
            h = blk(h, context)
        w = self.proj_out.weight
        if (h.is_cuda and B * Hh * Ww >= MFMA_LINEAR_RES_MIN_ROWS and not torch.is_grad_enabled() and h.is_contiguous() and h.dtype == torch.float16
                and is_channels_last(x) and x.dtype == torch.float16 and not w.requires_grad and conv_mfma.linear_supported(h, w.flatten(1))):
            # proj_out + bias + the residual in one launch (tokens [B, HW, C] and NHWC pixels are the same memory)
            y = conv_mfma.linear(h, w.flatten(1), self.proj_out.bias, residual=x.permute(0, 2, 3, 1).reshape(B, Hh * Ww, Cc))
            return y.view(B, Hh, Ww, -1).permute(0, 3, 1, 2)
        h = h.transpose(1, 2).reshape(B, -1, Hh, Ww)
        return _conv1x1(self.proj_out, h) + x


class _Seq(nn.Sequential):
    """TimestepEmbedSequential: routes emb to ResBlocks and context to SpatialTransformers."""

    def forward(self, x, emb, context):
        for layer in self:
            if isinstance(layer, ResBlock):
                x = layer(x, emb)
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context)
            elif isinstance(layer, nn.Conv2d) and layer.kernel_size == (3, 3) and layer.stride == (1, 1):
                x = _conv3x3(layer, x)                    # (conv_in: 8 -> 320 channels)
            else:
                x = layer(x)
        return x


class UNetModel(nn.Module):
    def __init__(self, in_channels=8, out_channels=4, model_channels=320, attention_resolutions=(4, 2, 1),
                 num_res_blocks=2, channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768, transformer_depth=1):
        super().__init__()
        self.model_channels = model_channels
        self.channels_last = True             # activations NHWC on a HIP device (see the module docstring)
        emb = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, emb), nn.SiLU(), nn.Linear(emb, emb))
        att = lambda ch: SpatialTransformer(ch, num_heads, ch // num_heads, context_dim, transformer_depth)
        self.input_blocks = nn.ModuleList([_Seq(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
        chans, ch, ds = [model_channels], model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, emb, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(att(ch))
                self.input_blocks.append(_Seq(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(_Seq(Downsample(ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = _Seq(ResBlock(ch, emb, ch), att(ch), ResBlock(ch, emb, ch))
        self.output_blocks = nn.ModuleList()
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [ResBlock(ch + chans.pop(), emb, model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(att(ch))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch))
                    ds //= 2
                self.output_blocks.append(_Seq(*layers))
        self.out = nn.Sequential(GroupNorm32(32, ch), nn.SiLU(), nn.Conv2d(model_channels, out_channels, 3, padding=1))
        nn.init.zeros_(self.out[2].weight)
        nn.init.zeros_(self.out[2].bias)

    def _batched_small_gemms(self, emb, context):
        """The UNet's M = batch GEMMs that do not depend on the activations, as 2 + 3 launches instead of 22 + 16 + 16 (each a
        7 us launch-bound kernel + its SiLU / bias add): the timestep projection of every ResBlock (`emb_layers` = SiLU,
        Linear(emb) -- the same `emb` for all, openaimodel.py:259-266; the first convolution's bias, which this mirror folds
        into the same per-(sample, channel) term, rides in the concatenated bias) and, when the context is ONE token, the
        value projection of every cross-attention (one GEMM) followed by its output projection (CrossAttention.single_token;
        one batched GEMM per width: 320 / 640 / 1280).  The same products row by row; the blocks get slices (views) of the
        results.  Frozen parameters on a HIP device only."""
        cache = self.__dict__.get("_small_gemm_cache")
        res = [m for m in self.modules() if isinstance(m, ResBlock)] if cache is None else cache["res"]
        # (cross-attentions ordered by width: the tokens of one width are ONE batched GEMM over a contiguous column range)
        att = sorted((m.attn2 for m in self.modules() if isinstance(m, BasicTransformerBlock)),
                     key=lambda a: a.to_v.out_features) if cache is None else cache["att"]
        key = tuple((b.emb_layers[1].weight.data_ptr(), b.emb_layers[1].weight._version, b.in_layers[2].bias._version) for b in res) + \
            tuple((a.to_v.weight.data_ptr(), a.to_v.weight._version, a.to_out[0].weight.data_ptr(), a.to_out[0].weight._version,
                   a.to_out[0].bias._version) for a in att)
        if cache is None or cache["key"] != key:
            with torch.no_grad():
                groups, o = [], 0          # (first column, modules, stacked to_out weights transposed [n, inner, C], biases [n, 1, C])
                for a in att:
                    if not groups or groups[-1][1][0].to_v.out_features != a.to_v.out_features or \
                            groups[-1][1][0].to_out[0].out_features != a.to_out[0].out_features:
                        groups.append([o, []])
                    groups[-1][1].append(a)
                    o += a.to_v.out_features
                groups = [(o0, ms, torch.stack([a.to_out[0].weight.t() for a in ms]).contiguous(),
                           torch.stack([a.to_out[0].bias for a in ms])[:, None, :].contiguous()) for o0, ms in groups]
                cache = dict(key=key, res=res, att=att, groups=groups,
                             We=torch.cat([b.emb_layers[1].weight for b in res]).contiguous(),
                             be=torch.cat([b.emb_layers[1].bias + b.in_layers[2].bias for b in res]).contiguous(),
                             Wv=torch.cat([a.to_v.weight for a in att]).contiguous())
            self.__dict__["_small_gemm_cache"] = cache
        e_all = F.linear(F.silu(emb), cache["We"], cache["be"])                  # [B, sum C_out]
        o = 0
        for b in res:
            c = b.emb_layers[1].out_features
            b.__dict__["_emb_add"] = e_all[:, o:o + c]
            o += c
        if context is not None and context.shape[1] == 1 and SINGLE_TOKEN_SHORTCUT:
            B = context.shape[0]
            v_all = F.linear(context, cache["Wv"])                               # [B, 1, sum inner]
            for o0, ms, WoT, bo in cache["groups"]:
                n, inner = len(ms), ms[0].to_v.out_features
                v = v_all[:, 0, o0:o0 + n * inner].view(B, n, inner).transpose(0, 1)      # [n, B, inner]: a strided view
                tok = torch.baddbmm(bo, v, WoT)                                  # to_out(to_v(context)) of the n blocks: [n, B, C]
                for i, a in enumerate(ms):
                    a.__dict__["_tok"] = tok[i]
        return res, att

    def precompute(self, timesteps, context, dtype):
        """Everything of `forward` that does not depend on the activations: the timestep embedding and -- frozen parameters,
        channels-last, no autograd, on a device -- the batched small GEMMs (`_batched_small_gemms`).  The guidance's one-graph step
        replays this as its own small graph on a side stream, ahead of the step (TemporalStableZero123Guidance._sds_graph); the
        result goes to `forward(..., pre=...)`."""
        emb = self.time_embed(timestep_embedding(timesteps, self.model_channels).type(dtype))
        frozen = not any(p.requires_grad for p in (self.time_embed[0].weight, self.out[2].weight))
        touched = self._batched_small_gemms(emb, context) if (emb.is_cuda and self.channels_last and frozen and BATCH_SMALL_GEMMS
                                                              and not torch.is_grad_enabled()) else None
        return emb, touched

    def forward(self, x, timesteps, context, pre=None):
        """`pre`: the result of `precompute` for the same (timesteps, context), computed here when absent."""
        emb, touched = self.precompute(timesteps, context, x.dtype) if pre is None else pre
        hs, h = [], (_to_nhwc(x) if x.is_cuda and self.channels_last else x)
        try:
            for m in self.input_blocks:
                h = m(h, emb, context)
                hs.append(h)
            h = self.middle_block(h, emb, context)
            for m in self.output_blocks:
                h = m(torch.cat([h, hs.pop()], dim=1), emb, context)
        finally:
            if touched is not None:
                for b in touched[0]:
                    b.__dict__.pop("_emb_add", None)
                for a in touched[1]:
                    a.__dict__.pop("_tok", None)
        h = _conv3x3(self.out[2], group_norm(self.out[0], h.type(x.dtype), silu=True, float32=True))
        return h.contiguous()


# ----------------------------------------------------------------------------- VAE encoder
class _VaeRes(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(32, cout, eps=1e-6)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x):
        # (x reaches the output around norm1 -- as it is or through the 1 x 1 shortcut: that branch's gradient is added inside the
        # norm's backward kernel, fused_norm.group_norm(skip=True))
        h, skip = group_norm(self.norm1, x, silu=True, skip=True)
        if hasattr(self, "nin_shortcut"):
            skip = _conv1x1(self.nin_shortcut, skip)
        if _fold_bias(self.conv1, h):
            h = group_norm(self.norm2, _conv3x3(self.conv1, h, bias=False), silu=True, add=self.conv1.bias)
            return _conv3x3(self.conv2, h, bias=True, residual=skip)                  # (Dropout(0) is the identity)
        h = self.conv2(self.dropout(group_norm(self.norm2, self.conv1(h), silu=True)))
        return skip + h


class _VaeAttn(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.norm = nn.GroupNorm(32, ch, eps=1e-6)
        self.q, self.k, self.v, self.proj_out = (nn.Conv2d(ch, ch, 1) for _ in range(4))

    def forward(self, x):
        B, Cc, Hh, Ww = x.shape
        h, x = group_norm(self.norm, x, skip=True)
        q, k, v = (_conv1x1(f, h).flatten(2).transpose(1, 2) for f in (self.q, self.k, self.v))            # b (hw) c
        if x.is_cuda and VAE_ATTENTION_BMM:
            # ONE head of 512 channels over 1024 positions: as the reference writes it (model.py AttnBlock: bmm, softmax, bmm).
            # The library's fused attention is built for head dimensions <= 256: its backward took 0.32 ms of the SDS step
            # here, the three batched GEMMs and their gradients take 0.1 ms.
            w_ = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * (int(Cc) ** -0.5), dim=2)
            o = torch.bmm(w_, v).transpose(1, 2).reshape(B, Cc, Hh, Ww)
        else:
            o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0].transpose(1, 2).reshape(B, Cc, Hh, Ww)
        return x + _conv1x1(self.proj_out, o)


class _VaeDown(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=0)

    def forward(self, x):
        return _conv3x3_stride2(self.conv, x, 0)


class _Level(nn.Module):
    pass


class VaeEncoder(nn.Module):
    def __init__(self, ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, z_channels=4):
        super().__init__()
        self.channels_last = True
        self.conv_in = nn.Conv2d(in_channels, ch, 3, padding=1)
        self.down = nn.ModuleList()
        cin = ch
        for i, m in enumerate(ch_mult):
            lvl = _Level()
            lvl.block = nn.ModuleList()
            lvl.attn = nn.ModuleList()
            for _ in range(num_res_blocks):
                lvl.block.append(_VaeRes(cin, ch * m))
                cin = ch * m
            if i != len(ch_mult) - 1:
                lvl.downsample = _VaeDown(cin)
            self.down.append(lvl)
        self.mid = _Level()
        self.mid.block_1 = _VaeRes(cin, cin)
        self.mid.attn_1 = _VaeAttn(cin)
        self.mid.block_2 = _VaeRes(cin, cin)
        self.norm_out = nn.GroupNorm(32, cin, eps=1e-6)
        self.conv_out = nn.Conv2d(cin, 2 * z_channels, 3, padding=1)

    def _first(self, x):
        """conv_in; with frozen parameters on a HIP device its data gradient (image <- 128 channels) runs on the HIP kernel."""
        from . import conv_mfma

        c = self.conv_in
        if (_USE_MFMA_CONV and torch.is_grad_enabled() and x.requires_grad and not c.weight.requires_grad and not c.bias.requires_grad
                and is_channels_last(x) and conv_mfma.first_conv_supported(x, c.weight)):
            key = (c.weight.data_ptr(), c.weight._version, c.bias._version)
            packed = getattr(c, "_dm4d_wt", None)
            if packed is None or packed[0] != key:
                wp, bp = conv_mfma.pad_weight(c.weight, c.bias) if _USE_MFMA_CONV_NARROW else (None, None)
                packed = c._dm4d_wt = (key, conv_mfma.pack_weight_transposed(c.weight), None if wp is None else conv_mfma.pack_weight(wp), bp)
            return conv_mfma.conv3x3_first_frozen(x, c.weight, c.bias, packed[1], packed[2], packed[3])
        return c(x)

    def forward(self, x):
        h = self._first(_to_nhwc(x) if x.is_cuda and self.channels_last else x)
        for lvl in self.down:
            for b in lvl.block:
                h = b(h)
            if hasattr(lvl, "downsample"):
                h = lvl.downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        return _conv3x3(self.conv_out, group_norm(self.norm_out, h, silu=True)).contiguous()


class FirstStage(nn.Module):
    """`first_stage_model` restricted to what the SDS step needs: encoder + quant_conv."""

    def __init__(self, **kw):
        super().__init__()
        self.encoder = VaeEncoder(**kw)
        z = kw.get("z_channels", 4)
        self.quant_conv = nn.Conv2d(2 * z, 2 * z, 1)

    def encode_moments(self, x):
        return self.quant_conv(self.encoder(x))


class _DiffusionWrapper(nn.Module):
    def __init__(self, unet):
        super().__init__()
        self.diffusion_model = unet


class Zero123(nn.Module):
    """LatentDiffusion (hybrid conditioning) reduced to the inference surface the guidance uses."""

    def __init__(self, unet_kwargs=None, vae_kwargs=None, scale_factor=0.18215, timesteps=1000, linear_start=0.00085,
                 linear_end=0.0120):
        super().__init__()
        uk = dict(unet_kwargs or {})
        self.model = _DiffusionWrapper(UNetModel(**uk))
        self.first_stage_model = FirstStage(**dict(vae_kwargs or {}))
        ctx = uk.get("context_dim", 768)
        self.cc_projection = nn.Linear(ctx + 4, ctx)
        nn.init.eye_(self.cc_projection.weight[:ctx, :ctx])
        nn.init.zeros_(self.cc_projection.bias)
        self.scale_factor = scale_factor
        betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2
        self.register_buffer("alphas_cumprod", torch.cumprod(1.0 - betas, dim=0).float(), persistent=False)
        # (a float32 copy that `.to(float16)` of the module does not touch: the guidance reads the schedule from its scheduler in
        # float32 whatever the weights' precision, temporal_stable_zero123_guidance.py:156)
        self.__dict__["_alphas_f32"] = self.alphas_cumprod.detach().clone()

    def apply_model(self, x_noisy, t, cond):
        """hybrid conditioning (ddpm.py:1953-1956): concat c_concat on channels, cross-attend to c_crossattn."""
        xc = torch.cat([x_noisy] + cond["c_concat"], dim=1)
        cc = torch.cat(cond["c_crossattn"], dim=1)
        return self.model.diffusion_model(xc, t, context=cc)

    def encode_first_stage_sample(self, x, noise=None):
        mean, logvar = self.first_stage_model.encode_moments(x).chunk(2, dim=1)
        logvar = logvar.clamp(-30.0, 20.0)
        if noise is None:
            noise = torch.randn(mean.shape, pin_memory=mean.is_cuda).to(mean, non_blocking=True)      # the reference samples this on the CPU (distributions.py:37-41); pinned + non-blocking: see encode_images
        return self.scale_factor * (mean + torch.exp(0.5 * logvar) * noise)


def load_zero123_state_dict(model: Zero123, sd):
    """Load an LDM Zero123 checkpoint state dict (keys `model.diffusion_model.*`, `first_stage_model.*`,
    `cc_projection.*`), ignoring what the SDS step never uses (decoder, CLIP, EMA, schedule buffers)."""
    own = model.state_dict()
    take = {k: v for k, v in sd.items() if k in own and tuple(v.shape) == tuple(own[k].shape)}
    missing = [k for k in own if k not in take]
    model.load_state_dict(take, strict=False)
    return missing


# ----------------------------------------------------------------------------- guidance
class _EncodeSample(nn.Module):
    """images in [0,1] + posterior noise -> sampled latents (the graphed half of encode_images)."""

    def __init__(self, model, dtype):
        super().__init__()
        self.model, self.dtype = model, dtype

    def forward(self, imgs, noise):
        return self.model.encode_first_stage_sample((imgs * 2.0 - 1.0).to(self.dtype), noise=noise.to(self.dtype)).to(imgs.dtype)


class _SdsStep(torch.autograd.Function):
    """The WHOLE SDS step as one hipGraph replay (TemporalStableZero123Guidance._sds_graph): images in, loss out; the image
    gradient was computed inside the graph (for an upstream gradient of 1) and is scaled by the upstream gradient here."""

    @staticmethod
    def forward(ctx, imgs, st):
        st.imgs.copy_(imgs)
        st.graph.replay()
        st.serial += 1
        ctx.st, ctx.serial = st, st.serial
        return st.loss.clone(), st.grad_norm.clone()

    @staticmethod
    def backward(ctx, g_loss, g_norm):
        st = ctx.st
        if ctx.serial != st.serial:
            raise RuntimeError("the SDS step's graph was replayed again before this backward: its image gradient is gone "
                               "(call backward before the next guidance step, or construct the guidance with one_graph=False)")
        return st.d_imgs * g_loss, None


class TemporalStableZero123Guidance(nn.Module):
    """`temporal-stable-zero123-guidance`.  `__call__(rgb[B,H,W,3], elevation, azimuth, camera_distances,
    frame_indices, rgb_as_latents=False) -> {"loss_sds", "grad_norm", "min_step", "max_step"}`."""

    host_frame_indices = True      # `frame_indices` (and elevation / azimuth) may be HOST tensors: the one-graph step prefers them

    def __init__(self, model: Zero123, c_crossattn, c_concat, cond_elevation_deg=0.0, cond_azimuth_deg=0.0,
                 guidance_scale=3.0, min_step_percent=0.02, max_step_percent=0.98, grad_clip=None,
                 half_precision_weights=True, use_graphs=True, channels_last=True, one_graph=None):
        super().__init__()
        # The SDS step is ~2000 small launches (UNet forward at 32x32 latents, VAE encoder forward + backward) whose
        # shapes never change: on a HIP device both halves are captured once per batch size as hipGraphs
        # (torch.cuda.CUDAGraph / make_graphed_callables) and replayed -- the step is host-bound otherwise (measured:
        # 30 ms eager for ~2000 launches).  Same kernels, same results; eager is used if capture is not possible.
        self.use_graphs = use_graphs
        self._unet_graphs, self._enc_graphed, self._graph_error = {}, {}, None
        # ONE graph for the whole step (encoder forward, conditioning, noise schedule, UNet, guidance arithmetic, loss, encoder
        # backward to the images): `_sds_graph`.  Off: the encoder and the UNet as two / three graphs with ~80 eager launches
        # between them (the round-3 arrangement; DM4D_SDS_ONE_GRAPH=0 for an A/B).
        self.one_graph = (os.environ.get("DM4D_SDS_ONE_GRAPH", "1") != "0") if one_graph is None else bool(one_graph)
        self._sds_graphs = {}
        # the arithmetic between the networks as two HIP launches inside that graph (float16 weights; csrc/sds_glue.hip) instead
        # of ~70 torch operators on 16 K-element tensors (DM4D_SDS_FUSED_GLUE=0 for an A/B)
        self.fused_glue = os.environ.get("DM4D_SDS_FUSED_GLUE", "1") != "0"
        self.weights_dtype = torch.float16 if half_precision_weights else torch.float32
        self.model = model.to(self.weights_dtype)
        # NHWC activations + filters on a HIP device (module docstring); `channels_last=False` keeps the NCHW library path
        self.model.model.diffusion_model.channels_last = self.model.first_stage_model.encoder.channels_last = bool(channels_last)
        if channels_last:
            self.model.to(memory_format=torch.channels_last)
        for p in self.model.parameters():
            p.requires_grad_(False)
        self.register_buffer("c_crossattn", c_crossattn.to(self.weights_dtype), persistent=False)   # [L,1,ctx] CLIP embeddings
        self.register_buffer("c_concat", c_concat.to(self.weights_dtype), persistent=False)         # [L,4,32,32] VAE modes
        self.cond_elevation_deg, self.cond_azimuth_deg = cond_elevation_deg, cond_azimuth_deg
        self.guidance_scale = guidance_scale
        self.num_train_timesteps = int(model.alphas_cumprod.shape[0])
        self.register_buffer("alphas", model.__dict__["_alphas_f32"].detach().float().cpu().clone(), persistent=False)   # `self.alphas` (:156), float32
        self.grad_clip_val = grad_clip
        self.set_min_max_steps(min_step_percent, max_step_percent)

    def set_min_max_steps(self, min_step_percent=0.02, max_step_percent=0.98):
        self.min_step = int(self.num_train_timesteps * min_step_percent)
        self.max_step = int(self.num_train_timesteps * max_step_percent)

    def encode_images(self, imgs, noise=None):
        """imgs [B,3,256,256] in [0,1] -> sampled latents [B,4,32,32] (imgs.dtype); differentiable w.r.t. imgs."""
        if noise is None:
            # drawn on the CPU like the reference's DiagonalGaussianDistribution.sample (distributions.py:37-41) -- but into PINNED
            # memory and copied without blocking: a pageable host-to-device copy waits for everything queued on the stream, i.e.
            # it stalled the host for the whole previous iteration (10 of its 14.4 ms) and nothing could be enqueued ahead
            noise = torch.randn(imgs.shape[0], 4, imgs.shape[2] // 8, imgs.shape[3] // 8, pin_memory=imgs.is_cuda).to(
                imgs.device, self.weights_dtype, non_blocking=True)
        if self.use_graphs and imgs.is_cuda and imgs.requires_grad and torch.is_grad_enabled() and self._graph_error is None:
            key = tuple(imgs.shape)
            try:
                if key not in self._enc_graphed:
                    enc = _EncodeSample(self.model, self.weights_dtype)
                    sample = (torch.rand(imgs.shape, device=imgs.device, dtype=imgs.dtype).requires_grad_(True), noise.clone())
                    self._enc_graphed[key] = torch.cuda.make_graphed_callables(enc, sample)
                return self._enc_graphed[key](imgs.contiguous(), noise.contiguous())
            except Exception as e:      # noqa: BLE001  (capture is an optimisation: report once, run eagerly)
                self._graph_error = f"{type(e).__name__}: {e}"
        return self.model.encode_first_stage_sample((imgs * 2.0 - 1.0).to(self.weights_dtype), noise=noise.to(self.weights_dtype)).to(imgs.dtype)

    def _unet(self, x, t, cond):
        """model.apply_model under no_grad; replayed from a hipGraph per batch size on a HIP device."""
        if not (self.use_graphs and x.is_cuda and self._graph_error is None):
            return self.model.apply_model(x, t, cond)
        cc, cat = cond["c_crossattn"][0], cond["c_concat"][0]
        key = (tuple(x.shape), tuple(cc.shape))
        try:
            if key not in self._unet_graphs:
                st = dict(x=torch.zeros_like(x), t=torch.zeros_like(t), cc=torch.zeros_like(cc), cat=torch.zeros_like(cat))
                run = lambda: self.model.apply_model(st["x"], st["t"], {"c_crossattn": [st["cc"]], "c_concat": [st["cat"]]})
                cur, side = torch.cuda.current_stream(x.device), torch.cuda.Stream(device=x.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    for _ in range(2):          # library handles, MIOpen / rocBLAS algorithm choices, allocator warm-up
                        run()
                cur.wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = run()
                self._unet_graphs[key] = (graph, st, out)
            graph, st, out = self._unet_graphs[key]
            st["x"].copy_(x); st["t"].copy_(t); st["cc"].copy_(cc); st["cat"].copy_(cat)
            graph.replay()
            return out.clone()
        except Exception as e:          # noqa: BLE001
            self._graph_error = f"{type(e).__name__}: {e}"
            return self.model.apply_model(x, t, cond)

    def _camera_T(self, elevation, azimuth):
        """The four-number relative camera embedding [B,1,4] (float32, on the device of `elevation`)."""
        return torch.stack([torch.deg2rad((90 - elevation) - (90 - self.cond_elevation_deg)),
                            torch.sin(torch.deg2rad(azimuth - self.cond_azimuth_deg)),
                            torch.cos(torch.deg2rad(azimuth - self.cond_azimuth_deg)),
                            torch.deg2rad(90 - torch.full_like(elevation, self.cond_elevation_deg))], dim=-1)[:, None, :]

    def _crossattn_from_T(self, T, idx):
        clip_emb = self.model.cc_projection(torch.cat([self.c_crossattn[idx], T], dim=-1))
        return torch.cat([torch.zeros_like(clip_emb), clip_emb], dim=0)

    def _cond_from_T(self, T, idx):
        """T [B,1,4] in the weights' dtype on the device, idx [B] frame indices."""
        return {"c_crossattn": [self._crossattn_from_T(T, idx)],
                "c_concat": [torch.cat([torch.zeros_like(self.c_concat[idx]), self.c_concat[idx]], dim=0)]}

    @torch.no_grad()
    def get_cond(self, elevation, azimuth, camera_distances, frame_indices=None):
        dev = self.c_crossattn.device
        T = self._camera_T(elevation, azimuth)
        # (elevation / azimuth may be HOST tensors -- DynamicStage keeps them there: the four numbers are then computed on the host
        # and uploaded once, without blocking: a pageable synchronous copy would wait for everything queued on the stream)
        T = T.to(self.weights_dtype).to(dev, non_blocking=True) if T.device.type == "cpu" else T.to(dev, self.weights_dtype)
        idx = frame_indices if frame_indices is not None else torch.zeros(len(T), dtype=torch.long, device=dev)
        return self._cond_from_T(T, idx)

    # ---- the step as ONE graph
    def _sds_graph(self, B, dev, clipped):
        """Static buffers + the captured step for a batch of B views: everything between the 256 x 256 images and (loss, dL/dimages)
        -- what `_forward` does op by op, in the same order with the same kernels, so the replay returns what the eager step
        returns; what changes from step to step (images, the two noises, timesteps, camera embedding, frame indices, the clip
        value) lives in the static buffers, the guidance scale is a constant of the capture (part of the key)."""
        import types

        dt = self.weights_dtype
        st = types.SimpleNamespace(serial=0)
        # (channels-last like the views the image heads hand over ([B, H, W, 3] permuted) and like the encoder wants them: the copy in,
        # the cast and the image gradient on its way back stay plain contiguous passes)
        st.imgs = torch.zeros(B, 3, 256, 256, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        st.post = torch.zeros(B, 4, 32, 32, device=dev, dtype=dt)
        with torch.no_grad():
            # the noise buffer with the STRIDES the latents have (a channel slice of the NHWC moments): `randn_like(latents)` deals its
            # numbers in memory order, so only a buffer of the same layout receives the eager step's noise element for element
            st.noise = torch.zeros_like(self.model.encode_first_stage_sample((st.imgs * 2.0 - 1.0).to(dt), noise=st.post).to(st.imgs.dtype))
        st.t = torch.full((B,), self.min_step, dtype=torch.long, device=dev)
        st.T = torch.zeros(B, 1, 4, device=dev, dtype=dt)
        st.fidx = torch.zeros(B, dtype=torch.long, device=dev)
        st.clip = torch.ones((), device=dev)
        scale = self.guidance_scale
        fused = self.fused_glue and dt == torch.float16

        def run_fused():
            # the arithmetic between the networks as two launches (csrc/sds_glue.hip: the same expressions with the same roundings)
            import ctypes as C_

            from . import _lib

            L = _lib.lib()
            ptr = lambda v: C_.c_void_p(v.data_ptr())
            strides = lambda v: (C_.c_int64 * 4)(*v.stride())
            stream = torch.cuda.current_stream(dev).cuda_stream
            unet = self.model.model.diffusion_model
            pre, ctx = (None, None) if st.pre is None else st.pre       # (the conditioning graph's static results, run_pre)
            moments = self.model.first_stage_model.encode_moments((st.imgs * 2.0 - 1.0).to(dt))
            with torch.no_grad():
                x_in = torch.empty(2 * B, 8, 32, 32, device=dev, dtype=dt, memory_format=torch.channels_last)
                t2 = torch.empty(2 * B, dtype=torch.long, device=dev)
                _lib.check(L.dm4d_sds_prepare(B, 32, 32, float(self.model.scale_factor), ptr(moments), strides(moments), ptr(st.post),
                                              strides(st.post), ptr(st.noise), strides(st.noise), ptr(st.latents), strides(st.latents),
                                              ptr(st.t), ptr(self.alphas), ptr(self.c_concat), strides(self.c_concat),
                                              ptr(st.fidx), ptr(x_in), strides(x_in), ptr(t2), stream), "dm4d_sds_prepare")
                pred = unet(x_in, t2, context=self._crossattn_from_T(st.T, st.fidx) if ctx is None else ctx, pre=pre)
                d_mom = torch.empty(moments.shape, device=dev, dtype=dt)      # (contiguous, like the `cat` autograd builds there: the library picks the
                                                                              #  quant_conv's backward kernel by the gradient's layout)
                loss, gnorm = torch.empty((), device=dev), torch.empty((), device=dev)
                _lib.check(L.dm4d_sds_finish(B, 32, 32, float(self.model.scale_factor), float(scale), ptr(pred), strides(pred),
                                             ptr(st.latents), strides(st.latents), ptr(st.noise), strides(st.noise), ptr(st.t),
                                             ptr(self.alphas), ptr(st.clip) if clipped else None, ptr(moments),
                                             strides(moments), ptr(st.post), strides(st.post), ptr(d_mom), strides(d_mom), ptr(loss),
                                             ptr(gnorm), stream), "dm4d_sds_finish")
            (d_imgs,) = torch.autograd.grad(moments, st.imgs, grad_outputs=d_mom)
            return loss, gnorm, d_imgs

        def run_ops():
            latents = self.model.encode_first_stage_sample((st.imgs * 2.0 - 1.0).to(dt), noise=st.post).to(st.imgs.dtype)
            with torch.no_grad():
                cond = self._cond_from_T(st.T, st.fidx)
                ac = self.alphas[st.t].view(-1, 1, 1, 1)
                noisy = ac.sqrt() * latents + (1 - ac).sqrt() * st.noise
                pred = self.model.apply_model(torch.cat([noisy] * 2).to(dt), torch.cat([st.t] * 2), cond)
                unc, cnd = pred.float().chunk(2)
                pred = unc + scale * (cnd - unc)
                grad = torch.nan_to_num((1 - ac) * (pred - st.noise))
                if clipped:
                    grad = grad.clamp(-st.clip, st.clip)
                target = latents - grad
            loss = 0.5 * F.mse_loss(latents, target, reduction="sum") / B
            (d_imgs,) = torch.autograd.grad(loss, st.imgs)
            return loss.detach(), grad.norm(), d_imgs

        run = run_fused if fused else run_ops
        st.fused_glue = fused
        if fused:
            st.latents = torch.zeros_like(st.noise)
            if not (self.c_concat.is_contiguous() and self.alphas.dtype == torch.float32 and tuple(self.c_concat.shape[1:]) == (4, 32, 32)):
                raise ValueError("fused SDS glue: c_concat must be a contiguous [L,4,32,32] tensor")
            # the two glue kernels read c_concat[fidx] and alphas[t] through raw pointers inside a captured graph, where an out-of-range
            # index or a buffer left on the host is a silent out-of-bounds read or a GPU fault (the torch indexing they replace would
            # raise): the buffers are pinned to the step's device here, the indices are validated on the host in _stage_inputs
            for name, buf in (("alphas", self.alphas), ("c_concat", self.c_concat)):
                if buf.device != torch.device(dev) or not buf.is_contiguous():
                    raise ValueError(f"fused SDS glue: `{name}` must be a contiguous tensor on {dev} (it is on {buf.device}); move the guidance with .to(device)")
            if self.alphas.numel() < self.max_step + 1:
                raise ValueError(f"fused SDS glue: the noise schedule has {self.alphas.numel()} entries, max_step is {self.max_step}")
            st.glue_ptrs = (self.alphas.data_ptr(), self.c_concat.data_ptr(), int(self.c_concat.shape[0]), int(self.alphas.numel()))

        # The part of the UNet that does not depend on the latents -- camera embedding -> cc_projection, timestep embedding, the batched
        # M = batch GEMMs of every ResBlock / cross-attention (UNetModel.precompute): ~25 launch-bound kernels, ~0.2 ms -- as its OWN small
        # graph, replayed on a side stream as soon as the previous step's graph is done: the host enqueues a guidance call while the
        # device is still rasterising, so these kernels run beside the render kernels instead of at the head of the UNet.  (As a side
        # BRANCH of one graph they cost more than they hid: DESIGN.md section 3 "Round 4".)  DM4D_SDS_PRE_GRAPH=0: one graph, as before.
        st.pre, st.pre_graph = None, None
        split = fused and os.environ.get("DM4D_SDS_PRE_GRAPH", "1") != "0"

        def run_pre():
            with torch.no_grad():
                ctx = self._crossattn_from_T(st.T, st.fidx)
                return self.model.model.diffusion_model.precompute(torch.cat([st.t, st.t]), ctx, dt), ctx

        cur, side = torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(2):          # library handles, algorithm choices, allocator warm-up
                if split:
                    st.pre = run_pre()
                run()
                st.pre = None
        cur.wait_stream(side)
        if split:
            st.pre_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(st.pre_graph):
                st.pre = run_pre()      # (static tensors of the graphs' shared pool, referenced from here for the graphs' lifetime)
            st.side = torch.cuda.Stream(device=dev)
            st.pre_done, st.main_done = torch.cuda.Event(), torch.cuda.Event()
            st.main_done.record(cur)
        st.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(st.graph, **({"pool": st.pre_graph.pool()} if split else {})):
            st.loss, st.grad_norm, st.d_imgs = run()
        return st

    def _stage_inputs(self, B, dev, elevation, azimuth, frame_indices, noise, t):
        """Everything a one-graph step needs besides the images, written into the step's static buffers; with the conditioning graph
        (`_sds_graph`) on the SIDE stream, followed by that graph, when every input comes from the host (or is drawn here): there is
        nothing to wait for but the previous step's graph, which reads the same buffers.  A device input orders the side stream
        behind the caller's, i.e. where the one-graph step had this work.  The random draws are the eager step's, in its order."""
        clip = self.grad_clip_val
        # host -> device inputs through FRESH pinned tensors of the caching host allocator (it keeps a block until the copy that
        # reads it has run; a buffer of our own would be overwritten by the next step while the host runs ahead), already in the
        # device buffer's dtype: a converting cross-device copy stages through pageable memory and blocks
        pinned = lambda v: torch.empty(v.shape, dtype=self.weights_dtype, pin_memory=True).copy_(v)
        pinned_long = lambda v: torch.empty(v.shape, dtype=torch.long, pin_memory=True).copy_(v)
        post = pinned(torch.randn(B, 4, 32, 32))                         # the posterior noise: on the CPU like the reference (encode_images)
        T = self._camera_T(elevation, azimuth)
        key = (B, float(self.guidance_scale), clip is not None)
        if key not in self._sds_graphs:
            self._sds_graphs[key] = self._sds_graph(B, dev, clip is not None)
        st = self._sds_graphs[key]
        if st.fused_glue:
            # what the captured glue kernels will index with, checked where torch's indexing would have raised: host inputs by value,
            # the schedule bounds always (device-resident indices cannot be checked without a synchronisation: the caller's contract)
            n_frames, n_alpha = int(self.c_concat.shape[0]), int(self.alphas.numel())
            if (self.alphas.data_ptr(), self.c_concat.data_ptr(), n_frames, n_alpha) != st.glue_ptrs:
                raise RuntimeError("the guidance's `alphas` / `c_concat` buffers were replaced after the SDS step was captured (their "
                                   "addresses are constants of the graph): clear `_sds_graphs` after moving or reloading the guidance")
            if not (0 <= self.min_step <= self.max_step < n_alpha):
                raise ValueError(f"timestep range [{self.min_step}, {self.max_step}] outside the noise schedule's {n_alpha} entries")
            if frame_indices is not None and frame_indices.device.type == "cpu" and frame_indices.numel() and \
                    not (0 <= int(frame_indices.min()) and int(frame_indices.max()) < n_frames):
                raise IndexError(f"frame_indices {frame_indices.tolist()} outside the {n_frames} conditioning frames")
            if frame_indices is not None and int(frame_indices.numel()) != B:
                raise ValueError(f"{int(frame_indices.numel())} frame indices for {B} views")
            if t is not None and t.device.type == "cpu" and t.numel() and not (0 <= int(t.min()) and int(t.max()) < n_alpha):
                raise IndexError(f"timesteps {t.tolist()} outside the noise schedule's {n_alpha} entries")
        cur = torch.cuda.current_stream(dev)
        with torch.no_grad():
            on_side = st.pre_graph is not None
            if on_side:
                on_host = lambda v: v is None or v.device.type == "cpu"
                st.side.wait_event(st.main_done)
                if not (T.device.type == "cpu" and on_host(frame_indices) and on_host(t) and on_host(noise)):
                    st.side.wait_stream(cur)
            with torch.cuda.stream(st.side) if on_side else contextlib.nullcontext():
                st.post.copy_(post, non_blocking=True)
                st.T.copy_(pinned(T) if T.device.type == "cpu" else T, non_blocking=True)
                if frame_indices is None:
                    st.fidx.zero_()
                else:
                    st.fidx.copy_(pinned_long(frame_indices) if frame_indices.device.type == "cpu" else frame_indices, non_blocking=True)
                if t is None:
                    st.t.random_(self.min_step, self.max_step + 1)
                else:
                    st.t.copy_(pinned_long(t) if t.device.type == "cpu" else t, non_blocking=True)
                if on_side:
                    st.pre_graph.replay()
                if noise is None:
                    st.noise.normal_()
                else:
                    st.noise.copy_(noise, non_blocking=True)
                if clip is not None:
                    st.clip.fill_(float(clip))
                if on_side:
                    st.pre_done.record(st.side)
        return st

    def prefetch(self, elevation, azimuth, frame_indices=None):
        """Stage the NEXT guidance call's conditioning ahead of it -- before the caller enqueues its render: the conditioning graph and
        the step's small inputs then run on the side stream beside the render kernels whatever the host's lead over the device.  Host
        tensors only (elevation / azimuth / frame_indices as the call will pass them; timesteps and noise are drawn here); a no-op
        returning False until the step's graphs exist (the first call captures them) or when the step is not the two-graph one.  The
        call that follows must pass the same tensors (checked) and no `noise` / `t` of its own, else the staging is redone."""
        self._prefetched = None
        B = int(elevation.shape[0])
        st = self._sds_graphs.get((B, float(self.guidance_scale), self.grad_clip_val is not None))
        host = lambda v: v is None or (torch.is_tensor(v) and v.device.type == "cpu")
        if (st is None or st.pre_graph is None or not (host(elevation) and host(azimuth) and host(frame_indices))
                or os.environ.get("DM4D_SDS_PREFETCH", "1") == "0"):      # (A/B switch)
            return False
        self._stage_inputs(B, st.imgs.device, elevation, azimuth, frame_indices, None, None)
        self._prefetched = (st, elevation.clone(), azimuth.clone(), None if frame_indices is None else frame_indices.clone())
        return True

    def _forward_one_graph(self, x, elevation, azimuth, frame_indices, noise, t):
        """x [B,3,256,256] float32 in [0,1] (requires grad)."""
        B, dev = int(x.shape[0]), x.device
        pf, self._prefetched = self.__dict__.get("_prefetched"), None
        same = lambda a, b: (a is None and b is None) or (a is not None and b is not None and a.device == b.device
                                                          and a.shape == b.shape and torch.equal(a, b))
        if (pf is not None and noise is None and t is None and int(pf[0].imgs.shape[0]) == B and same(pf[1], elevation)
                and same(pf[2], azimuth) and same(pf[3], frame_indices)):
            st = pf[0]                                   # staged by prefetch(): nothing left to do but wait for it
        else:
            st = self._stage_inputs(B, dev, elevation, azimuth, frame_indices, noise, t)
        cur = torch.cuda.current_stream(dev)
        if st.pre_graph is not None:
            cur.wait_event(st.pre_done)
        loss, grad_norm = _SdsStep.apply(x, st)
        if st.pre_graph is not None:
            st.main_done.record(cur)
        return {"loss_sds": loss, "grad_norm": grad_norm, "min_step": self.min_step, "max_step": self.max_step}

    def forward(self, rgb, elevation, azimuth, camera_distances, frame_indices=None, rgb_as_latents=False,
                noise=None, t=None, **kwargs):
        # on a HIP device every GroupNorm / residual add / GEGLU of the step is expected on its HIP operator: a layout
        # regression upstream (NCHW activations ...) is reported instead of silently costing the fused kernels
        from .fused_norm import expect_fused

        if not (rgb.is_cuda and self.model.model.diffusion_model.channels_last):      # CPU golden tests / the NCHW library path
            return self._forward(rgb, elevation, azimuth, camera_distances, frame_indices, rgb_as_latents, noise, t)
        with expect_fused():
            return self._forward(rgb, elevation, azimuth, camera_distances, frame_indices, rgb_as_latents, noise, t)

    def _forward(self, rgb, elevation, azimuth, camera_distances, frame_indices=None, rgb_as_latents=False, noise=None, t=None):
        B = rgb.shape[0]
        x = rgb.permute(0, 3, 1, 2)
        if (self.one_graph and self.use_graphs and not rgb_as_latents and rgb.is_cuda and rgb.requires_grad and torch.is_grad_enabled()
                and rgb.dtype == torch.float32 and self._graph_error is None):
            x256 = x if tuple(x.shape[-2:]) == (256, 256) else F.interpolate(x, (256, 256), mode="bilinear", align_corners=False)
            # (the generators' states are only saved around a CAPTURE -- reading the device generator's state is a blocking copy)
            rng = (torch.get_rng_state(), torch.cuda.get_rng_state(rgb.device)) if not self._sds_graphs else None
            try:
                return self._forward_one_graph(x256, elevation, azimuth, frame_indices, noise, t)
            except Exception as e:      # noqa: BLE001  (capture is an optimisation: report once, take the multi-graph / eager step)
                self._graph_error = f"{type(e).__name__}: {e}"
                if rng is not None:
                    torch.set_rng_state(rng[0])
                    torch.cuda.set_rng_state(rng[1], rgb.device)
        if rgb_as_latents:
            latents = F.interpolate(x, (32, 32), mode="bilinear", align_corners=False) * 2 - 1
        else:
            # (a bilinear resize to the size the image already has is the identity: DynamicStage hands over 256 x 256 views)
            latents = self.encode_images(x if tuple(x.shape[-2:]) == (256, 256) else F.interpolate(x, (256, 256), mode="bilinear", align_corners=False))
        cond = self.get_cond(elevation, azimuth, camera_distances, frame_indices)
        if t is None:
            t = torch.randint(self.min_step, self.max_step + 1, [B], dtype=torch.long, device=latents.device)
        with torch.no_grad():
            if noise is None:
                noise = torch.randn_like(latents)
            ac = self.alphas.to(latents.device)[t].view(-1, 1, 1, 1)
            noisy = ac.sqrt() * latents + (1 - ac).sqrt() * noise                      # DDIMScheduler.add_noise
            pred = self._unet(torch.cat([noisy] * 2).to(self.weights_dtype), torch.cat([t] * 2), cond)
        unc, cnd = pred.float().chunk(2)
        pred = unc + self.guidance_scale * (cnd - unc)
        grad = torch.nan_to_num((1 - ac) * (pred - noise))
        if self.grad_clip_val is not None:
            grad = grad.clamp(-self.grad_clip_val, self.grad_clip_val)
        target = (latents - grad).detach()
        loss = 0.5 * F.mse_loss(latents, target, reduction="sum") / B        # d loss / d latents == grad
        return {"loss_sds": loss, "grad_norm": grad.norm(), "min_step": self.min_step, "max_step": self.max_step}

    def update_step(self, epoch, global_step, min_step_percent=None, max_step_percent=None, grad_clip=None):
        if grad_clip is not None:
            self.grad_clip_val = grad_clip
        if min_step_percent is not None and max_step_percent is not None:
            self.set_min_max_steps(min_step_percent, max_step_percent)


StableZero123Guidance = TemporalStableZero123Guidance   # static twin: same step with frame_indices=None
