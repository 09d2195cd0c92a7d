"""The image-space terms of a static-stage iteration as ONE operator (csrc/statichead.hip, include/dm4d.h: dm4d_static_head_*).

What ``SuGaRStatic.training_step`` (custom/threestudio-dreammesh4d/system/sugar_static.py:110-340) and the epilogue of the static
renderer (renderer/diff_sugar_rasterizer_normal.py:196-226) do with the rendered batch -- clamp; the normal map and its mask; the
masked depth; on the reference views the two masked MSEs; on the random views the guidance's input and the three total-variation
terms -- is ~130 torch operators over 5 x 512^2 images forward + backward (1.3 ms of a 13.9 ms iteration).  ``static_head`` returns
(terms [5] = mse_rgb, mse_mask, tv_rgb, tv_depth, tv_normal; half_rgb [n_rnd, H/2, W/2, 3]) with one launch each way; the torch
composition it replaces stays in ``static_stage.StaticStage.iteration`` (CPU tensors, other renderers) and is what
``tests/test_static_stage_gpu.py`` checks it against."""
import torch

from . import _lib


class _StaticHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, depth, alpha, ref_pos, rnd_pos, ref_images, ref_masks, fidx_ref, n_ref, n_rnd, norm):
        L = _lib.lib()
        B, C, H, W = color.shape
        dev = color.device
        c, d, a = color.detach().contiguous(), depth.detach().contiguous(), alpha.detach().contiguous()
        nb = L.dm4d_static_head_blocks(H, W)
        partial = torch.empty(B * nb, 8, dtype=torch.float32, device=dev)
        half = torch.empty(n_rnd, H // 2, W // 2, 3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_static_head_forward(B, H, W, c.data_ptr(), d.data_ptr(), a.data_ptr(), ref_pos.data_ptr(), rnd_pos.data_ptr(),
                                                  ref_images.data_ptr(), ref_masks.data_ptr(), fidx_ref.data_ptr(), n_ref, n_rnd,
                                                  partial.data_ptr(), half.data_ptr() if n_rnd else 0,
                                                  torch.cuda.current_stream(dev).cuda_stream), "dm4d_static_head_forward")
        ctx.save_for_backward(c, d, a, ref_pos, rnd_pos, ref_images, ref_masks, fidx_ref)
        ctx.n = (n_ref, n_rnd)
        from .loss_sum import partial_sums

        return partial_sums(partial, norm), half      # [8] sums -> the five terms (F.mse_loss's / tv_loss's normalisations: `norm` [8][5]), one launch

    @staticmethod
    def backward(ctx, g_terms, g_half):
        L = _lib.lib()
        c, d, a, ref_pos, rnd_pos, ref_images, ref_masks, fidx_ref = ctx.saved_tensors
        n_ref, n_rnd = ctx.n
        B, C, H, W = c.shape
        dev = c.device
        gc, gd, ga = torch.empty_like(c), torch.empty_like(d), torch.empty_like(a)
        gt = torch.zeros(5, dtype=torch.float32, device=dev) if g_terms is None else g_terms.detach().to(torch.float32).contiguous()
        gh = None if g_half is None else g_half.detach().to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_static_head_backward(B, H, W, c.data_ptr(), d.data_ptr(), a.data_ptr(), ref_pos.data_ptr(), rnd_pos.data_ptr(),
                                                   ref_images.data_ptr(), ref_masks.data_ptr(), fidx_ref.data_ptr(), n_ref, n_rnd,
                                                   gt.data_ptr(), 0 if gh is None or not n_rnd else gh.data_ptr(), gc.data_ptr(), gd.data_ptr(),
                                                   ga.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "dm4d_static_head_backward")
        return gc, gd, ga, None, None, None, None, None, None, None, None


_NORM = {}


def _norm_matrix(H, W, n_ref, n_rnd, dev):
    """[8, 5]: sums {mse rgb, mse mask, tv rgb h, w, tv depth h, w, tv normal h, w} -> (mse_rgb, mse_mask, tv_rgb, tv_depth, tv_normal)."""
    key = (H, W, n_ref, n_rnd, str(dev))
    if key not in _NORM:
        m = torch.zeros(8, 5, dtype=torch.float64)
        m[0, 0] = 1.0 / (max(n_ref, 1) * H * W * 3)
        m[1, 1] = 1.0 / (max(n_ref, 1) * H * W)
        for t, c in ((0, 3), (1, 1), (2, 3)):          # threestudio/utils/loss.py:8-16: 2 (h_tv / (c (h - 1) w) + w_tv / (c h (w - 1))) / b
            m[2 + 2 * t, 2 + t] = 2.0 / (c * (H - 1) * W * max(n_rnd, 1))
            m[3 + 2 * t, 2 + t] = 2.0 / (c * H * (W - 1) * max(n_rnd, 1))
        _NORM[key] = m.tolist()          # host numbers: the sum kernel takes the matrix by value
    return _NORM[key]


def static_head(color, depth, alpha, ref_pos, rnd_pos, ref_images, ref_masks, fidx_ref, n_ref, n_rnd):
    """color [B,6,H,W] (RGB | normal), depth / alpha [B,1,H,W] float32 on the HIP device (H, W even); ref_pos / rnd_pos [B] int32 (the
    view's index among the reference / random views, -1 otherwise); ref_images [L,H,W,3], ref_masks [L,H,W,1] float32; fidx_ref
    [n_ref] int64.  Returns (terms [5], half_rgb [n_rnd,H/2,W/2,3]), differentiable in color, depth and alpha:
    terms = (mse(ref m, clamp(rgb) m), mse(m, alpha)) over the reference views, (tv(clamp(rgb)), tv(depth), tv(normal map)) over the
    random views, with depth and the normal map normalize(n) 0.5 alpha + 0.5 receiving gradient only where alpha > 0.99."""
    B, C, H, W = color.shape
    if not (color.is_cuda and C == 6 and all(t.dtype == torch.float32 and t.device == color.device for t in (color, depth, alpha, ref_images, ref_masks))
            and H % 2 == 0 and W % 2 == 0 and tuple(depth.shape) == (B, 1, H, W) and tuple(alpha.shape) == (B, 1, H, W)):
        raise ValueError("static_head: float32 HIP tensors color [B,6,H,W], depth / alpha [B,1,H,W] with even H, W")
    if not (ref_images.is_contiguous() and ref_masks.is_contiguous() and ref_images.dim() == 4 and tuple(ref_images.shape[1:]) == (H, W, 3)
            and tuple(ref_masks.shape) == (ref_images.shape[0], H, W, 1)):
        raise ValueError(f"static_head: ref_images [L,{H},{W},3] / ref_masks [L,{H},{W},1] contiguous, got {tuple(ref_images.shape)} / {tuple(ref_masks.shape)}")
    if not (ref_pos.dtype == torch.int32 and rnd_pos.dtype == torch.int32 and fidx_ref.dtype == torch.int64
            and all(t.device == color.device for t in (ref_pos, rnd_pos, fidx_ref)) and tuple(ref_pos.shape) == (B,) and tuple(rnd_pos.shape) == (B,)):
        raise ValueError("static_head: ref_pos / rnd_pos [B] int32, fidx_ref int64, on the images' device")
    return _StaticHead.apply(color, depth, alpha, ref_pos, rnd_pos, ref_images, ref_masks, fidx_ref, int(n_ref), int(n_rnd),
                             _norm_matrix(H, W, int(n_ref), int(n_rnd), color.device))
