"""The image-space head of a dynamic-stage iteration as ONE operator (csrc/imagehead.hip, include/dm4d.h: dm4d_image_head_*).

What `C/system/sugar_4dgen.py:148-190` does with the rendered batch -- clamp to [0, 1]; MSE against the reference image and mask
on the reference views; hand the random views to the Zero123 guidance, which resizes them to 256 x 256 first -- is ~45 torch
operators over 25 MB tensors forward + backward (0.4 ms of a 13.3 ms iteration at 8 views of 512 x 512).  `image_head` returns
(mse_rgb, mse_mask, half_rgb [n_rnd, H/2, W/2, 3]) with one launch each way; the torch composition it replaces stays in
`dynamic_stage.DynamicStage.iteration` for CPU tensors and is what `tests/test_dynamic_stage_gpu.py` checks it against."""
import torch

from . import _lib


class _ImageHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, alpha, ref_pos, rnd_pos, ref_images, ref_masks, fidx_ref, n_ref, n_rnd):
        L = _lib.lib()
        B, C, H, W = color.shape
        dev = color.device
        c, a = color.detach().contiguous(), alpha.detach().contiguous()
        nb = L.dm4d_image_head_blocks(H, W)
        partial = torch.empty(B, nb, 2, dtype=torch.float32, device=dev)
        half = torch.empty(n_rnd, H // 2, W // 2, 3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_image_head_forward(B, H, W, C, c.data_ptr(), a.data_ptr(), ref_pos.data_ptr(), rnd_pos.data_ptr(),
                                                 ref_images.data_ptr(), ref_masks.data_ptr(), fidx_ref.data_ptr(), n_ref, n_rnd,
                                                 partial.data_ptr(), half.data_ptr() if n_rnd else 0,
                                                 torch.cuda.current_stream(dev).cuda_stream), "dm4d_image_head_forward")
        ctx.save_for_backward(c, a, ref_pos, rnd_pos, ref_images, ref_masks, fidx_ref)
        ctx.n = (n_ref, n_rnd)
        from .loss_sum import partial_sums

        d = float(max(n_ref, 1) * H * W)
        s = partial_sums(partial.view(-1, 2), [[1.0 / (3.0 * d), 0.0], [0.0, 1.0 / d]])      # (one launch: sum + F.mse_loss's normalisation)
        return s[0], s[1], half

    @staticmethod
    def backward(ctx, g_rgb, g_mask, g_half):
        L = _lib.lib()
        c, a, ref_pos, rnd_pos, ref_images, ref_masks, fidx_ref = ctx.saved_tensors
        n_ref, n_rnd = ctx.n
        B, C, H, W = c.shape
        dev = c.device
        gc, ga = torch.empty_like(c), torch.empty_like(a)
        f = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        g_rgb, g_mask, g_half = f(g_rgb), f(g_mask), f(g_half)
        p = lambda t: 0 if t is None else t.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(L.dm4d_image_head_backward(B, H, W, C, c.data_ptr(), a.data_ptr(), ref_pos.data_ptr(), rnd_pos.data_ptr(),
                                                  ref_images.data_ptr(), ref_masks.data_ptr(), fidx_ref.data_ptr(), n_ref, n_rnd,
                                                  p(g_rgb), p(g_mask), p(g_half) if n_rnd else 0, gc.data_ptr(), ga.data_ptr(),
                                                  torch.cuda.current_stream(dev).cuda_stream), "dm4d_image_head_backward")
        return gc, ga, None, None, None, None, None, None, None


def image_head(color, alpha, ref_pos, rnd_pos, ref_images, ref_masks, fidx_ref, n_ref, n_rnd):
    """color [B, C >= 3, H, W], alpha [B, 1, H, W] float32 on the HIP device (H, W even); ref_pos / rnd_pos [B] int32 (the view's
    index among the reference / random views, -1 otherwise); ref_images [L, H, W, 3], ref_masks [L, H, W, 1]; fidx_ref [n_ref]
    int64.  Returns (mse(ref_images[fidx_ref], clamp(rgb)[ref]), mse(alpha[ref], ref_masks[fidx_ref]),
    resize_to_half(clamp(rgb)[rnd]) as [n_rnd, H/2, W/2, 3]), differentiable in color and alpha."""
    if not (color.is_cuda and color.dtype == torch.float32 and alpha.dtype == torch.float32 and color.shape[2] % 2 == 0 and color.shape[3] % 2 == 0):
        raise ValueError("image_head: float32 HIP tensors with even H, W")
    if not (ref_images.is_contiguous() and ref_masks.is_contiguous() and ref_pos.dtype == torch.int32 and rnd_pos.dtype == torch.int32
            and fidx_ref.dtype == torch.int64):
        raise ValueError("image_head: contiguous reference images / masks, int32 positions, int64 frame indices")
    # the kernel reads the references through raw float32 pointers: a float64 / uint8 frame stack (numpy / cv2 images) would be
    # read as garbage where the torch composition this operator replaces would have promoted it
    H, W = int(color.shape[2]), int(color.shape[3])
    if not (ref_images.dtype == torch.float32 and ref_masks.dtype == torch.float32 and ref_images.device == color.device
            and ref_masks.device == color.device and ref_images.dim() == 4 and tuple(ref_images.shape[1:]) == (H, W, 3)
            and tuple(ref_masks.shape) == (ref_images.shape[0], H, W, 1)):
        raise ValueError(f"image_head: ref_images [L,{H},{W},3] / ref_masks [L,{H},{W},1] must be float32 on {color.device}, got "
                         f"{tuple(ref_images.shape)} {ref_images.dtype} {ref_images.device} / {tuple(ref_masks.shape)} {ref_masks.dtype}")
    if alpha.device != color.device or any(t.device != color.device for t in (ref_pos, rnd_pos, fidx_ref)):
        raise ValueError("image_head: every tensor on the colour image's device")
    return _ImageHead.apply(color, alpha, ref_pos, rnd_pos, ref_images, ref_masks, fidx_ref, int(n_ref), int(n_rnd))
