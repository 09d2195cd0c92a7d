"""The set-up half of the Zero123 guidance: ``prepare_embeddings`` / ``prepare_embeddings_video``
(custom/threestudio-dreammesh4d/guidance/temporal_stable_zero123_guidance.py:174-226) without the packages the image lacks.

What the reference does once, at ``configure`` time, per conditioning frame:

    rgba = cv2.resize(cv2.imread(path, IMREAD_UNCHANGED) -> RGBA, (256, 256), INTER_AREA) / 255          (:175-186)
    rgb  = rgba[..., :3] * a + (1 - a)                                                                   (:187)
    c_crossattn = model.get_learned_conditioning(2 rgb - 1)  = FrozenCLIPImageEmbedder.encode(.)          (:220)
                = clip ViT-L/14 ``encode_image(preprocess(x))[:, None]``   (extern/ldm_zero123/modules/encoders/modules.py:432-477:
                  bicubic resize to 224 x 224 with align_corners=True, (x + 1) / 2, CLIP mean / std)
    c_concat    = model.encode_first_stage(2 rgb - 1).mode()               (:221: the VAE posterior's MEAN, not scaled)

``clip`` (OpenAI), ``kornia`` and ``cv2`` are un-vendored dependencies (requirements.txt) absent from this image: the tower below is a
plain restatement of ``clip.model.VisionTransformer`` with the SAME state-dict keys, so the weights the Zero123 checkpoint carries under
``cond_stage_model.model.visual.*`` load into it unchanged; the resize is ``F.interpolate(mode="bicubic", align_corners=True)`` (what
kornia's ``resize`` calls for a (h, w) size without antialias), the image file is read with PIL and area-resized here.

PARITY UNPINNED: none of the three packages is available to generate golden vectors, so this module is checked against its own
definitions only (attention against an explicit softmax, the area resize against block means, keys against the documented layout of
the OpenAI checkpoint).  It runs ONCE per conditioning frame at set-up time -- plain torch operators, not part of the hot path;
``cond_embeddings_path`` (threestudio_host) stays as the override for embeddings computed elsewhere.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class LayerNormF32(nn.LayerNorm):
    """clip.model.LayerNorm: computed in float32 whatever the input's dtype (the guidance keeps these modules' parameters in
    float32 when the rest of the model is cast to float16, guidance :117-134)."""

    def forward(self, x):
        return F.layer_norm(x.float(), self.normalized_shape, None if self.weight is None else self.weight.float(),
                            None if self.bias is None else self.bias.float(), self.eps).to(x.dtype)


class QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class _Mlp(nn.Sequential):
    def __init__(self, d):
        super().__init__()
        self.c_fc = nn.Linear(d, 4 * d)
        self.gelu = QuickGELU()
        self.c_proj = nn.Linear(4 * d, d)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.attn = nn.MultiheadAttention(d, heads)          # (keys: attn.in_proj_weight / in_proj_bias / out_proj.*)
        self.ln_1 = LayerNormF32(d)
        self.mlp = _Mlp(d)
        self.ln_2 = LayerNormF32(d)

    def forward(self, x):                                     # x: [L, B, d]
        y = self.ln_1(x)
        x = x + self.attn(y, y, y, need_weights=False)[0]
        return x + self.mlp(self.ln_2(x))


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


class CLIPVisionTower(nn.Module):
    """``clip.model.VisionTransformer`` (ViT-L/14 by default: 224 x 224 input, 14 x 14 patches, width 1024, 24 layers, 16 heads,
    768-d output) with its state-dict keys: conv1.weight, class_embedding, positional_embedding, ln_pre.*, transformer.resblocks.N.*,
    ln_post.*, proj."""

    def __init__(self, input_resolution=224, patch_size=14, width=1024, layers=24, heads=16, output_dim=768):
        super().__init__()
        self.input_resolution, self.output_dim = input_resolution, output_dim
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNormF32(width)
        self.transformer = _Transformer(width, layers, heads)
        self.ln_post = LayerNormF32(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))

    @property
    def dtype(self):
        return self.conv1.weight.dtype

    def forward(self, x):
        x = self.conv1(x.to(self.dtype))                       # [B, width, g, g]
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
        cls = self.class_embedding.to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype, device=x.device)
        x = torch.cat([cls, x], dim=1) + self.positional_embedding.to(x.dtype)
        x = self.ln_pre(x)
        x = self.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
        x = self.ln_post(x[:, 0, :])
        return x @ self.proj

    @classmethod
    def from_state_dict(cls, sd, prefix="cond_stage_model.model.visual."):
        """The tower sized from, and loaded with, the entries of a checkpoint's ``state_dict`` under ``prefix`` (strict)."""
        own = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        if "conv1.weight" not in own:
            raise KeyError(f"the checkpoint has no `{prefix}conv1.weight`: its CLIP image encoder is missing (pass the conditioning "
                           "embeddings as cond_embeddings_path instead)")
        width, patch = int(own["conv1.weight"].shape[0]), int(own["conv1.weight"].shape[-1])
        layers = 1 + max(int(k.split(".")[2]) for k in own if k.startswith("transformer.resblocks."))
        grid = int(round((own["positional_embedding"].shape[0] - 1) ** 0.5))
        m = cls(input_resolution=grid * patch, patch_size=patch, width=width, layers=layers, heads=width // 64,
                output_dim=int(own["proj"].shape[1]))
        m.load_state_dict(own, strict=True)
        return m


def clip_preprocess(x):
    """FrozenCLIPImageEmbedder.preprocess (modules.py:457-469): x in [-1, 1] -> bicubic 224 x 224 (align_corners=True, no antialias),
    back to [0, 1], CLIP mean / std."""
    x = F.interpolate(x.float(), size=(224, 224), mode="bicubic", align_corners=True)
    x = (x + 1.0) / 2.0
    mean = torch.tensor(CLIP_MEAN, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, device=x.device).view(1, 3, 1, 1)
    return (x - mean) / std


@torch.no_grad()
def clip_image_embedding(tower: CLIPVisionTower, img_m11):
    """``FrozenCLIPImageEmbedder.encode`` (modules.py:471-480): [B,3,H,W] in [-1, 1] -> [B,1,768] float32."""
    return tower(clip_preprocess(img_m11)).float().unsqueeze(1)


# ------------------------------------------------------------------------------------------------ image file -> rgb_256
def resize_area_u8(img, out_h, out_w):
    """``cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_AREA)`` for uint8 [H,W,C] when shrinking: every output pixel is the
    area-weighted mean of the input pixels its footprint covers, rounded to uint8 (integer ratios: exact block means rounded half up,
    as OpenCV's integer fast path; other ratios: fractional coverage weights, rounded to nearest).  Enlarging falls back to bilinear,
    as OpenCV's INTER_AREA does."""
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim != 3:
        raise ValueError("resize_area_u8: uint8 [H,W,C] image expected")
    H, W, C = img.shape
    if out_h > H or out_w > W:
        t = torch.from_numpy(img).permute(2, 0, 1)[None].float()
        o = F.interpolate(t, size=(out_h, out_w), mode="bilinear", align_corners=False)
        return o[0].permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).numpy()
    if H % out_h == 0 and W % out_w == 0:
        fy, fx = H // out_h, W // out_w
        s = img.reshape(out_h, fy, out_w, fx, C).astype(np.int64).sum(axis=(1, 3))
        return ((s + (fy * fx) // 2) // (fy * fx)).astype(np.uint8)

    def weights(n_in, n_out):
        scale = n_in / n_out
        Wm = np.zeros((n_out, n_in), np.float64)
        for o in range(n_out):
            lo, hi = o * scale, (o + 1) * scale
            for i in range(int(np.floor(lo)), min(int(np.ceil(hi)), n_in)):
                Wm[o, i] = max(0.0, min(hi, i + 1) - max(lo, i)) / scale
        return Wm

    wy, wx = weights(H, out_h), weights(W, out_w)
    o = np.einsum("oh,hwc->owc", wy, img.astype(np.float64))
    o = np.einsum("pw,owc->opc", wx, o)
    return np.clip(np.rint(o), 0, 255).astype(np.uint8)


def load_rgba_256(path):
    """guidance :175-195: the RGBA file -> ``rgb_256`` [1,3,256,256] float32 in [0, 1], composited on white."""
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    from PIL import Image

    im = Image.open(path)
    rgba = np.asarray(im.convert("RGBA"), dtype=np.uint8)
    rgba = resize_area_u8(rgba, 256, 256).astype(np.float32) / 255.0
    rgb = rgba[..., :3] * rgba[..., 3:] + (1.0 - rgba[..., 3:])
    return torch.from_numpy(rgb).unsqueeze(0).permute(0, 3, 1, 2).contiguous()


def video_frame_path(video_dir, i):
    """guidance :201-204: ``{i:03}_rgba.png``, else ``{i}.png``."""
    p = os.path.join(video_dir, f"{i:03}_rgba.png")
    return p if os.path.exists(p) else os.path.join(video_dir, f"{i}.png")


@torch.no_grad()
def prepare_embeddings(model, tower, image_paths, device, weights_dtype=torch.float16):
    """``prepare_embeddings_video`` (guidance :197-212) / ``prepare_embeddings`` (:174-195) over ``image_paths``:
    -> (rgb_256 [n,3,256,256] float32, c_crossattn [n,1,768], c_concat [n,4,32,32]) in ``weights_dtype`` like ``get_img_embeds`` (:214-222).
    ``model``: a ``zero123.Zero123`` (its VAE encoder; used in the dtype it is in); ``tower``: the CLIP image tower."""
    rgb, cc, ct = [], [], []
    p = next(model.first_stage_model.parameters())
    for path in image_paths:
        x = load_rgba_256(path).to(device)
        img = (x * 2.0 - 1.0)
        cc.append(clip_image_embedding(tower, img.to(weights_dtype)).to(weights_dtype))
        moments = model.first_stage_model.encode_moments(img.to(p.dtype))
        ct.append(moments.chunk(2, dim=1)[0].to(weights_dtype))                  # DiagonalGaussianDistribution.mode() = the mean
        rgb.append(x)
    return torch.cat(rgb, 0), torch.cat(cc, 0), torch.cat(ct, 0)
