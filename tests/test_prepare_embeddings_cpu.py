"""CPU: the set-up half of the Zero123 guidance (dreammesh4d_amd/clip_vit.py) -- ``prepare_embeddings`` /
``prepare_embeddings_video`` of custom/threestudio-dreammesh4d/guidance/temporal_stable_zero123_guidance.py:174-226 -- and the
guidance plugins constructed from the shipped configuration's OWN keys (cond_video_dir / cond_image_path + the checkpoint), with no
``cond_embeddings_path``.  PARITY UNPINNED for the CLIP tower and the cv2 resize (`clip`, `kornia`, `cv2` are absent from the image):
the checks are against the definitions (explicit attention, block means, the OpenAI checkpoint's key layout)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dreammesh4d_amd import clip_vit


def test_tower_has_the_openai_checkpoint_keys_and_loads_from_a_prefixed_state_dict():
    torch.manual_seed(0)
    t = clip_vit.CLIPVisionTower(input_resolution=224, patch_size=14, width=128, layers=3, heads=2, output_dim=48)
    keys = set(t.state_dict())
    want = {"conv1.weight", "class_embedding", "positional_embedding", "ln_pre.weight", "ln_pre.bias", "ln_post.weight", "ln_post.bias", "proj"}
    for i in range(3):
        want |= {f"transformer.resblocks.{i}.{k}" for k in (
            "attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight", "attn.out_proj.bias", "ln_1.weight", "ln_1.bias",
            "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias", "ln_2.weight", "ln_2.bias")}
    assert keys == want
    assert tuple(t.positional_embedding.shape) == (257, 128) and tuple(t.proj.shape) == (128, 48)
    full = clip_vit.CLIPVisionTower()                                            # ViT-L/14, meta tensors: shapes only
    assert tuple(full.conv1.weight.shape) == (1024, 3, 14, 14) and len(full.transformer.resblocks) == 24 and tuple(full.proj.shape) == (1024, 768)
    sd = {"cond_stage_model.model.visual." + k: v for k, v in t.state_dict().items()}
    sd["model.diffusion_model.something"] = torch.zeros(1)
    t2 = clip_vit.CLIPVisionTower.from_state_dict(sd)
    x = torch.rand(2, 3, 224, 224) * 2 - 1
    assert torch.equal(t(x), t2(x))
    with pytest.raises(KeyError):
        clip_vit.CLIPVisionTower.from_state_dict({"model.x": torch.zeros(1)})


def test_tower_forward_equals_the_written_out_transformer():
    """clip.model.VisionTransformer.forward restated with explicit softmax attention and QuickGELU."""
    torch.manual_seed(1)
    W, Hd, L = 64, 2, 2
    t = clip_vit.CLIPVisionTower(input_resolution=28, patch_size=14, width=W, layers=L, heads=Hd, output_dim=16).eval()
    x = torch.randn(3, 3, 28, 28)
    with torch.no_grad():
        got = t(x)
        h = F.conv2d(x, t.conv1.weight, stride=14).flatten(2).permute(0, 2, 1)                 # [B, 4, W]
        h = torch.cat([t.class_embedding.expand(3, 1, W), h], 1) + t.positional_embedding
        h = F.layer_norm(h, (W,), t.ln_pre.weight, t.ln_pre.bias)
        for blk in t.transformer.resblocks:
            y = F.layer_norm(h, (W,), blk.ln_1.weight, blk.ln_1.bias)
            qkv = y @ blk.attn.in_proj_weight.T + blk.attn.in_proj_bias
            q, k, v = [z.reshape(3, 5, Hd, W // Hd).transpose(1, 2) for z in qkv.chunk(3, -1)]
            a = torch.softmax(q @ k.transpose(-1, -2) / (W // Hd) ** 0.5, -1) @ v
            h = h + a.transpose(1, 2).reshape(3, 5, W) @ blk.attn.out_proj.weight.T + blk.attn.out_proj.bias
            y = F.layer_norm(h, (W,), blk.ln_2.weight, blk.ln_2.bias)
            y = y @ blk.mlp.c_fc.weight.T + blk.mlp.c_fc.bias
            y = y * torch.sigmoid(1.702 * y)
            h = h + y @ blk.mlp.c_proj.weight.T + blk.mlp.c_proj.bias
        want = F.layer_norm(h[:, 0], (W,), t.ln_post.weight, t.ln_post.bias) @ t.proj
    assert torch.allclose(got, want, atol=2e-5, rtol=1e-5), float((got - want).abs().max())
    # float16 storage with float32 LayerNorms (guidance :117-134) stays close to the float32 result
    t16 = clip_vit.CLIPVisionTower(input_resolution=28, patch_size=14, width=W, layers=L, heads=Hd, output_dim=16)
    t16.load_state_dict(t.state_dict())
    try:
        with torch.no_grad():
            g16 = t16.to(torch.bfloat16)(x).float()
    except RuntimeError:
        return
    assert float((g16 - want).abs().max()) < 0.15 * float(want.abs().max())


def test_preprocess_is_bicubic_224_then_clip_normalisation():
    x = torch.rand(2, 3, 256, 256) * 2 - 1
    y = clip_vit.clip_preprocess(x)
    assert tuple(y.shape) == (2, 3, 224, 224)
    # align_corners=True: the four corners are the input's corners
    for c in range(3):
        for (i, j), (a, b) in {(0, 0): (0, 0), (223, 223): (255, 255), (0, 223): (0, 255)}.items():
            want = ((x[0, c, a, b] + 1) / 2 - clip_vit.CLIP_MEAN[c]) / clip_vit.CLIP_STD[c]
            assert abs(float(y[0, c, i, j]) - float(want)) < 1e-4
    const = clip_vit.clip_preprocess(torch.zeros(1, 3, 256, 256))                        # 0 in [-1, 1] = mid grey
    assert torch.allclose(const[0, :, 10, 10], (0.5 - torch.tensor(clip_vit.CLIP_MEAN)) / torch.tensor(clip_vit.CLIP_STD), atol=1e-6)


def test_area_resize():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (512, 512, 4), dtype=np.uint8)
    r = clip_vit.resize_area_u8(a, 256, 256)
    blk = a.reshape(256, 2, 256, 2, 4).astype(np.int64).sum((1, 3))
    assert np.array_equal(r, ((blk + 2) // 4).astype(np.uint8))                          # the integer-ratio path: block means, round half up
    b = rng.integers(0, 256, (300, 420, 3), dtype=np.uint8)
    r = clip_vit.resize_area_u8(b, 256, 256)
    assert r.shape == (256, 256, 3) and abs(float(r.mean()) - float(b.mean())) < 0.5     # area weights preserve the mean
    c = np.full((300, 420, 3), 77, np.uint8)
    assert np.all(clip_vit.resize_area_u8(c, 256, 256) == 77)
    assert clip_vit.resize_area_u8(np.full((128, 128, 3), 9, np.uint8), 256, 256).shape == (256, 256, 3)


def _write_frames(d, n, size=320):
    from PIL import Image

    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(3)
    for i in range(n):
        rgba = rng.integers(0, 256, (size, size, 4), dtype=np.uint8)
        rgba[: size // 4, :, 3] = 0                                                     # a transparent band: composited on white
        Image.fromarray(rgba, "RGBA").save(os.path.join(d, f"{i:03}_rgba.png" if i % 2 == 0 else f"{i}.png"))


def test_rgb_256_is_composited_on_white(tmp_path):
    _write_frames(str(tmp_path), 2, size=512)
    x = clip_vit.load_rgba_256(clip_vit.video_frame_path(str(tmp_path), 0))
    assert tuple(x.shape) == (1, 3, 256, 256) and x.dtype == torch.float32 and 0.0 <= float(x.min()) and float(x.max()) <= 1.0
    assert torch.all(x[:, :, :64] == 1.0)                                                # alpha 0 -> white
    assert clip_vit.video_frame_path(str(tmp_path), 1).endswith("1.png")                 # guidance :201-204: the second naming scheme


def _tiny_checkpoint(tmp_path):
    """A random-weight checkpoint + model YAML of the shapes `_zero123_from_config` reads (load/zero123/*.yaml's structure)."""
    import yaml

    from dreammesh4d_amd import zero123 as z

    torch.manual_seed(0)
    uk = dict(in_channels=8, out_channels=4, model_channels=32, attention_resolutions=(4, 2, 1), num_res_blocks=1, channel_mult=(1, 2), num_heads=4,
              context_dim=32)
    vk = dict(ch=32, ch_mult=(1, 2, 4, 4), num_res_blocks=1, in_channels=3, z_channels=4)
    model = z.Zero123(unet_kwargs=uk, vae_kwargs=vk)
    tower = clip_vit.CLIPVisionTower(width=64, layers=2, heads=1, output_dim=32)
    sd = dict(model.state_dict())
    sd.update({"cond_stage_model.model.visual." + k: v for k, v in tower.state_dict().items()})
    ckpt = str(tmp_path / "zero123.ckpt")
    torch.save({"state_dict": sd}, ckpt)
    conf = {"model": {"params": {"timesteps": 1000, "linear_start": 0.00085, "linear_end": 0.0120, "scale_factor": 0.18215,
                                 "unet_config": {"params": {k: (list(v) if isinstance(v, tuple) else v) for k, v in uk.items()}},
                                 "first_stage_config": {"params": {"ddconfig": {k: (list(v) if isinstance(v, tuple) else v) for k, v in vk.items()}}}}}}
    yml = str(tmp_path / "model.yaml")
    with open(yml, "w") as fh:
        yaml.safe_dump(conf, fh)
    return ckpt, yml, model, tower


def test_guidance_plugins_construct_from_the_shipped_keys_without_extra_ones(tmp_path):
    """`find("temporal-stable-zero123-guidance")(cfg)` with the keys of configs/sugar_dynamic_dg.yaml:100-117 only: cond_video_dir holds
    the frames, the checkpoint holds UNet + VAE encoder + cc_projection + the CLIP tower; likewise `stable-zero123-guidance` with
    cond_image_path (configs/sugar_static_refine.yaml:91-103).  The embeddings equal prepare_embeddings() called directly."""
    from dreammesh4d_amd import threestudio_host as ts

    ckpt, yml, model, tower = _tiny_checkpoint(tmp_path)
    vdir = str(tmp_path / "video")
    _write_frames(vdir, 3)
    cfg = {"num_frames": 3, "pretrained_config": yml, "pretrained_model_name_or_path": ckpt, "vram_O": True, "cond_video_dir": vdir,
           "cond_elevation_deg": 5.0, "cond_azimuth_deg": 0.0, "cond_camera_distance": 3.8, "guidance_scale": 3.0, "min_step_percent": 0.02,
           "max_step_percent": 0.5, "chunk_size": None, "half_precision_weights": False}
    g = ts.find("temporal-stable-zero123-guidance")(cfg)
    assert tuple(g.c_crossattn.shape) == (3, 1, 32) and tuple(g.c_concat.shape) == (3, 4, 32, 32)
    assert not hasattr(g.model, "_clip_visual_sd")                                      # the tower's tensors are dropped after set-up
    paths = [clip_vit.video_frame_path(vdir, i) for i in range(3)]
    rgb, cc, ct = clip_vit.prepare_embeddings(model, tower.eval(), paths, torch.device("cpu"), torch.float32)
    assert tuple(rgb.shape) == (3, 3, 256, 256)
    assert torch.allclose(g.c_crossattn.float().cpu(), cc, atol=1e-5) and torch.allclose(g.c_concat.float().cpu(), ct, atol=1e-5)
    # c_concat is the posterior MEAN, unscaled (guidance :221: encode_first_stage(img).mode())
    mom = model.first_stage_model.encode_moments(rgb[:1] * 2 - 1)
    assert torch.allclose(ct[:1], mom[:, :4], atol=1e-6)
    scfg = {k: v for k, v in cfg.items() if k not in ("num_frames", "cond_video_dir", "chunk_size")}
    scfg["cond_image_path"] = paths[0]
    s = ts.find("stable-zero123-guidance")(scfg)
    assert tuple(s.c_crossattn.shape) == (1, 1, 32) and torch.allclose(s.c_crossattn.float().cpu(), cc[:1], atol=1e-5)
    # a missing frame is reported by name, not as a CLIP error
    with pytest.raises(FileNotFoundError):
        ts.find("temporal-stable-zero123-guidance")(dict(cfg, num_frames=4))
