"""CPU tests of the Zero123 mirror (dreammesh4d_amd/zero123.py): architecture parity with the
reference's UNetModel / Encoder through tests/golden/zero123_small.npz (reduced width, name-seeded
weights), checkpoint key layout, and the SDS arithmetic of the guidance step."""
import os

import numpy as np
import torch

from dreammesh4d_amd import zero123 as z

G = os.path.join(os.path.dirname(__file__), "golden")


def seeded_fill(module, base, scale=0.05):
    """Must equal tests/golden/make_golden.py::seeded_fill."""
    sd = module.state_dict()
    with torch.no_grad():
        for i, k in enumerate(sorted(sd.keys())):
            t = sd[k]
            if not t.dtype.is_floating_point:
                continue
            g = torch.Generator().manual_seed(base + i)
            v = torch.randn(t.shape, generator=g) * scale
            if ("norm" in k or k.endswith("in_layers.0.weight") or k.endswith("out_layers.0.weight") or k == "out.0.weight") and k.endswith("weight") and t.dim() == 1:
                v = v + 1.0
            t.copy_(v)
    return sorted(sd.keys())


def test_unet_matches_reference_at_reduced_width():
    g = np.load(os.path.join(G, "zero123_small.npz"))
    unet = z.UNetModel(in_channels=8, out_channels=4, model_channels=32, attention_resolutions=(4, 2, 1), num_res_blocks=2,
                       channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=48).eval()
    keys = seeded_fill(unet, base=1000)
    assert keys == g["unet_keys"].tolist()                         # identical state-dict layout
    assert sum(p.numel() for p in unet.parameters()) == int(g["unet_params"])
    with torch.no_grad():
        y = unet(torch.tensor(g["x"]), torch.tensor(g["t"]), torch.tensor(g["ctx"]))
    assert np.abs(y.numpy() - g["y"]).max() < 2e-5


def test_vae_encoder_matches_reference_at_reduced_width():
    g = np.load(os.path.join(G, "zero123_small.npz"))
    enc = z.VaeEncoder(ch=32, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, z_channels=4).eval()
    keys = seeded_fill(enc, base=5000)
    assert keys == g["enc_keys"].tolist()
    with torch.no_grad():
        m = enc(torch.tensor(g["img"]))
    assert np.abs(m.numpy() - g["moments"]).max() < 2e-5


def test_full_size_layout_of_the_ldm_checkpoint():
    with torch.device("meta"):
        net = z.Zero123()
    keys = list(net.state_dict().keys())
    n_unet = sum(v.numel() for k, v in net.state_dict().items() if k.startswith("model.diffusion_model."))
    assert n_unet == 859_532_484 + 0 or abs(n_unet - 859.5e6) < 1e6      # 859.5 M (in 8 / out 4 channels)
    for k in ("model.diffusion_model.input_blocks.1.1.transformer_blocks.0.attn2.to_k.weight",
              "model.diffusion_model.middle_block.1.proj_out.weight", "model.diffusion_model.out.2.bias",
              "first_stage_model.encoder.down.2.downsample.conv.weight", "first_stage_model.encoder.mid.attn_1.q.weight",
              "first_stage_model.quant_conv.weight", "cc_projection.weight"):
        assert k in keys, k
    assert net.state_dict()["cc_projection.weight"].shape == (768, 772)


def _tiny(grad_clip=None):
    torch.manual_seed(0)
    model = z.Zero123(unet_kwargs=dict(model_channels=32, context_dim=32, num_heads=4),
                      vae_kwargs=dict(ch=32))
    for p in model.model.diffusion_model.out.parameters():        # un-zero the output conv so eps_hat is not 0
        torch.nn.init.normal_(p, std=0.05)
    L = 5
    return z.TemporalStableZero123Guidance(model, torch.randn(L, 1, 32), torch.randn(L, 4, 32, 32), guidance_scale=3.0,
                                           half_precision_weights=False, grad_clip=grad_clip)


def test_sds_loss_gradient_is_the_sds_gradient():
    guid = _tiny()
    B = 2
    lat = (torch.rand(B, 32, 32, 4) * 0.5 + 0.25).requires_grad_(True)      # rgb_as_latents path: no VAE
    el, az, cd = torch.tensor([10.0, 40.0]), torch.tensor([-30.0, 120.0]), torch.full((B,), 3.8)
    noise, t = torch.randn(B, 4, 32, 32), torch.tensor([100, 400])
    out = guid(lat, el, az, cd, frame_indices=torch.tensor([1, 3]), rgb_as_latents=True, noise=noise, t=t)
    out["loss_sds"].backward()
    # recompute the SDS gradient by hand (guidance.py:338-365)
    with torch.no_grad():
        latents = lat.detach().permute(0, 3, 1, 2) * 2 - 1
        ac = guid.model.alphas_cumprod[t].view(-1, 1, 1, 1)
        noisy = ac.sqrt() * latents + (1 - ac).sqrt() * noise
        cond = guid.get_cond(el, az, cd, torch.tensor([1, 3]))
        pred = guid.model.apply_model(torch.cat([noisy] * 2), torch.cat([t] * 2), cond)
        unc, cnd = pred.chunk(2)
        grad = (1 - ac) * (unc + 3.0 * (cnd - unc) - noise)
    got = lat.grad.permute(0, 3, 1, 2) / 2.0 * B          # d latents / d rgb = 2 ; loss divides by B
    assert torch.allclose(got, grad, atol=1e-5)
    assert abs(out["grad_norm"].item() - grad.norm().item()) < 1e-3
    assert out["min_step"] == 20 and out["max_step"] == 980


def test_cond_layout_schedule_and_full_image_path():
    guid = _tiny(grad_clip=0.5)
    el, az, cd = torch.tensor([5.0]), torch.tensor([90.0]), torch.tensor([3.8])
    cond = guid.get_cond(el, az, cd, torch.tensor([2]))
    cc, cat = cond["c_crossattn"][0], cond["c_concat"][0]
    assert cc.shape == (2, 1, 32) and cat.shape == (2, 4, 32, 32)
    assert not cc[0].any() and not cat[0].any() and torch.equal(cat[1], guid.c_concat[2])     # uncond = zeros
    ac = guid.model.alphas_cumprod
    betas = torch.linspace(0.00085 ** 0.5, 0.0120 ** 0.5, 1000, dtype=torch.float64) ** 2
    assert torch.allclose(ac.double(), torch.cumprod(1 - betas, 0), atol=1e-6) and ac.shape == (1000,)
    rgb = torch.rand(1, 64, 64, 3, requires_grad=True)                 # goes through interpolate + the VAE encoder
    out = guid(rgb, el, az, cd, frame_indices=torch.tensor([0]))
    out["loss_sds"].backward()
    assert torch.isfinite(rgb.grad).all() and rgb.grad.abs().sum() > 0
    guid.update_step(0, 10, min_step_percent=0.02, max_step_percent=0.5)
    assert guid.max_step == 500


def test_single_token_cross_attention_in_closed_form():
    """Zero123's context is one token per sample: softmax over a single key is exactly 1, so cross-attention is
    to_out(to_v(context)) broadcast over the positions (zero123.CrossAttention.single_token).  Same numbers as the general
    path (query projection, attention, output GEMM over all positions); longer contexts take the general path."""
    import torch

    from dreammesh4d_amd import zero123 as z

    torch.manual_seed(0)
    unet = z.UNetModel(model_channels=32, context_dim=24, num_heads=4)
    with torch.no_grad():
        for m in unet.modules():
            if isinstance(m, z.SpatialTransformer):
                torch.nn.init.normal_(m.proj_out.weight, std=0.1)
            if isinstance(m, z.ResBlock):
                torch.nn.init.normal_(m.out_layers[3].weight, std=0.05)
        torch.nn.init.normal_(unet.out[2].weight, std=0.1)
    x, t = torch.randn(2, 8, 16, 16), torch.tensor([100, 700])
    ctx1, ctx3 = torch.randn(2, 1, 24), torch.randn(2, 3, 24)
    try:
        with torch.no_grad():
            z.SINGLE_TOKEN_SHORTCUT = True
            a1, a3 = unet(x, t, ctx1), unet(x, t, ctx3)
            z.SINGLE_TOKEN_SHORTCUT = False
            b1, b3 = unet(x, t, ctx1), unet(x, t, ctx3)
    finally:
        z.SINGLE_TOKEN_SHORTCUT = True
    assert float(a1.abs().max()) > 1e-3
    assert float((a1 - b1).abs().max()) <= 2e-6 * float(b1.abs().max())
    assert torch.equal(a3, b3)
    att = [m for m in unet.modules() if isinstance(m, z.CrossAttention)][1]          # an attn2
    q = torch.randn(2, 5, att.to_q.in_features)
    c = torch.randn(2, 1, att.to_k.in_features)
    z.SINGLE_TOKEN_SHORTCUT = False
    try:
        general = att(q, c)
    finally:
        z.SINGLE_TOKEN_SHORTCUT = True
    assert torch.allclose(att(q, c), general, rtol=1e-6, atol=1e-7) and att(q, c).shape == (2, 5, att.to_q.in_features)


def test_batched_small_gemms_hand_every_block_its_own_projection():
    """UNetModel._batched_small_gemms: ONE GEMM for the timestep projections of all ResBlocks (emb_layers(emb) + the first
    convolution's bias), ONE for the value projections of all single-token cross-attentions and one batched GEMM per width for
    their output projections; every block must get exactly what its own layers would compute (openaimodel.py:259-266,
    attention.py:196-213).  Host logic, run on the CPU."""
    import torch

    from dreammesh4d_amd import zero123 as z

    torch.manual_seed(1)
    unet = z.UNetModel(model_channels=32, context_dim=24, num_heads=4)
    with torch.no_grad():
        for m in unet.modules():
            if isinstance(m, z.ResBlock):
                torch.nn.init.normal_(m.in_layers[2].bias, std=0.3)
    emb, ctx = torch.randn(3, unet.time_embed[2].out_features), torch.randn(3, 1, 24)
    with torch.no_grad():
        res, att = unet._batched_small_gemms(emb, ctx)
        try:
            assert len(res) == sum(isinstance(m, z.ResBlock) for m in unet.modules()) and len(att) == sum(isinstance(m, z.BasicTransformerBlock) for m in unet.modules())
            for b in res:
                want = b.emb_layers(emb) + b.in_layers[2].bias
                got = b.__dict__["_emb_add"]
                assert got.shape == want.shape and float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
            for a in att:
                want = a.to_out(a.to_v(ctx))
                got = a.__dict__["_tok"]
                assert got.shape == (3, want.shape[-1]) and float((got[:, None] - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-7
                assert a.single_token(ctx).shape == want.shape and float((a.single_token(ctx) - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-7
        finally:
            for b in res:
                b.__dict__.pop("_emb_add", None)
            for a in att:
                a.__dict__.pop("_tok", None)
    # the cache follows the parameters: an in-place update of a weight is picked up
    with torch.no_grad():
        res[0].emb_layers[1].weight.mul_(2.0)
        unet._batched_small_gemms(emb, ctx)
        assert float((res[0].__dict__["_emb_add"] - (res[0].emb_layers(emb) + res[0].in_layers[2].bias)).abs().max()) < 1e-4
        att[0].to_out[0].weight.mul_(0.5)
        unet._batched_small_gemms(emb, ctx)
        assert float((att[0].__dict__["_tok"][:, None] - att[0].to_out(att[0].to_v(ctx))).abs().max()) < 1e-4
        for b in res:
            b.__dict__.pop("_emb_add", None)
        for a in att:
            a.__dict__.pop("_tok", None)
