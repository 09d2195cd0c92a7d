"""-m gpu: the NHWC GroupNorm (+ SiLU, + per-(sample, channel) add) operator of the Zero123 step (csrc/groupnorm.hip through
dreammesh4d_amd/fused_norm.py) against plain PyTorch float32 of the same op -- forward and dL/dx -- for the channel counts of
the UNet (320 ... 2560: 10 ... 80 channels per group) and of the VAE encoder (128 ... 512), ragged slabs, float16 and float32
storage; and the guidance step with NHWC activations against the NCHW library path."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(2, 320, 32, 32), (3, 2560, 8, 8), (2, 128, 64, 64), (1, 960, 16, 16), (2, 64, 5, 7), (2, 1920, 16, 16), (1, 512, 32, 32),
          (2, 32, 3, 3), (1, 1280, 1, 1)]


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")


def _reference(x, add, w, b, G, eps, silu):
    z = x.float() if add is None else x.float() + add.float()[:, :, None, None]
    y = F.group_norm(z, G, w.float(), b.float(), eps)
    return F.silu(y) if silu else y


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("shape", SHAPES)
def test_forward_and_input_gradient_against_float32_torch(shape, dtype):
    _need_gpu()
    from dreammesh4d_amd import fused_norm

    dev = torch.device("cuda:0")
    N, C, H, W = shape
    G = 32
    g = torch.Generator(device="cpu").manual_seed(N * 1000 + C + H)
    x = (torch.randn(shape, generator=g) * 2.0 + 0.5).to(dev, dtype).contiguous(memory_format=torch.channels_last)
    add = torch.randn(N, C, generator=g).to(dev, dtype)
    m = torch.nn.GroupNorm(G, C, eps=1e-5).to(dev, dtype)
    with torch.no_grad():
        m.weight.copy_(1.0 + 0.3 * torch.randn(C, generator=g))
        m.bias.copy_(0.2 * torch.randn(C, generator=g))
    for p in m.parameters():
        p.requires_grad_(False)
    tol = dict(rtol=2e-3, atol=2e-3) if dtype == torch.float16 else dict(rtol=2e-5, atol=2e-5)
    for silu in (False, True):
        for a in (None, add):
            assert fused_norm.fused_ok(m, x)
            with torch.no_grad():
                y = fused_norm.group_norm(m, x, silu=silu, add=a)
                y2 = fused_norm.group_norm(m, x, silu=silu, add=a)
            assert y.shape == x.shape and y.dtype == dtype and fused_norm.is_channels_last(y)
            assert torch.equal(y, y2)                                         # no atomics: reproducible
            want = _reference(x, a, m.weight, m.bias, G, m.eps, silu)
            torch.testing.assert_close(y.float(), want, **tol)
        # dL/dx (frozen gamma / beta)
        xg = x.clone().requires_grad_(True)
        dy = torch.randn(shape, generator=g).to(dev, dtype).contiguous(memory_format=torch.channels_last)
        y = fused_norm.group_norm(m, xg, silu=silu)
        y.backward(dy)
        xr = x.float().clone().requires_grad_(True)
        _reference(xr, None, m.weight, m.bias, G, m.eps, silu).backward(dy.float())
        scale = float(xr.grad.abs().max())
        gtol = (4e-3 if dtype == torch.float16 else 3e-5) * max(scale, 1e-3)
        assert float((xg.grad.float() - xr.grad).abs().max()) <= gtol, (silu, float((xg.grad.float() - xr.grad).abs().max()), scale)
        assert fused_norm.is_channels_last(xg.grad)
    # a per-channel constant in front of the norm (a convolution's bias folded in), with dL/dx
    cb = (0.5 * torch.randn(C, generator=g)).to(dev, dtype)
    xg = x.clone().requires_grad_(True)
    dy = torch.randn(shape, generator=g).to(dev, dtype).contiguous(memory_format=torch.channels_last)
    y = fused_norm.group_norm(m, xg, silu=True, add=cb)
    y.backward(dy)
    xr = x.float().clone().requires_grad_(True)
    yr = F.silu(F.group_norm(xr + cb.float().view(1, C, 1, 1), G, m.weight.float(), m.bias.float(), m.eps))
    yr.backward(dy.float())
    torch.testing.assert_close(y.float(), yr, **tol)
    scale = float(xr.grad.abs().max())
    assert float((xg.grad.float() - xr.grad).abs().max()) <= (4e-3 if dtype == torch.float16 else 3e-5) * max(scale, 1e-3)
    # an NCHW gradient arriving at a channels-last operator
    xg = x.clone().requires_grad_(True)
    y = fused_norm.group_norm(m, xg, silu=True)
    (y.contiguous() * 2.0).sum().backward()
    assert torch.isfinite(xg.grad).all()


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_gradient_of_the_branch_around_the_norm_is_added_in_the_backward_kernel(dtype):
    """group_norm(..., skip=True) -> (y, x'): out = f(y) + g(x') as a ResnetBlock uses it (x + h(norm(x))).  dL/dx must be the norm's
    input gradient PLUS the branch's, from one backward launch pair (no accumulation launch), also when only one of the two
    outputs is used."""
    _need_gpu()
    from dreammesh4d_amd import fused_norm

    dev = torch.device("cuda:0")
    for shape in ((2, 128, 64, 64), (3, 512, 9, 7), (1, 320, 32, 32)):
        N, C, H, W = shape
        g = torch.Generator(device="cpu").manual_seed(C + H)
        x = (torch.randn(shape, generator=g) * 1.5 - 0.3).to(dev, dtype).contiguous(memory_format=torch.channels_last)
        m = torch.nn.GroupNorm(32, C, eps=1e-6).to(dev, dtype).requires_grad_(False)
        with torch.no_grad():
            m.weight.copy_(1.0 + 0.3 * torch.randn(C, generator=g))
            m.bias.copy_(0.2 * torch.randn(C, generator=g))
        dy = torch.randn(shape, generator=g).to(dev, dtype).contiguous(memory_format=torch.channels_last)
        ds = torch.randn(shape, generator=g).to(dev, dtype).contiguous(memory_format=torch.channels_last)
        xr = x.float().clone().requires_grad_(True)
        yr = F.silu(F.group_norm(xr, 32, m.weight.float(), m.bias.float(), m.eps))
        (yr * dy.float()).sum().backward()
        g_norm = xr.grad.clone()
        tol = (4e-3 if dtype == torch.float16 else 3e-5) * max(float((g_norm + ds.float()).abs().max()), 1e-3)
        # both outputs used
        xg = x.clone().requires_grad_(True)
        y, xs = fused_norm.group_norm(m, xg, silu=True, skip=True)
        assert xs.data_ptr() == xg.data_ptr() and torch.equal(y, fused_norm.group_norm(m, x, silu=True))
        ((y * dy).sum() + (xs * ds).sum()).backward()
        assert float((xg.grad.float() - (g_norm + ds.float())).abs().max()) <= tol
        # equal to autograd's own accumulation of the two-launch path up to one rounding of the sum
        xa = x.clone().requires_grad_(True)
        ((fused_norm.group_norm(m, xa, silu=True) * dy).sum() + (xa * ds).sum()).backward()
        assert float((xg.grad.float() - xa.grad.float()).abs().max()) <= tol
        # only the norm / only the branch
        xg = x.clone().requires_grad_(True)
        y, xs = fused_norm.group_norm(m, xg, silu=True, skip=True)
        (y * dy).sum().backward()
        assert float((xg.grad.float() - g_norm).abs().max()) <= tol
        xg = x.clone().requires_grad_(True)
        y, xs = fused_norm.group_norm(m, xg, silu=True, skip=True)
        (xs * ds).sum().backward()
        assert torch.equal(xg.grad, ds)
        # no grad: (y, x)
        with torch.no_grad():
            y, xs = fused_norm.group_norm(m, x, silu=True, skip=True)
        assert xs is x


def test_what_falls_back_to_torch_and_what_is_rejected():
    _need_gpu()
    from dreammesh4d_amd import _lib, fused_norm

    dev = torch.device("cuda:0")
    m = torch.nn.GroupNorm(32, 64).to(dev)
    x = torch.randn(2, 64, 8, 8, device=dev)
    assert not fused_norm.fused_ok(m, x)                                                        # NCHW
    assert not fused_norm.fused_ok(m, x.contiguous(memory_format=torch.channels_last))          # trainable gamma with grad enabled
    with torch.no_grad():
        assert fused_norm.fused_ok(m, x.contiguous(memory_format=torch.channels_last))
        y = fused_norm.group_norm(m, x, silu=True)                                               # torch path
        torch.testing.assert_close(y, F.silu(m(x)))
    assert not fused_norm.fused_ok(m.half(), x.contiguous(memory_format=torch.channels_last))   # dtype mismatch
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    z = torch.zeros(64, device=dev)
    p = z.data_ptr()
    assert L.dm4d_groupnorm_nhwc_forward(1, 4, 6, 4, 1, p, 0, 0, p, p, 1e-5, 0, p, p, p, 1, s) != 0       # C not a multiple of G
    assert L.dm4d_groupnorm_nhwc_forward(1, 4, 8, 4, 7, p, 0, 0, p, p, 1e-5, 0, p, p, p, 1, s) != 0       # dtype
    assert L.dm4d_groupnorm_nhwc_forward(1, 4, 8, 4, 1, p, 0, 0, p, p, 1e-5, 0, p, p, p, 1000, s) != 0    # splits
    assert L.dm4d_groupnorm_nhwc_forward(1, 4, 8, 4, 1, 0, 0, 0, p, p, 1e-5, 0, p, p, p, 1, s) != 0       # null x
    assert L.dm4d_groupnorm_nhwc_forward(1, 4, 8, 4, 1, p, p, 3, p, p, 1e-5, 0, p, p, p, 1, s) != 0       # add_stride not 0 / C
    assert L.dm4d_groupnorm_nhwc_forward(0, 4, 8, 4, 1, 0, 0, 0, 0, 0, 1e-5, 0, 0, 0, 0, 1, s) == 0       # empty batch
    assert L.dm4d_add_bias_nhwc(4, 6, 1, p, p, p, p, s) != 0 and L.dm4d_geglu(4, 6, 1, p, p, s) != 0     # C / D not a multiple of 4
    assert L.dm4d_add_bias_nhwc(0, 8, 1, 0, 0, 0, 0, s) == 0 and L.dm4d_geglu(0, 8, 1, 0, 0, s) == 0


def test_guidance_step_nhwc_against_the_nchw_library_path():
    """The whole SDS step (UNet under no_grad + VAE encoder forward / backward) with channels-last activations and the HIP
    GroupNorm against the NCHW path through torch's GroupNorm: same loss and same gradient up to float16 rounding."""
    _need_gpu()
    import os

    from dreammesh4d_amd import fused_norm, zero123 as z

    dev = torch.device("cuda:0")
    outs = []
    for cl in (False, True):
        torch.manual_seed(0)
        model = z.Zero123(unet_kwargs=dict(model_channels=64, context_dim=32, num_heads=4), vae_kwargs=dict(ch=32)).to(dev)
        with torch.no_grad():
            for p in model.model.diffusion_model.out.parameters():
                torch.nn.init.normal_(p, std=0.05)
            for blk in model.model.diffusion_model.modules():       # zero-initialised convolutions: give them something to do
                if isinstance(blk, z.ResBlock):
                    torch.nn.init.normal_(blk.out_layers[3].weight, std=0.02)
                if isinstance(blk, z.SpatialTransformer):
                    torch.nn.init.normal_(blk.proj_out.weight, std=0.02)
        g = torch.Generator(device="cpu").manual_seed(1)
        guid = z.TemporalStableZero123Guidance(model, torch.randn(4, 1, 32, generator=g), torch.randn(4, 4, 32, 32, generator=g),
                                               cond_elevation_deg=5.0, half_precision_weights=True, use_graphs=False, channels_last=cl).to(dev)
        rgb = torch.rand(2, 256, 256, 3, generator=g).to(dev).requires_grad_(True)
        if cl:      # STRICT: a GroupNorm / add / GEGLU on torch operators, a 3x3 convolution on MIOpen, a q/k/v projection or a
            os.environ["DM4D_STRICT_FUSED"] = "1"      # supported attention on the library inside the step raises here
        try:
            out = guid(rgb, torch.tensor([10.0, 20.0], device=dev), torch.tensor([30.0, 200.0], device=dev), torch.full((2,), 3.8, device=dev),
                       frame_indices=torch.tensor([1, 3], device=dev), noise=torch.randn(2, 4, 32, 32, generator=g).to(dev),
                       t=torch.tensor([300, 420], device=dev))
        finally:
            os.environ.pop("DM4D_STRICT_FUSED", None)
        out["loss_sds"].backward()
        outs.append((float(out["loss_sds"]), rgb.grad.clone()))
        if cl:
            # every GroupNorm / residual add / GEGLU of the NHWC step ran on its HIP operator; a layout regression is LOUD:
            # an NCHW activation reaching group_norm inside the step is counted and (strict mode) raises
            assert fused_norm.fallback_count() == before, fused_norm.FALLBACKS
            os.environ["DM4D_STRICT_FUSED"] = "1"
            try:
                with pytest.raises(RuntimeError, match="fell back to torch operators"):
                    with fused_norm.expect_fused():
                        with torch.no_grad():
                            fused_norm.group_norm(torch.nn.GroupNorm(32, 64).to(dev), torch.randn(2, 64, 8, 8, device=dev))
            finally:
                del os.environ["DM4D_STRICT_FUSED"]
            assert fused_norm.fallback_count() == before + 1
        else:
            before = fused_norm.fallback_count()
    with pytest.raises(ValueError):                                    # `add` of the wrong shape is rejected, not read out of bounds
        with torch.no_grad():
            fused_norm.group_norm(torch.nn.GroupNorm(32, 64).to(dev), torch.randn(2, 64, 8, 8, device=dev).contiguous(memory_format=torch.channels_last),
                                  add=torch.zeros(3, 64, device=dev))
    (l0, g0), (l1, g1) = outs
    assert abs(l0 - l1) <= 2e-2 * abs(l0), (l0, l1)
    assert float((g0 - g1).abs().max()) <= 0.05 * float(g0.abs().max()) and float((g0 - g1).norm()) <= 0.02 * float(g0.norm())


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_residual_bias_add_and_geglu(dtype):
    """csrc/pointwise.hip: a + b + bias[c] on NHWC activations (with its pass-through gradient) and GEGLU, against torch."""
    _need_gpu()
    from dreammesh4d_amd import fused_norm

    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    tol = dict(rtol=2e-3, atol=2e-3) if dtype == torch.float16 else dict(rtol=1e-6, atol=1e-6)
    for shape in ((2, 320, 32, 32), (3, 64, 5, 7), (1, 1280, 8, 8)):
        a = torch.randn(shape, generator=g).to(dev, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        b = torch.randn(shape, generator=g).to(dev, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        bias = torch.randn(shape[1], generator=g).to(dev, dtype)
        y = fused_norm.add_bias(a, b, bias)
        assert fused_norm.is_channels_last(y) and y.grad_fn is not None and type(y.grad_fn).__name__.startswith("_AddBias")
        torch.testing.assert_close(y.float(), a.float() + b.float() + bias.float().view(1, -1, 1, 1), **tol)
        dy = torch.randn(shape, generator=g).to(dev, dtype)
        y.backward(dy)
        assert torch.equal(a.grad, dy) and torch.equal(b.grad, dy)
        y2 = fused_norm.add_bias(a.detach().contiguous(), b.detach(), bias)              # NCHW operand: torch path
        torch.testing.assert_close(y2.float(), y.detach().float(), **tol)
    for rows, D in (((8, 1024), 1280), ((2, 7), 40), ((3, 64), 5120)):
        p = torch.randn(*rows, 2 * D, generator=g).to(dev, dtype)
        with torch.no_grad():
            y = fused_norm.geglu(p)
        x_, gate = p.float().chunk(2, dim=-1)
        torch.testing.assert_close(y.float(), x_ * F.gelu(gate), **tol)
        assert y.shape == (*rows, D) and y.dtype == dtype
        pr = p.clone().requires_grad_(True)                                               # gradient wanted: torch path
        fused_norm.geglu(pr).sum().backward()
        assert pr.grad is not None


def test_add_layer_norm_matches_torch():
    """dm4d_add_layernorm_f16: LayerNorm(x + tok) and x + tok + bias2 in one launch, against torch in float32."""
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from dreammesh4d_amd.fused_norm import add_layer_norm

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    for (B, L, C) in ((2, 64, 320), (3, 17, 640), (1, 5, 1280), (2, 9, 2048)):
        x = torch.randn(B, L, C, generator=g).to(dev).half()
        tok = torch.randn(B, 1, C, generator=g).to(dev).half()
        b2 = torch.randn(C, generator=g).to(dev).half()
        ln = torch.nn.LayerNorm(C).to(dev).half()
        with torch.no_grad():
            ln.weight.copy_(torch.randn(C, generator=g).to(dev)); ln.bias.copy_(torch.randn(C, generator=g).to(dev))
        for t, b in ((None, None), (tok, None), (tok, b2), (None, b2)):
            n, xb = add_layer_norm(ln, x, t, b)
            s = x if t is None else (x + t)                                   # float16 sum, as the separate add would leave it
            ref_n = torch.nn.functional.layer_norm(s.float(), (C,), ln.weight.float(), ln.bias.float(), ln.eps)
            ref_xb = s.float() + (0 if b is None else b.float())
            assert (n.float() - ref_n).abs().max() <= 2 ** -9 * max(1.0, ref_n.abs().max().item())
            assert (xb.float() - ref_xb).abs().max() <= 2 ** -10 * max(1.0, ref_xb.abs().max().item())
    with pytest.raises(ValueError):
        add_layer_norm(torch.nn.LayerNorm(12).to(dev).half(), torch.zeros(1, 2, 12, device=dev, dtype=torch.float16))


def test_statistics_survive_a_large_mean():
    """|mean| / std = 3000 in float32: the two-moment form E[x^2] - E[x]^2 loses every digit of the variance (1e-7 x 9e6); the
    operator accumulates SHIFTED sums (csrc/groupnorm.hip) and must stay at the accuracy of a float64 reference."""
    _need_gpu()
    from dreammesh4d_amd import fused_norm

    dev = torch.device("cuda:0")
    N, C, H, W, G = 2, 64, 24, 20, 32
    g = torch.Generator(device="cpu").manual_seed(11)
    noise = 0.1 * torch.randn(N, C, H, W, generator=g, dtype=torch.float64)
    offs = 300.0 + 2.0 * torch.randn(N, C, 1, 1, generator=g, dtype=torch.float64) * 0.01          # channels of a group differ a little
    x64 = noise + offs
    x = x64.float().to(dev).contiguous(memory_format=torch.channels_last)
    m = torch.nn.GroupNorm(G, C, eps=1e-5).to(dev)
    for p in m.parameters():
        p.requires_grad_(False)
    with torch.no_grad():
        y = fused_norm.group_norm(m, x)
    want = F.group_norm(x.double().cpu(), G, None, None, 1e-5)              # float64 statistics of the float32 data
    err = float((y.double().cpu() - want).abs().max())
    assert err < 5e-3, err                                                   # (the two-moment float32 form: errors of order 1)


@pytest.mark.parametrize("N,C,H,W", [(8, 320, 16, 16), (8, 640, 8, 8), (8, 1920, 16, 16), (8, 2560, 4, 4), (2, 512, 12, 16), (3, 64, 1, 1)])
def test_slab_resident_single_launch_equals_the_two_pass_kernels(N, C, H, W, monkeypatch):
    """csrc/groupnorm.hip::k_groupnorm_slab_f16 (samples of <= 16 x 16 pixels: one launch, the slab in registers) against a float64
    GroupNorm (+ add, SiLU) of the same float16 data, with a large common offset in the data (|mean| / std = 60: shifted sums) --
    and the statistics it leaves for the backward."""
    _need_gpu()
    from dreammesh4d_amd import fused_norm

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C + H)
    x = (6.0 + 0.1 * torch.randn(N, C, H, W, generator=g)).half().to(dev).contiguous(memory_format=torch.channels_last)
    add = torch.randn(N, C, generator=g).half().to(dev) * 0.05
    m = torch.nn.GroupNorm(32, C, eps=1e-6).to(dev).half()
    with torch.no_grad():
        m.weight.copy_(torch.randn(C, generator=g).to(dev)); m.bias.copy_(torch.randn(C, generator=g).to(dev))
    for p in m.parameters():
        p.requires_grad_(False)
    with torch.no_grad():
        y = fused_norm.group_norm(m, x, silu=True, add=add)
    xin = x.double().cpu() + add.double().cpu()[:, :, None, None]
    want = F.silu(F.group_norm(xin, 32, m.weight.double().cpu(), m.bias.double().cpu(), 1e-6))
    err = float((y.double().cpu() - want).abs().max())
    assert err <= 2 ** -9 * float(want.abs().max()) + 2e-3, err              # float16 rounding of the result
