"""The hipGraph replay of the Zero123 SDS step (zero123.TemporalStableZero123Guidance, use_graphs=True) must give what the
eager launches give: same loss, same gradient on the rendered images."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_graphed_sds_step_equals_eager():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import zero123 as z

    dev = torch.device("cuda:0")
    L, B = 6, 3
    torch.manual_seed(0)
    model = z.Zero123(unet_kwargs=dict(model_channels=32, context_dim=32, num_heads=4), vae_kwargs=dict(ch=32))
    for p in model.model.diffusion_model.out.parameters():
        torch.nn.init.normal_(p, std=0.05)
    cc, cat = torch.randn(L, 1, 32), torch.randn(L, 4, 32, 32)
    kw = dict(cond_elevation_deg=5.0, half_precision_weights=False)
    eager = z.TemporalStableZero123Guidance(model, cc, cat, use_graphs=False, **kw).to(dev)
    graphed = z.TemporalStableZero123Guidance(model, cc, cat, use_graphs=True, one_graph=False, **kw).to(dev)     # shares the weights
    whole = z.TemporalStableZero123Guidance(model, cc, cat, use_graphs=True, one_graph=True, **kw).to(dev)       # the whole step as ONE graph
    el = torch.tensor([10.0, 30.0, 60.0], device=dev)
    az = torch.tensor([-40.0, 90.0, 170.0], device=dev)
    fi = torch.tensor([0, 2, 5], device=dev)
    res = {}
    for step in range(3):                      # step 0 captures, 1-2 replay (with different inputs)
        g = torch.Generator().manual_seed(100 + step)
        rgb0 = torch.rand(B, 64, 64, 3, generator=g).to(dev)
        noise = torch.randn(B, 4, 32, 32, generator=g).to(dev)
        t = torch.randint(20, 980, (B,), generator=g).to(dev)
        for name, guid in (("eager", eager), ("graphed", graphed), ("whole", whole)):
            rgb = rgb0.clone().requires_grad_(True)
            torch.manual_seed(7 + step)        # the VAE posterior noise (sampled on the CPU, like the reference)
            out = guid(rgb, el, az, torch.full_like(el, 3.8), frame_indices=fi, noise=noise, t=t)
            out["loss_sds"].backward()
            res[(name, step)] = (out["loss_sds"].detach().clone(), rgb.grad.clone())
        a = res[("eager", step)]
        assert torch.isfinite(a[0]) and float(a[1].abs().max()) > 0
        for other in ("graphed", "whole"):
            b = res[(other, step)]
            assert abs(float(a[0]) - float(b[0])) <= 1e-5 * abs(float(a[0])), (other, step, float(a[0]), float(b[0]))
            assert float((a[1] - b[1]).abs().max()) <= 1e-5 * float(a[1].abs().max()), (other, step)
    assert graphed._graph_error is None, graphed._graph_error
    assert whole._graph_error is None, whole._graph_error
    assert len(graphed._unet_graphs) == 1 and len(graphed._enc_graphed) == 1 and not eager._unet_graphs and not graphed._sds_graphs
    assert len(whole._sds_graphs) == 1 and not whole._unet_graphs and not whole._enc_graphed


def test_whole_step_graph_draws_the_eager_steps_random_numbers_and_scales_with_the_upstream_gradient():
    """Without explicit noise / timesteps the one-graph step consumes the generators exactly like the eager step (posterior noise
    on the CPU generator, then t, then the noise on the device generator): same seeds, same loss.  The image gradient inside the
    graph is for an upstream gradient of 1; a weighted loss scales it."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import zero123 as z

    dev = torch.device("cuda:0")
    L, B = 4, 2
    torch.manual_seed(1)
    model = z.Zero123(unet_kwargs=dict(model_channels=32, context_dim=32, num_heads=4), vae_kwargs=dict(ch=32))
    for p in model.model.diffusion_model.out.parameters():
        torch.nn.init.normal_(p, std=0.05)
    cc, cat = torch.randn(L, 1, 32), torch.randn(L, 4, 32, 32)
    kw = dict(cond_elevation_deg=5.0, half_precision_weights=False, grad_clip=0.5)
    eager = z.TemporalStableZero123Guidance(model, cc, cat, use_graphs=False, **kw).to(dev)
    whole = z.TemporalStableZero123Guidance(model, cc, cat, use_graphs=True, one_graph=True, **kw).to(dev)
    el, az = torch.tensor([10.0, 30.0]), torch.tensor([-40.0, 90.0])          # host tensors, as DynamicStage passes them
    fi = torch.tensor([1, 3], device=dev)
    rgb0 = torch.rand(B, 256, 256, 3, generator=torch.Generator().manual_seed(5)).to(dev)
    for step in range(3):
        got = {}
        for name, guid in (("eager", eager), ("whole", whole)):
            guid.update_step(0, step, grad_clip=0.5 - 0.1 * step)              # a scheduled clip value: a static buffer, not a re-capture
            rgb = rgb0.clone().requires_grad_(True)
            torch.manual_seed(11 + step)
            torch.cuda.manual_seed(13 + step)
            out = guid(rgb, el, az, torch.full_like(el, 3.8), frame_indices=fi)
            (0.25 * out["loss_sds"]).backward()
            got[name] = (float(out["loss_sds"]), float(out["grad_norm"]), rgb.grad.clone())
        a, b = got["eager"], got["whole"]
        assert abs(a[0] - b[0]) <= 1e-5 * abs(a[0]) and abs(a[1] - b[1]) <= 1e-5 * abs(a[1]), (step, a[:2], b[:2])
        assert float(a[2].abs().max()) > 0 and float((a[2] - b[2]).abs().max()) <= 1e-5 * float(a[2].abs().max()), step
    assert whole._graph_error is None and len(whole._sds_graphs) == 1


@pytest.mark.parametrize("clip", [None, 0.3])
def test_fused_glue_kernels_equal_the_torch_operators_between_the_networks(clip):
    """csrc/sds_glue.hip (dm4d_sds_prepare / dm4d_sds_finish: posterior sample, noising, the UNet's input, guidance arithmetic,
    loss, dL/dmoments -- two launches inside the step's graph) against the ~70 torch operators they replace, float16 weights:
    the same expressions with the same roundings: loss and |grad| to 2e-6 (float32 sums in another order), the image gradient to
    float16 resolution (a last-bit difference of exp in a few elements of dL/dmoments)."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import zero123 as z

    dev = torch.device("cuda:0")
    L, B = 5, 3
    torch.manual_seed(2)
    model = z.Zero123(unet_kwargs=dict(model_channels=64, context_dim=32, num_heads=4), vae_kwargs=dict(ch=32))
    for p in model.model.diffusion_model.out.parameters():
        torch.nn.init.normal_(p, std=0.05)
    with torch.no_grad():      # a posterior log-variance that reaches the lower clamp (random weights give ~0: the clamp's mask would go untested)
        probe = model.first_stage_model.encode_moments(torch.rand(2, 3, 256, 256) * 2 - 1)[:, 4:]
        model.first_stage_model.quant_conv.weight[4:] *= 8.0 / float(probe.std())
        # (mean -14: the typical standard deviation exp(-7) is still a NORMAL float16 number -- at -20 the whole chain runs in subnormals,
        # where one last-bit difference of exp is a percent)
        model.first_stage_model.quant_conv.bias[4:] = -14.0 - 8.0 / float(probe.std()) * (probe.mean() - model.first_stage_model.quant_conv.bias[4:])
    cc, cat = torch.randn(L, 1, 32), torch.randn(L, 4, 32, 32)
    kw = dict(cond_elevation_deg=5.0, half_precision_weights=True, grad_clip=clip, use_graphs=True, one_graph=True)
    ops = z.TemporalStableZero123Guidance(model, cc, cat, **kw).to(dev)
    ops.fused_glue = False
    fused = z.TemporalStableZero123Guidance(model, cc, cat, **kw).to(dev)
    assert fused.fused_glue
    el, az = torch.tensor([10.0, 30.0, -5.0]), torch.tensor([-40.0, 90.0, 170.0])
    fi = torch.tensor([1, 4, 0], device=dev)
    for step in range(3):
        rgb0 = torch.rand(B, 256, 256, 3, generator=torch.Generator().manual_seed(50 + step)).to(dev)
        got = {}
        for name, guid in (("ops", ops), ("fused", fused)):
            rgb = rgb0.clone().requires_grad_(True)
            torch.manual_seed(21 + step)
            torch.cuda.manual_seed(23 + step)
            out = guid(rgb, el, az, torch.full_like(el, 3.8), frame_indices=fi)
            (0.5 * out["loss_sds"]).backward()
            got[name] = (float(out["loss_sds"].detach()), float(out["grad_norm"].detach()), rgb.grad.clone())
        a, b = got["ops"], got["fused"]
        assert a[0] > 0 and a[1] > 0 and float(a[2].abs().max()) > 0
        assert abs(a[0] - b[0]) <= 2e-6 * abs(a[0]) and abs(a[1] - b[1]) <= 2e-6 * abs(a[1]), (step, a[:2], b[:2])
        # (dL/dmoments differs in the last float16 bit of a few elements -- exp of the two libraries; with this test's stretched
        # log-variance that is up to ~1e-3 of the image gradient's range, 2e-5 at full size with the shipped statistics)
        # (through the random-weight encoder's float16 backward a last-bit difference in an element of dL/dmoments is amplified
        # locally: the kernels themselves are checked element by element in the next test)
        assert float((a[2] - b[2]).abs().max()) <= 2e-2 * float(a[2].abs().max()), (step, float((a[2] - b[2]).abs().max()), float(a[2].abs().max()))
        assert float((a[2] - b[2]).abs().mean()) <= 5e-3 * float(a[2].abs().mean()), step
    with torch.no_grad():
        lv = model.first_stage_model.encode_moments((rgb0.permute(0, 3, 1, 2) * 2 - 1).to(torch.float16).contiguous(memory_format=torch.channels_last))[:, 4:].float()
    assert float((lv < -30).float().mean()) > 0.005 and float((lv > -30).float().mean()) > 0.5          # both sides of the clamp were exercised
    st_o, st_f = list(ops._sds_graphs.values())[0], list(fused._sds_graphs.values())[0]
    assert ops._graph_error is None and fused._graph_error is None and st_f.fused_glue and not st_o.fused_glue


def test_glue_kernels_element_by_element_against_torch_autograd():
    """dm4d_sds_prepare / dm4d_sds_finish alone, on random float16 moments (log-variance on both sides of the lower clamp) and a
    random "UNet prediction", against the torch operators with autograd: latents and the UNet's input bit-identical, loss and
    |grad| to 1e-6, dL/dmoments identical except for isolated elements (<= 0.1 %) at most two float16 ulps apart."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import ctypes as C_

    import torch.nn.functional as F

    from dreammesh4d_amd import _lib

    dev = torch.device("cuda:0")
    Lb = _lib.lib()
    for B, clip in ((3, None), (4, 0.25), (1, None)):
        g = torch.Generator().manual_seed(10 + B)
        moments = torch.randn(B, 8, 32, 32, generator=g)
        moments[:, 4:] = moments[:, 4:] * 8 - 14
        moments = moments.to(dev, torch.float16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        post = torch.randn(B, 4, 32, 32, generator=g).to(dev, torch.float16)
        noise = torch.randn(B, 4, 32, 32, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        pred = torch.randn(2 * B, 4, 32, 32, generator=g).to(dev, torch.float16)
        t = torch.randint(20, 980, (B,), generator=g).to(dev)
        betas = torch.linspace(0.00085 ** 0.5, 0.0120 ** 0.5, 1000, dtype=torch.float64) ** 2
        alphas = torch.cumprod(1.0 - betas, dim=0).float().to(dev)
        cc = torch.randn(5, 4, 32, 32, generator=g).to(dev, torch.float16)
        fidx = torch.randint(0, 5, (B,), generator=g).to(dev)
        scale, gs = 0.18215, 3.0
        mean, logvar = moments.chunk(2, dim=1)
        latents = (scale * (mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * post)).to(torch.float32)
        with torch.no_grad():
            ac = alphas[t].view(-1, 1, 1, 1)
            noisy = ac.sqrt() * latents + (1 - ac).sqrt() * noise
            unc, cnd = pred.float().chunk(2)
            grad = torch.nan_to_num((1 - ac) * ((unc + gs * (cnd - unc)) - noise))
            if clip is not None:
                grad = grad.clamp(-clip, clip)
            target = latents - grad
        loss = 0.5 * F.mse_loss(latents, target, reduction="sum") / B
        (dm_ref,) = torch.autograd.grad(loss, moments)
        ptr = lambda v: C_.c_void_p(v.data_ptr())
        st = lambda v: (C_.c_int64 * 4)(*v.stride())
        md = moments.detach()
        lat2 = torch.zeros_like(noise)
        x_in = torch.empty(2 * B, 8, 32, 32, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        t2 = torch.empty(2 * B, dtype=torch.long, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(Lb.dm4d_sds_prepare(B, 32, 32, scale, ptr(md), st(md), ptr(post), st(post), ptr(noise), st(noise), ptr(lat2), st(lat2), ptr(t),
                                       ptr(alphas), ptr(cc), st(cc), ptr(fidx), ptr(x_in), st(x_in), ptr(t2), s), "dm4d_sds_prepare")
        dm = torch.empty(md.shape, device=dev, dtype=torch.float16)
        lo, gn = torch.empty((), device=dev), torch.empty((), device=dev)
        cl = None if clip is None else torch.tensor(clip, device=dev)
        _lib.check(Lb.dm4d_sds_finish(B, 32, 32, scale, gs, ptr(pred), st(pred), ptr(lat2), st(lat2), ptr(noise), st(noise), ptr(t), ptr(alphas),
                                      None if cl is None else ptr(cl), ptr(md), st(md), ptr(post), st(post), ptr(dm), st(dm), ptr(lo), ptr(gn), s),
                   "dm4d_sds_finish")
        assert torch.equal(latents.detach(), lat2)
        assert torch.equal(x_in[:B, :4], noisy.half()) and torch.equal(x_in[B:, :4], noisy.half())
        assert torch.equal(x_in[:B, 4:], torch.zeros_like(x_in[:B, 4:])) and torch.equal(x_in[B:, 4:], cc[fidx])
        assert torch.equal(t2, torch.cat([t, t]))
        assert abs(float(lo) - float(loss)) <= 1e-6 * float(loss) and abs(float(gn) - float(grad.norm())) <= 1e-6 * float(grad.norm())
        lv = md[:, 4:].float()
        assert float((lv < -30).float().mean()) > 0.005 and bool((dm[:, 4:][lv < -30] == 0).all())            # the clamp's mask
        off = dm != dm_ref
        assert float(off.float().mean()) <= 1e-3, float(off.float().mean())
        big = torch.maximum(dm.float().abs(), dm_ref.float().abs()).clamp_min(2.0 ** -14)                      # (subnormals: the spacing of 2^-14)
        ulp = torch.exp2(torch.floor(torch.log2(big)) - 10)
        worst = float(((dm.float() - dm_ref.float()).abs() / ulp).max())
        assert worst <= 2.0, worst          # float16 ulps: a last-bit difference in dL/dmean (<= 0.1 % of the elements) carried through two more roundings


def test_fused_unet_path_equals_plain_forward(monkeypatch):
    """The float16 / no_grad UNet on a HIP device takes shortcuts that must not change what it computes: q, k, v of a
    self-attention from ONE GEMM (strided views into its result), the residual adds in the GEMMs' C operand written in place,
    the timestep projections, single-token value and output projections of all blocks from batched GEMMs.  Against the same
    module with every switch off (the reference's op-by-op forward, openaimodel.py / attention.py), on the same weights."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import zero123 as z

    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    unet = z.UNetModel(model_channels=64, context_dim=48, num_heads=4)
    with torch.no_grad():
        for m in unet.modules():            # (the zero-initialised projections would hide the transformer blocks)
            if isinstance(m, z.SpatialTransformer):
                torch.nn.init.normal_(m.proj_out.weight, std=0.05)
        torch.nn.init.normal_(unet.out[2].weight, std=0.05)
    unet = unet.to(dev, torch.float16).to(memory_format=torch.channels_last).requires_grad_(False)
    x = torch.randn(4, 8, 32, 32, device=dev, dtype=torch.float16)
    t = torch.tensor([10, 400, 700, 990], device=dev)
    ctx = torch.randn(4, 1, 48, device=dev, dtype=torch.float16)
    with torch.no_grad():
        fused = unet(x, t, ctx).float()
        for name in ("FUSE_QKV", "FUSE_ADD_LAYERNORM", "BATCH_SMALL_GEMMS"):
            assert getattr(z, name)
            monkeypatch.setattr(z, name, False)
        plain = unet(x, t, ctx).float()
    assert torch.isfinite(plain).all() and float(plain.abs().max()) > 1e-2
    # float16 roundings in another order (one GEMM's accumulation against three, sums fused differently): ~1e-3 relative
    assert float((fused - plain).abs().max()) <= 4e-3 * float(plain.abs().max()), float((fused - plain).abs().max()) / float(plain.abs().max())


def test_conditioning_graph_on_the_side_stream_and_prefetch_equal_the_one_graph_step(monkeypatch):
    """float16 weights: the step is the conditioning graph on a side stream + the main graph (zero123._sds_graph).  With the same
    seeds it returns what the one-graph step returns (DM4D_SDS_PRE_GRAPH=0), whether the conditioning is staged by the call itself
    or ahead of it by prefetch(); a call whose tensors differ from the prefetched ones stages again."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import zero123 as z

    dev = torch.device("cuda:0")
    L, B = 5, 2
    torch.manual_seed(3)
    model = z.Zero123(unet_kwargs=dict(model_channels=32, context_dim=32, num_heads=4), vae_kwargs=dict(ch=32))
    for p in model.model.diffusion_model.out.parameters():
        torch.nn.init.normal_(p, std=0.05)
    cc, cat = torch.randn(L, 1, 32), torch.randn(L, 4, 32, 32)
    kw = dict(cond_elevation_deg=5.0, half_precision_weights=True)
    one = z.TemporalStableZero123Guidance(model, cc, cat, **kw).to(dev)
    two = z.TemporalStableZero123Guidance(model, cc, cat, **kw).to(dev)          # shares the weights
    assert two.prefetch(torch.tensor([10.0, 30.0]), torch.tensor([-40.0, 90.0])) is False      # no graphs yet: nothing to replay ahead
    el, az, fi = torch.tensor([10.0, 30.0]), torch.tensor([-40.0, 90.0]), torch.tensor([1, 4])      # host tensors
    rgb0 = torch.rand(B, 256, 256, 3, generator=torch.Generator().manual_seed(5)).to(dev)

    def run(guid, step, prefetch):
        rgb = rgb0.clone().requires_grad_(True)
        torch.manual_seed(11 + step)
        torch.cuda.manual_seed(13 + step)
        staged = guid.prefetch(el, az, frame_indices=fi) if prefetch else None
        out = guid(rgb, el, az, torch.full_like(el, 3.8), frame_indices=fi)
        out["loss_sds"].backward()
        torch.cuda.synchronize()
        return float(out["loss_sds"]), float(out["grad_norm"]), rgb.grad.clone(), staged

    for step in range(3):
        monkeypatch.setenv("DM4D_SDS_PRE_GRAPH", "0")
        a = run(one, step, False)
        monkeypatch.delenv("DM4D_SDS_PRE_GRAPH")
        b = run(two, step, False)
        c = run(two, step, True)
        assert c[3] is True                     # (the call before has captured the graphs: there is something to replay ahead)
        for other in (b, c):
            assert a[0] == other[0] and a[1] == other[1], (step, a[:2], other[:2])
            assert float(a[2].abs().max()) > 0 and torch.equal(a[2], other[2]), step
    assert one._graph_error is None and two._graph_error is None
    st1, st2 = next(iter(one._sds_graphs.values())), next(iter(two._sds_graphs.values()))
    assert st1.pre_graph is None and st2.pre_graph is not None
    # a prefetch for OTHER cameras than the call's: the call stages its own
    assert two.prefetch(el + 1.0, az, frame_indices=fi)
    d = run(two, 2, False)
    assert two.__dict__.get("_prefetched") is None
    assert torch.isfinite(torch.tensor(d[0])) and float(d[2].abs().max()) > 0
