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
    graphed = z.TemporalStableZero123Guidance(model, cc, cat, use_graphs=True, **kw).to(dev)     # shares the weights
    el = torch.tensor([10.0, 30.0, 60.0], device=dev)
    az = torch.tensor([-40.0, 90.0, 170.0], device=dev)
    fi = torch.tensor([0, 2, 5], device=dev)
    res = {}
    for step in range(3):                      # step 0 captures, 1-2 replay (with different inputs)
        g = torch.Generator().manual_seed(100 + step)
        rgb0 = torch.rand(B, 64, 64, 3, generator=g).to(dev)
        noise = torch.randn(B, 4, 32, 32, generator=g).to(dev)
        t = torch.randint(20, 980, (B,), generator=g).to(dev)
        for name, guid in (("eager", eager), ("graphed", graphed)):
            rgb = rgb0.clone().requires_grad_(True)
            torch.manual_seed(7 + step)        # the VAE posterior noise (sampled on the CPU, like the reference)
            out = guid(rgb, el, az, torch.full_like(el, 3.8), frame_indices=fi, noise=noise, t=t)
            out["loss_sds"].backward()
            res[(name, step)] = (out["loss_sds"].detach().clone(), rgb.grad.clone())
        a, b = res[("eager", step)], res[("graphed", step)]
        assert torch.isfinite(a[0]) and float(a[1].abs().max()) > 0
        assert abs(float(a[0]) - float(b[0])) <= 1e-5 * abs(float(a[0])), (step, float(a[0]), float(b[0]))
        assert float((a[1] - b[1]).abs().max()) <= 1e-5 * float(a[1].abs().max()), step
    assert graphed._graph_error is None, graphed._graph_error
    assert len(graphed._unet_graphs) == 1 and len(graphed._enc_graphed) == 1 and not eager._unet_graphs
