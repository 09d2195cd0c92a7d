"""-m gpu: the optimiser step of the path in MESSAGE space as two launches (csrc/gradpack.hip, ``dm4d_adamw_message``,
``distributed.ShardedAdamW`` with one process on a HIP device) against ``torch.optim.AdamW`` over ALL elements and against the
torch-operator form of the same class: dense and index-listed parameters (a contiguous and a channels_last plane), two groups with
different learning rates, a step skipped on the device by ``found_inf``, the deferred weight decay of the elements outside the
message applied by ``materialize()``."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dev, seed):
    g = torch.Generator().manual_seed(seed)
    mlp = torch.nn.Parameter(torch.randn(64, 48, generator=g).to(dev))
    bias = torch.nn.Parameter(torch.randn(48, generator=g).to(dev))
    plane_a = torch.nn.Parameter(torch.randn(1, 16, 24, 20, generator=g).to(dev))
    plane_b = torch.nn.Parameter(torch.randn(1, 16, 12, 10, generator=g).to(dev).contiguous(memory_format=torch.channels_last))
    ia = torch.randperm(plane_a.numel(), generator=g)[:900].sort().values.to(dev)
    ib = torch.randperm(plane_b.numel(), generator=g)[:300].sort().values.to(dev)
    return [mlp, bias, plane_a, plane_b], {plane_a: ia, plane_b: ib}


def _grads(params, touched, step, dev):
    from dreammesh4d_amd.distributed import storage_flat

    g = torch.Generator().manual_seed(1000 + step)
    out = []
    for p in params:
        gr = torch.randn(p.shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last if p.dim() == 4 and not p.is_contiguous() else torch.contiguous_format)
        if p in touched:                     # gradient only where the message has an element (as the HexPlane planes)
            m = torch.zeros(p.numel(), device=dev)
            m[touched[p]] = 1.0
            gr = (storage_flat(gr) * m).view_as(storage_flat(gr)).as_strided(p.shape, p.stride())
        out.append(gr)
    return out


def test_message_space_adamw_equals_the_dense_optimiser():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import distributed as D

    dev = torch.device("cuda:0")
    runs = {}
    for mode in ("torch", "ops", "fused"):
        params, touched = _setup(dev, 3)
        groups = [{"params": params[:2], "lr": 3.2e-3, "name": "deformation"}, {"params": params[2:], "lr": 3.2e-2, "name": "grid"}]
        if mode == "torch":
            opt = torch.optim.AdamW(groups, lr=0.0, betas=(0.9, 0.99), eps=1e-15, fused=True)
        else:
            red = D.GradAllReducer(params, touched=touched)
            opt = D.ShardedAdamW(groups, red, betas=(0.9, 0.99), eps=1e-15)
            opt.fused = mode == "fused"
        for step in range(6):
            for p, gr in zip(params, _grads(params, touched, step, dev)):
                p.grad = gr
            flag = torch.tensor(1.0 if step == 2 else 0.0, device=dev)       # step 2 is skipped on the device
            lr_scale = 0.97 ** step                                            # the schedule moves the rates between steps
            for gq, base in zip(opt.param_groups, (3.2e-3, 3.2e-2)):
                gq["lr"] = base * lr_scale
            if mode == "torch":
                opt.found_inf, opt.grad_scale = flag, None
                opt.step()
            else:
                opt.step(found_inf=flag)
        if mode != "torch":
            assert int(opt.step_t) == 5
            opt.materialize()
        runs[mode] = [p.detach().clone() for p in params]
    for k, (a, b, c) in enumerate(zip(runs["torch"], runs["ops"], runs["fused"])):
        scale = float(a.abs().max())
        assert float((a - c).abs().max()) <= 2e-6 * scale, (k, float((a - c).abs().max()), scale)
        assert float((b - c).abs().max()) <= 2e-6 * scale, (k, float((b - c).abs().max()), scale)
    # the parameters moved (the check above is not comparing initial values), also outside the message (weight decay)
    p0, _ = _setup(dev, 3)
    assert all(float((x - y).abs().max()) > 1e-4 for x, y in zip(p0, runs["fused"]))


def test_optimiser_state_round_trip():
    """state_dict / load_state_dict of the message-space optimiser: a second instance continued from the first one's state takes
    the same steps bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import distributed as D

    dev = torch.device("cuda:0")
    out = []
    for resume in (False, True):
        params, touched = _setup(dev, 4)
        groups = [{"params": params[:2], "lr": 3.2e-3, "name": "deformation"}, {"params": params[2:], "lr": 3.2e-2, "name": "grid"}]
        opt = D.ShardedAdamW(groups, D.GradAllReducer(params, touched=touched), betas=(0.9, 0.99), eps=1e-15)
        for step in range(4):
            if resume and step == 2:
                sd, vals = opt.state_dict(), [p.detach().clone() for p in params]
                params, touched = _setup(dev, 4)
                with torch.no_grad():
                    for p, v in zip(params, vals):
                        p.copy_(v)
                groups = [{"params": params[:2], "lr": 1.0, "name": "deformation"}, {"params": params[2:], "lr": 1.0, "name": "grid"}]
                opt = D.ShardedAdamW(groups, D.GradAllReducer(params, touched=touched), betas=(0.9, 0.99), eps=1e-15)
                opt.load_state_dict(sd)
            for p, gr in zip(params, _grads(params, touched, step, dev)):
                p.grad = gr
            opt.step()
        opt.materialize()
        out.append([p.detach().clone() for p in params])
    assert all(torch.equal(a, b) for a, b in zip(*out))
