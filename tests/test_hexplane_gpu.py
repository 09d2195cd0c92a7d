"""GPU parity of the fused HexPlane kernels and of the fused deformation MLP against the PyTorch-op path of the
same module (which the CPU golden test pins to the reference's geometry/deformation.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")


@pytest.mark.parametrize("resolution,multires,M,B,heads", [((8, 8, 8, 5), (1, 2), 37, 3, "all"),
                                                           ((64, 64, 64, 25), (1, 2, 4, 8), 1000, 4, "all"),
                                                           ((16, 16, 16, 7), (1, 2, 4), 333, 2, "pos+rot")])
@pytest.mark.parametrize("layout", ["channels_last", "contiguous"])
def test_fused_hexplane_matches_grid_sample_path(resolution, multires, M, B, heads, layout):
    """Both plane storages the C ABI accepts: the module's own (torch.channels_last) and the reference's contiguous
    [1,32,H,W] tensors."""
    _need_gpu()
    from dreammesh4d_amd.deformation import DeformationNetwork

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    full = heads == "all"
    net = DeformationNetwork(resolution=resolution, multires=multires, no_ds=not full, no_dr=False, no_do=not full).to(dev)
    planes = [p for grid in net.deformation_net.grid.grids for p in grid]
    assert all(p.is_contiguous(memory_format=torch.channels_last) and p.shape[1] == 32 for p in planes)
    if layout == "contiguous":
        for p in planes:
            p.data = p.data.contiguous()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if "_deform" in name:
                p.add_((0.05 * torch.randn(p.shape, generator=g)).to(dev))
    nodes = (torch.rand(M, 3, generator=g) * 1.3 - 0.65).to(dev)
    nodes[0] = torch.tensor([1.4, -1.3, 0.2])                   # outside the aabb -> border clamp
    ts = torch.linspace(0, 1, B + 2)[1:-1].to(dev)
    ts[0] = 0.0                                                  # t = -1 exactly: lower border of the time axis
    # fused path
    out = net.node_outputs(nodes, ts)
    present = [x is not None for x in out]
    assert present == ([True] * 4 if full else [True, True, False, False])
    out = [x for x in out if x is not None]
    w = [torch.randn(x.shape, generator=g).to(dev) for x in out]
    loss = sum((a * b).sum() for a, b in zip(out, w))
    loss.backward()
    fused = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    net.zero_grad(set_to_none=True)
    # PyTorch-op path on the same device
    pts = nodes.unsqueeze(0).expand(B, M, 3).reshape(-1, 3)
    t = (ts.view(B, 1, 1).expand(B, M, 1).reshape(-1, 1)) * 2.0 - 1.0
    dx, dr, ds, do = net.forward_dynamic_delta(pts, t)
    ref = (dx.view(B, M, 3), dr.view(B, M, 4)) + ((ds.view(B, M, 6), do.view(B, M)) if full else ())
    for a, b in zip(out, ref):
        assert (a - b).abs().max() < 2e-6
    loss2 = sum((a * b).sum() for a, b in zip(ref, w))
    loss2.backward()
    for n, p in net.named_parameters():
        if p.grad is None:
            continue
        scale = p.grad.abs().max() + 1e-12
        assert (fused[n] - p.grad).abs().max() / scale < 2e-4, n
    # the structured-sparse exchange of the multi-GPU step relies on this: spatial planes get gradient ONLY at the
    # plan's touched texels
    from dreammesh4d_amd.distributed import storage_flat, touched_from_plan
    touched = touched_from_plan(net.deformation_net.grid, net._hex_plan)
    assert len(touched) == 3 * len(multires)
    for par, idx in touched.items():
        mask = torch.ones(par.numel(), dtype=torch.bool, device=dev)
        mask[idx] = False
        gpar = fused[[n for n, q in net.named_parameters() if q is par][0]]
        assert gpar.stride() == par.stride()
        assert not storage_flat(gpar)[mask].any() and storage_flat(gpar)[idx].any()
        assert idx.numel() == idx.unique().numel() <= par.numel()
    # deterministic: the gather backward gives bit-identical gradients on a re-run
    net.zero_grad(set_to_none=True)
    out3 = [x for x in net.node_outputs(nodes, ts) if x is not None]
    sum((a * b).sum() for a, b in zip(out3, w)).backward()
    for n, p in net.named_parameters():
        if p.grad is not None:       # HexPlane gather backward and the MLP's fixed-order row sums
            assert torch.equal(p.grad, fused[n]), n


def test_gradient_message_pack_unpack_matches_torch_path():
    """csrc/gradpack.hip (one launch each way) against the torch-op path the gloo CPU tests exercise."""
    _need_gpu()
    from dreammesh4d_amd.distributed import GradAllReducer

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    shapes = [(1, 8, 9, 7), (13,), (5, 6), (3,)]
    params = [torch.nn.Parameter(torch.zeros(s, device=dev)) for s in shapes]
    idx = torch.randperm(8 * 9 * 7, generator=g)[:40].sort().values.to(dev)
    grads = [torch.randn(s, generator=g).to(dev) for s in shapes]
    sparse = torch.zeros(8 * 9 * 7, device=dev)
    sparse[idx] = grads[0].view(-1)[idx]
    grads[0] = sparse.view(shapes[0])
    for p, gr in zip(params, grads):
        p.grad = gr.clone()
    params[3].grad = None                                    # a parameter that got no gradient
    red = GradAllReducer(params, touched={params[0]: idx})
    red.pack()
    want = torch.cat([grads[0].view(-1)[idx], grads[1].view(-1), grads[2].view(-1), torch.zeros(3, device=dev)])
    assert torch.equal(red.flat, want)
    assert red.nbytes == (40 + 13 + 30 + 3) * 4 and red.dense_elements == 8 * 9 * 7 + 13 + 30 + 3
    red.flat.mul_(2.0)                                       # stand-in for the all-reduce (sum of 2 equal ranks)
    red.unpack(0.5)
    for p, gr in zip(params[:3], grads[:3]):
        assert torch.equal(p.grad, gr)
    assert params[3].grad is not None and not params[3].grad.any()


def test_in_place_gradient_buffers_equal_fresh_gradients_over_steps():
    """`grads_in_place`: persistent dense gradient planes installed as `.grad` (no 134 MB zero fill per step:
    DM4D_HEX_KEEP_SPATIAL).  Over steps whose timestamps MOVE (other time rows are touched every step) the gradients must
    be bit-identical to the plain path's, the buffers must be the same storage every step, and a backward that finds a
    `.grad` in place (accumulation) must fall back to the plain path and accumulate."""
    _need_gpu()
    from dreammesh4d_amd.deformation import DeformationNetwork

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    nets = []
    for in_place in (False, True):
        torch.manual_seed(0)
        net = DeformationNetwork(resolution=(16, 16, 16, 9), multires=(1, 2, 4), no_ds=False, no_dr=False, no_do=False).to(dev)
        with torch.no_grad():
            gg = torch.Generator().manual_seed(1)
            for name, p in net.named_parameters():
                if "_deform" in name:
                    p.add_((0.05 * torch.randn(p.shape, generator=gg)).to(dev))
        net.grads_in_place = in_place
        nets.append(net)
    M, B = 300, 3
    nodes = (torch.rand(M, 3, generator=g) * 1.3 - 0.65).to(dev)
    ptrs = None
    for step in range(4):
        ts = torch.rand(B, generator=g).to(dev)
        w = None
        grads = []
        for net in nets:
            if step != 3:
                net.zero_grad(set_to_none=True)      # step 3: gradients of step 2 left in place -> accumulation
            out = [x for x in net.node_outputs(nodes, ts) if x is not None]
            if w is None:
                w = [torch.randn(x.shape, generator=g).to(dev) for x in out]
            sum((a * b).sum() for a, b in zip(out, w)).backward()
            grads.append({n: p.grad for n, p in net.named_parameters() if p.grad is not None})
        assert grads[0].keys() == grads[1].keys()
        for n in grads[0]:
            assert torch.equal(grads[0][n], grads[1][n]), (step, n)
        planes = [p for grid in nets[1].deformation_net.grid.grids for p in grid]
        now = [p.grad.data_ptr() for p in planes]
        assert ptrs is None or now == ptrs, step           # the same persistent storage every step
        ptrs = now
        assert nets[1]._hex_plan.grad_buffers is not None and all(p.grad is b for p, b in zip(planes, nets[1]._hex_plan.grad_buffers))


@pytest.mark.parametrize("heads", ["all", "pos+rot"])
def test_node_network_operator_is_bit_identical_to_the_two_operator_path(heads):
    """csrc/nodenet.hip (HexPlane query + MLP in one launch, 2 t - 1 inside, the backward's independent jobs side by side) runs
    the same kernel bodies as hexplane.hip + deform_mlp.hip behind torch's addcmul: outputs and every gradient bit for bit."""
    _need_gpu()
    from dreammesh4d_amd.deformation import DeformationNetwork

    dev = torch.device("cuda:0")
    full = heads == "all"
    g = torch.Generator().manual_seed(5)
    M, B = 700, 4
    nodes = (torch.rand(M, 3, generator=g) * 1.3 - 0.65).to(dev)
    ts = torch.rand(B, generator=g).to(dev)
    ts[1] = 0.0
    res = []
    for fuse in (True, False):
        torch.manual_seed(0)
        net = DeformationNetwork(resolution=(16, 16, 16, 9), multires=(1, 2, 4, 8), no_ds=not full, no_dr=False, no_do=not full).to(dev)
        with torch.no_grad():
            gg = torch.Generator().manual_seed(1)
            for name, p in net.named_parameters():
                if "_deform" in name:
                    p.add_((0.05 * torch.randn(p.shape, generator=gg)).to(dev))
        net.fuse_node_network = fuse
        out = [x for x in net.node_outputs(nodes, ts) if x is not None]
        gw = torch.Generator().manual_seed(2)
        w = [torch.randn(x.shape, generator=gw).to(dev) for x in out]
        sum((a * b).sum() for a, b in zip(out, w)).backward()
        res.append(([o.detach() for o in out], {n: p.grad for n, p in net.named_parameters() if p.grad is not None}))
    (o1, g1), (o2, g2) = res
    assert len(o1) == len(o2) == (4 if full else 2)
    for a, b in zip(o1, o2):
        assert torch.equal(a, b)
    assert g1.keys() == g2.keys() and len(g1) > 20
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n
