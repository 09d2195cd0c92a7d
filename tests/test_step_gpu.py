"""-m gpu: the step object (dreammesh4d_amd/step.py, csrc/step.hip: dm4d_step_*) against the two operators it wraps --
``DeformationNetwork.node_outputs`` + ``views.render_views`` -- on the same inputs: the same kernels in the same order, so images,
node outputs and every parameter gradient must agree BIT FOR BIT; plus its contract (one step in flight, gradients dropped
every step, rebuild when the capacities grow)."""
import numpy as np
import pytest
import torch

from dreammesh4d_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _setup(dev, method="hybrid", n_faces=2400, M=100, B=4, NF=2, H=144, W=176, seed=3):
    from dreammesh4d_amd import geometry as geo, ops, views
    from dreammesh4d_amd.deformation import DeformationNetwork

    sc = syn.mesh_bound_scene(n_faces, n_nodes=M, k=4, seed=seed)
    T = lambda a: torch.tensor(a, device=dev)
    graph = ops.DeformGraph(sc["verts"], sc["nbr_idx"], sc["nbr_w"], M, dev)
    topo = ops.MeshTopology(sc["faces"], len(sc["verts"]), 6, dev)
    verts, faces = T(sc["verts"]), T(sc["faces"])
    st = dict(qs=geo.quaternions(verts, faces, T(sc["complex"]), 6), sc=geo.scaling(T(sc["log_scales"]), syn.THICKNESS),
              op=geo.strengths(T(sc["densities"])), rgb=geo.points_rgb(T(sc["sh_dc"])))
    torch.manual_seed(0)
    net = DeformationNetwork(resolution=(16, 16, 16, 9), multires=(1, 2), no_ds=False, no_dr=False, no_do=False).to(dev)
    g0 = torch.Generator(device="cpu").manual_seed(5)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if "_deform" in name:
                p.add_((0.02 * torch.randn(p.shape, generator=g0)).to(dev))
    cams = [syn.make_camera(H, W, elev_deg=10 + 9 * b, azim_deg=-100 + 71 * b) for b in range(B)]
    vm = torch.stack([T(c.viewmatrix) for c in cams])
    pm = torch.stack([T(c.projmatrix) for c in cams])
    fidx = torch.tensor([0, 1, 1, 0][:B], device=dev, dtype=torch.int32) if NF != B else None
    t = torch.tensor([0.21, 0.67][:NF], device=dev)
    r = lambda: views.ViewRenderer(graph, topo, H, W, cams[0].tanfov, method=method)
    gen = torch.Generator().manual_seed(1)
    gC, gD, gA = torch.randn(B, 6, H, W, generator=gen).to(dev), (0.1 * torch.randn(B, 1, H, W, generator=gen)).to(dev), torch.randn(B, 1, H, W, generator=gen).to(dev)
    gV = (0.01 * torch.randn(NF, graph.V, 3, generator=gen)).to(dev)
    return sc, net, T(sc["nodes"]), st, vm, pm, fidx, t, r, (gC, gD, gA, gV)


@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("method,depth", [("hybrid", False), ("hybrid", True), ("lbs", False), ("dqs", True)])
def test_step_object_is_bit_identical_to_node_outputs_plus_render_views(method, depth, fuse):
    """Bit-identical FOR THE SAME fuse_face_backward SETTING (fuse = True is what DynamicStage / bench.py run: record gather + face
    backward as one kernel on both paths); the fused setting against the two-kernel setting: images bit-identical, parameter gradients
    equal up to the order in which a frame's views are added (test_fused_face_backward_equals_two_kernels_within_rounding)."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import views
    from dreammesh4d_amd.step import DynamicStep

    dev = torch.device("cuda:0")
    sc, net, nodes, st, vm, pm, fidx, t, mk, (gC, gD, gA, gV) = _setup(dev, method)
    bg6 = torch.ones(6, device=dev)
    params = [p for p in net.parameters() if p.requires_grad]

    def backward(out):
        outs, gs = [out["color"], out["alpha"], out["vxyz"]], [gC, gA, gV]
        if depth:
            outs.append(out["depth"]); gs.append(gD)
        torch.autograd.backward(outs, gs)

    # reference: the two operators
    r1 = mk()
    r1.fuse_face_backward = fuse
    net.grads_in_place = False
    dx, dr, ds, do = net.node_outputs(nodes, t)
    o1 = views.render_views(r1, dx, dr, ds, do, st["qs"], st["sc"], st["op"], st["rgb"], vm, pm, bg6, frame_index=fidx)
    backward(o1)
    ref = {k: v.detach().clone() for k, v in o1.items()}
    ref_nodes = [None if x is None else x.detach().clone() for x in (dx, dr, ds, do)]
    ref_grads = [None if p.grad is None else p.grad.detach().clone() for p in params]
    for p in params:
        p.grad = None
    # the step object (twice: the second call reuses every buffer)
    r2 = mk()
    r2.fuse_face_backward = fuse
    step = DynamicStep(r2, net, nodes, st["qs"], st["sc"], st["op"], st["rgb"], bg6, n_views=vm.shape[0], n_frames=t.shape[0])
    for rep in range(2):
        o2 = step(t, vm, pm, fidx)
        for k in ("color", "depth", "alpha", "radii", "vxyz", "vrot"):
            assert torch.equal(o2[k], ref[k]), (k, rep)
        no = step.node_outputs()
        for name, want in zip(("dx", "dr", "ds", "do"), ref_nodes):
            if want is not None:
                assert torch.equal(no[name].view(want.shape), want), name
        backward(o2)
        n_checked = 0
        for p, want in zip(params, ref_grads):
            if want is None or not bool(want.any()):
                continue              # (parameters the two-operator path leaves without gradient: the unused timenet)
            assert p.grad is not None
            assert torch.equal(p.grad, want), rep
            n_checked += 1
        assert n_checked >= 10
        assert r2.check() == r1.check()
        for p in params:
            p.grad = None


def test_fused_face_backward_equals_two_kernels_within_rounding():
    """The shipped training path (fuse_face_backward = True) against the two-kernel path through the step object: the same images bit for
    bit; every parameter gradient within 2e-6 of the tensor's scale (the per-view corner records are summed in another order)."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd.step import DynamicStep

    dev = torch.device("cuda:0")
    sc, net, nodes, st, vm, pm, fidx, t, mk, (gC, gD, gA, gV) = _setup(dev, "hybrid")
    bg6 = torch.ones(6, device=dev)
    params = [p for p in net.parameters() if p.requires_grad]
    res = {}
    for fuse in (False, True):
        r = mk()
        r.fuse_face_backward = fuse
        step = DynamicStep(r, net, nodes, st["qs"], st["sc"], st["op"], st["rgb"], bg6, n_views=vm.shape[0], n_frames=t.shape[0])
        o = step(t, vm, pm, fidx)
        torch.autograd.backward([o["color"], o["alpha"], o["vxyz"]], [gC, gA, gV])
        res[fuse] = ({k: o[k].detach().clone() for k in ("color", "depth", "alpha")}, [None if p.grad is None else p.grad.detach().clone() for p in params])
        for p in params:
            p.grad = None
    for k in ("color", "depth", "alpha"):
        assert torch.equal(res[False][0][k], res[True][0][k]), k
    n = 0
    for a, b in zip(res[False][1], res[True][1]):
        if a is None or not bool(a.any()):
            continue
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()), float((a - b).abs().max() / a.abs().max())
        n += 1
    assert n >= 10


def test_step_object_rgb_gradient_only():
    """dm4d_step_backward_rgb (what DynamicStage runs with the shipped loss weights: nothing reads the normal image): every parameter
    gradient bit-identical to dm4d_step_backward fed exact zeros on channels 3..5; the upstream gradient's normal channels hold NaN."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd.step import DynamicStep

    dev = torch.device("cuda:0")
    sc, net, nodes, st, vm, pm, fidx, t, mk, (gC, gD, gA, gV) = _setup(dev, "hybrid")
    bg6 = torch.ones(6, device=dev)
    params = [p for p in net.parameters() if p.requires_grad]
    gz, gj = gC.clone(), gC.clone()
    gz[:, 3:] = 0.0
    gj[:, 3:] = float("nan")
    res = {}
    for rgb in (False, True):
        r = mk()
        r.fuse_face_backward = True
        r.rgb_gradient_only = rgb
        step = DynamicStep(r, net, nodes, st["qs"], st["sc"], st["op"], st["rgb"], bg6, n_views=vm.shape[0], n_frames=t.shape[0])
        o = step(t, vm, pm, fidx)
        torch.autograd.backward([o["color"], o["alpha"], o["vxyz"]], [gj if rgb else gz, gA, gV])
        res[rgb] = [None if p.grad is None else p.grad.detach().clone() for p in params]
        for p in params:
            p.grad = None
    n = 0
    for a, b in zip(res[False], res[True]):
        if a is None or not bool(a.any()):
            continue
        assert torch.isfinite(b).all() and torch.equal(a, b)
        n += 1
    assert n >= 10


def test_step_object_contract():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd.step import DynamicStep

    dev = torch.device("cuda:0")
    sc, net, nodes, st, vm, pm, fidx, t, mk, (gC, gD, gA, gV) = _setup(dev)
    bg6 = torch.ones(6, device=dev)
    r = mk()
    step = DynamicStep(r, net, nodes, st["qs"], st["sc"], st["op"], st["rgb"], bg6, n_views=4, n_frames=2)
    o = step(t, vm, pm, fidx)
    torch.autograd.backward([o["color"]], [gC])
    g_first = [p.grad.clone() for p in step.params]
    # a parameter that still holds a gradient: refused (no silent overwrite of an accumulated gradient)
    o = step(t, vm, pm, fidx)
    with pytest.raises(RuntimeError, match="still holds a gradient"):
        torch.autograd.backward([o["color"]], [gC])
    for p in step.params:
        p.grad = None
    # backward of a step whose buffers a later forward overwrote: refused
    o_old = step(t, vm, pm, fidx)
    o_new = step(t, vm, pm, fidx)
    with pytest.raises(RuntimeError, match="already overwritten"):
        torch.autograd.backward([o_old["color"]], [gC])
    torch.autograd.backward([o_new["color"]], [gC])
    for a, p in zip(g_first, step.params):
        assert torch.equal(a, p.grad)              # deterministic: the same step gives the same bits
    # grown capacities rebuild the object (new workspaces), same image
    ref = o_new["color"].clone()
    for p in step.params:
        p.grad = None
    r.capacity, r.record_capacity = 2 * r.capacity, 2 * r.record_capacity
    o = step(t, vm, pm, fidx)
    assert torch.equal(o["color"], ref) and step.key[0] == r.capacity
    # wrong dtypes / shapes are rejected before the library sees a pointer
    with pytest.raises(ValueError):
        step(t.double(), vm, pm, fidx)
    with pytest.raises(ValueError):
        step(t, vm[:2], pm, fidx)
    with pytest.raises(ValueError):
        step(t, vm, pm, fidx.long())
    with pytest.raises(ValueError):
        DynamicStep(r, net, nodes, st["qs"], st["sc"].clone().requires_grad_(True), st["op"], st["rgb"], bg6, n_views=4, n_frames=2)


def test_twenty_unit_step_equals_per_view_and_per_frame_composition():
    """The partition BASELINE.json configs[3] / SURVEY.md 8(e) name, at its size: bench.py's step object over 4 frames x (4 SDS views + 1
    reference view) = 20 (frame, view) units of the 199,980-Gaussian scene at 512 x 512, against the composition of the same work from
    smaller calls --
      * every unit ALONE through ``views.render_views`` (B = 1, the frame's node outputs): image, depth, alpha, radii and the duplicate
        count bit for bit (a unit of the batch is the unit on its own: tile order, sort, blend do not see the other 19);
      * every FRAME alone (its 5 views, one ``render_views`` call with the backward): the node-output gradients of the frame bit for bit
        (the vertex kernel adds a frame's views in view order whatever else is in the batch), and from those, through
        ``DeformationNetwork.node_outputs``' backward, every parameter gradient bit for bit.
    One unit against two oracle passes: tests/test_views_gpu.py::test_bench_scene_one_view_against_two_oracle_passes."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import bench
    from dreammesh4d_amd import views

    dev = torch.device("cuda:0")
    wl = bench.Workload(dev, 0, 1)
    assert wl.views_per_step == 20 and wl.views_per_frame == 5 and bench.FRAMES_PER_STEP == 4
    st = wl.dstep
    out = st(wl.frame_t, wl.vm16, wl.pm16, wl.fidx)
    torch.autograd.backward([out["color"], out["alpha"]], [wl.gC, wl.gA])
    D = wl.renderer.check()
    ref = {k: out[k].detach().clone() for k in ("color", "depth", "alpha", "radii", "vxyz", "vrot")}
    names = st.head_names
    ref_gnode = {k: st.gnode[k].detach().clone() for k in names}
    params = st.params
    ref_grads = [p.grad.detach().clone() for p in params]
    assert all(bool(g.any()) for g in ref_grads[-4:])
    for p in wl.net.parameters():
        p.grad = None

    # ---- composition
    wl.net.grads_in_place = False
    dx, dr, ds, do = wl.net.node_outputs(wl.nodes, wl.frame_t)
    raw = {"dx": dx, "dr": dr, "ds": ds, "do": do}
    for k in names:
        assert torch.equal(st.node_outputs()[k].view(raw[k].shape), raw[k]), k
    def renderer():
        r = views.ViewRenderer(wl.graph, wl.topo, bench.H, bench.W, wl.cams[0].tanfov, method="hybrid")
        r.fuse_face_backward = True
        return r
    # every unit alone, forward only
    r1 = renderer()
    with torch.no_grad():
        for u in range(20):
            f = int(wl.fidx[u])
            o = views.render_views(r1, dx[f:f + 1], dr[f:f + 1], ds[f:f + 1], do[f:f + 1], wl.qs, wl.scales, wl.opac, wl.rgb,
                                   wl.vm[u:u + 1], wl.pm[u:u + 1], wl.bg6)
            for k in ("color", "depth", "alpha", "radii"):
                assert torch.equal(o[k][0], ref[k][u]), (k, u)
            assert torch.equal(o["vxyz"][0], ref["vxyz"][f]) and torch.equal(o["vrot"][0], ref["vrot"][f])
            assert r1.check()[0] == D[u], u
    # every frame alone, with the backward
    g_frames = {k: [] for k in names}
    for f in range(4):
        leaves = {k: raw[k][f:f + 1].detach().clone().requires_grad_(True) for k in names}
        r5 = renderer()
        v = slice(5 * f, 5 * f + 5)
        assert [int(x) for x in wl.fidx[v]] == [f] * 5
        o = views.render_views(r5, leaves["dx"], leaves["dr"], leaves["ds"], leaves["do"], wl.qs, wl.scales, wl.opac, wl.rgb, wl.vm[v], wl.pm[v],
                               wl.bg6, frame_index=torch.zeros(5, dtype=torch.int32, device=dev))
        for k in ("color", "depth", "alpha"):
            assert torch.equal(o[k], ref[k][v]), (k, f)
        torch.autograd.backward([o["color"], o["alpha"]], [wl.gC[v], wl.gA[v]])
        for k in names:
            g_frames[k].append(leaves[k].grad)
    g_nodes = {k: torch.cat(g_frames[k], 0) for k in names}
    for k in names:
        assert torch.equal(g_nodes[k].reshape(ref_gnode[k].shape), ref_gnode[k]), k
    torch.autograd.backward([raw[k] for k in names], [g_nodes[k] for k in names])
    n = 0
    for p, want in zip(params, ref_grads):
        assert p.grad is not None and torch.equal(p.grad, want), n
        n += 1
    assert n >= 30
