"""GPU parity: HIP skinning + face->Gaussian kernels (through the C ABI) vs oracle/skinning.py.
Forward: float32 kernel vs float64 oracle, tolerance 2e-6 absolute on positions/quaternions
(unit-scale quantities).  Backward: vs fp64 autograd of the oracle, relative 1e-4."""
import math

import numpy as np
import pytest
import torch

from dreammesh4d_amd import synthetic as syn

pytestmark = pytest.mark.gpu
D = torch.float64


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")


def _setup(n_faces, M, K, seed):
    sc = syn.mesh_bound_scene(n_faces, n_nodes=M, k=K, seed=seed)
    g = torch.Generator().manual_seed(seed)
    raw = {"dx": 0.05 * torch.randn(M, 3, generator=g), "dr": 0.15 * torch.randn(M, 4, generator=g),
           "ds": 0.05 * torch.randn(M, 6, generator=g), "do": torch.randn(M, 1, generator=g)}
    return sc, raw


def _close(a, b, rtol, name):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b).max() / (np.abs(b).max() + 1e-30)
    assert err < rtol, (name, err)


@pytest.mark.parametrize("grad_mode", ["exact", "pypose"])
@pytest.mark.parametrize("method", ["lbs", "dqs", "hybrid"])
@pytest.mark.parametrize("n_faces,M,K", [(2000, 150, 4), (600, 40, 8)])
def test_skin_vertices_parity(method, n_faces, M, K, grad_mode):
    """grad_mode "exact": fp64 autograd of the oracle's tensor ops; "pypose": the oracle's restatement of pypose's backward
    rules (oracle/skinning.py: SO3_Log / so3_Exp / SO3_Act / SO3_Mul) -- the convention the reference trains with."""
    _need_gpu()
    from dreammesh4d_amd import ops
    from oracle import skinning as sk

    dev = torch.device("cuda:0")
    sc, raw = _setup(n_faces, M, K, seed=K)
    graph = ops.DeformGraph(sc["verts"], sc["nbr_idx"], sc["nbr_w"], M, dev)
    leaves = {k: v.to(dev).requires_grad_(True) for k, v in raw.items()}
    xyz, rot = ops.skin_vertices(graph, leaves["dx"], leaves["dr"], leaves["ds"], leaves["do"].view(-1), method=method, grad_mode=grad_mode)
    # oracle in float64 on the float32 inputs
    o = {k: v.to(D).requires_grad_(True) for k, v in raw.items()}
    verts, idx = torch.tensor(sc["verts"], dtype=D), torch.tensor(sc["nbr_idx"])
    w = torch.tensor(sc["nbr_w"], dtype=D)
    trans, q, S, op = sk.node_attributes(o["dx"], o["dr"], o["ds"], o["do"])
    oxyz, orot = sk.skin_vertices(verts, idx, w, trans, q, S, op, method, grad_mode=grad_mode)
    assert np.abs(xyz.detach().cpu().numpy() - oxyz.detach().numpy()).max() < 2e-6
    assert np.abs(rot.detach().cpu().numpy() - orot.detach().numpy()).max() < 2e-6
    gen = torch.Generator().manual_seed(1)
    gx, gr = torch.randn(xyz.shape, generator=gen), torch.randn(rot.shape, generator=gen)
    torch.autograd.backward([xyz, rot], [gx.to(dev), gr.to(dev)])
    torch.autograd.backward([oxyz, orot], [gx.to(D), gr.to(D)])
    _close(leaves["dx"].grad.cpu(), o["dx"].grad, 2e-4, "dx")
    _close(leaves["dr"].grad.cpu(), o["dr"].grad, 2e-4, "dr")
    if method != "dqs":
        _close(leaves["ds"].grad.cpu(), o["ds"].grad, 2e-4, "ds")
    if method == "hybrid":
        _close(leaves["do"].grad.cpu(), o["do"].grad, 2e-4, "do")
    # deterministic backward
    g1 = leaves["dr"].grad.clone()
    for v in leaves.values():
        v.grad = None
    xyz2, rot2 = ops.skin_vertices(graph, leaves["dx"], leaves["dr"], leaves["ds"], leaves["do"].view(-1), method=method, grad_mode=grad_mode)
    torch.autograd.backward([xyz2, rot2], [gx.to(dev), gr.to(dev)])
    assert torch.equal(g1, leaves["dr"].grad)


@pytest.mark.parametrize("grad_mode", ["exact", "pypose"])
@pytest.mark.parametrize("G", [6, 1, 3, 4])
def test_face_gaussians_parity(G, grad_mode):
    _need_gpu()
    from dreammesh4d_amd import ops
    from oracle import skinning as sk

    dev = torch.device("cuda:0")
    sc, raw = _setup(1200, 80, 4, seed=3)
    verts, faces = torch.tensor(sc["verts"], dtype=D), torch.tensor(sc["faces"])
    F, V = faces.shape[0], verts.shape[0]
    N = F * G
    gen = torch.Generator().manual_seed(5)
    vxyz = (verts + 0.02 * torch.randn(V, 3, generator=gen, dtype=D)).float()
    vrot = torch.nn.functional.normalize(torch.tensor([0, 0, 0, 1.0]) + 0.2 * torch.randn(V, 4, generator=gen), dim=-1)
    cplx = torch.randn(N, 2, generator=gen, dtype=D)
    qs = sk.static_quaternions(verts, faces, cplx, n_per_face=G).float()
    topo = ops.MeshTopology(sc["faces"], V, G, dev)
    lx, lr = vxyz.to(dev).requires_grad_(True), vrot.to(dev).requires_grad_(True)
    means, rots, normals = ops.face_gaussians(topo, lx, lr, qs.to(dev), grad_mode=grad_mode)
    ox, orr = vxyz.to(D).requires_grad_(True), vrot.to(D).requires_grad_(True)
    om, oq, on = sk.face_gaussians(ox, orr, faces, qs.to(D), n_per_face=G, grad_mode=grad_mode)
    assert means.shape == (N, 3) and rots.shape == (N, 4) and normals.shape == (N, 3)
    assert np.abs(means.detach().cpu().numpy() - om.detach().numpy()).max() < 1e-6
    assert np.abs(rots.detach().cpu().numpy() - oq.detach().numpy()).max() < 2e-6
    assert np.abs(normals.detach().cpu().numpy() - on.detach().numpy()).max() < 1e-5
    gm, gq, gn = (torch.randn(N, 3, generator=gen), torch.randn(N, 4, generator=gen), torch.randn(N, 3, generator=gen))
    torch.autograd.backward([means, rots, normals], [gm.to(dev), gq.to(dev), gn.to(dev)])
    torch.autograd.backward([om, oq, on], [gm.to(D), gq.to(D), gn.to(D)])
    _close(lx.grad.cpu(), ox.grad, 2e-4, "vxyz")
    _close(lr.grad.cpu(), orr.grad, 2e-4, "vrot")


def test_identity_deformation_reproduces_static_geometry_on_gpu():
    _need_gpu()
    from dreammesh4d_amd import ops
    from oracle import skinning as sk

    dev = torch.device("cuda:0")
    sc, _ = _setup(3000, 200, 4, seed=9)
    M = 200
    graph = ops.DeformGraph(sc["verts"], sc["nbr_idx"], sc["nbr_w"], M, dev)
    z = lambda *s: torch.zeros(*s, device=dev)
    xyz, rot = ops.skin_vertices(graph, z(M, 3), z(M, 4), z(M, 6), z(M), method="hybrid")
    assert np.abs(xyz.cpu().numpy() - sc["verts"]).max() < 1e-6
    assert torch.allclose(rot.cpu(), torch.tensor([0, 0, 0, 1.0]).expand(len(sc["verts"]), 4), atol=1e-7)
    topo = ops.MeshTopology(sc["faces"], len(sc["verts"]), 6, dev)
    qs = sk.static_quaternions(torch.tensor(sc["verts"], dtype=D), torch.tensor(sc["faces"]),
                               torch.tensor(sc["complex"], dtype=D)).float()
    means, rots, normals = ops.face_gaussians(topo, xyz, rot, qs.to(dev))
    assert np.abs(rots.cpu().numpy() - qs.numpy()).max() < 1e-6


def test_full_size_200k_mesh_forward():
    """configs[3] size: 33,334 faces -> ~200k Gaussians, 1000 nodes, K = 4, hybrid."""
    _need_gpu()
    from dreammesh4d_amd import ops
    from oracle import skinning as sk

    dev = torch.device("cuda:0")
    sc, raw = _setup(33_334, 1000, 4, seed=0)
    graph = ops.DeformGraph(sc["verts"], sc["nbr_idx"], sc["nbr_w"], 1000, dev)
    xyz, rot = ops.skin_vertices(graph, raw["dx"].to(dev), raw["dr"].to(dev), raw["ds"].to(dev), raw["do"].view(-1).to(dev))
    verts, idx, w = torch.tensor(sc["verts"], dtype=D), torch.tensor(sc["nbr_idx"]), torch.tensor(sc["nbr_w"], dtype=D)
    trans, q, S, op = sk.node_attributes(raw["dx"].to(D), raw["dr"].to(D), raw["ds"].to(D), raw["do"].to(D))
    oxyz, orot = sk.skin_vertices(verts, idx, w, trans, q, S, op, "hybrid")
    assert np.abs(xyz.cpu().numpy() - oxyz.numpy()).max() < 2e-6
    faces = torch.tensor(sc["faces"])
    qs = sk.static_quaternions(verts, faces, torch.tensor(sc["complex"], dtype=D)).float()
    topo = ops.MeshTopology(sc["faces"], len(sc["verts"]), 6, dev)
    means, rots, normals = ops.face_gaussians(topo, xyz, rot, qs.to(dev))
    # stage 2 is checked on the SAME float32 vertices the kernel consumed (normals of 5 mm triangles
    # amplify the float32 rounding of the positions by |x| / edge ~ 100)
    om, oq, on = sk.face_gaussians(xyz.cpu().to(D), rot.cpu().to(D), faces, qs.to(D))
    assert means.shape[0] == sc["n_gaussians"] and abs(means.shape[0] - 200_004) < 4000
    assert np.abs(means.cpu().numpy() - om.numpy()).max() < 2e-6
    assert np.abs(rots.cpu().numpy() - oq.numpy()).max() < 4e-6
    assert np.abs(normals.cpu().numpy() - on.numpy()).max() < 1e-4


def test_pypose_matrix_gradient_convention_on_device():
    """ops.quat_xyzw_to_matrix(grad_mode="pypose") (the ARAP term's vertex rotation matrices, dynamic_sugar.py:640-655) against
    the oracle's SO3_Act rule applied to the three basis vectors."""
    _need_gpu()
    from dreammesh4d_amd import ops
    from oracle import skinning as sk

    g = torch.Generator().manual_seed(2)
    q = torch.nn.functional.normalize(torch.randn(50, 4, generator=g, dtype=D), dim=-1)
    G_ = torch.randn(50, 3, 3, generator=g, dtype=D)
    qd = q.float().cuda().requires_grad_(True)
    R = ops.quat_xyzw_to_matrix(qd, "pypose")
    R.backward(G_.float().cuda())
    qo = q.clone().requires_grad_(True)
    eye = torch.eye(3, dtype=D)
    Ro = torch.stack([sk._ActPP.apply(qo, eye[j].expand(50, 3)) for j in range(3)], dim=-1)
    assert np.abs(R.detach().cpu().numpy() - Ro.detach().numpy()).max() < 1e-6
    Ro.backward(G_)
    _close(qd.grad.cpu(), qo.grad, 1e-5, "pypose matrix gradient")
    assert not qd.grad[:, 3].any()                                        # tangent gradient padded with a zero
