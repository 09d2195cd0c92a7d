"""GPU parity of the ARAP energy kernels (csrc/meshreg.hip) against the oracle (oracle/mesh_reg.py, itself pinned to
the reference's ARAPCoach by tests/golden/arap_small.npz) and against the golden directly."""
import os

import numpy as np
import pytest
import torch

from dreammesh4d_amd import synthetic as syn

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "arap_small.npz")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")


def test_golden_energy_and_gradients():
    _need_gpu()
    from dreammesh4d_amd.mesh_reg import ARAPCoach

    dev = torch.device("cuda:0")
    g = np.load(GOLD)
    coach = ARAPCoach(g["verts"], g["faces"], dev)
    xyz = torch.tensor(g["xyz_prime"], device=dev, requires_grad=True)
    R = torch.tensor(g["rotations"], device=dev, requires_grad=True)
    E = coach.compute_arap_energy(xyz, R)
    assert abs(float(E) - float(g["energy"])) <= 5e-6 * abs(float(g["energy"]))
    E.backward()
    assert np.abs(xyz.grad.cpu().numpy() - g["g_xyz"]).max() <= 2e-5 * np.abs(g["g_xyz"]).max()
    assert np.abs(R.grad.cpu().numpy() - g["g_rot"]).max() <= 2e-5 * np.abs(g["g_rot"]).max()


def test_batched_timestamps_against_the_oracle_on_a_large_mesh():
    _need_gpu()
    from dreammesh4d_amd.mesh_reg import ARAPCoach
    from oracle import mesh_reg as M

    dev = torch.device("cuda:0")
    verts, faces = syn.uv_sphere(6000, radius=0.6)
    coach = ARAPCoach(verts, faces, dev)
    adj = M.build(verts, faces)
    assert np.allclose(coach.edge_weights, adj["w"], rtol=1e-6, atol=1e-7)
    T, V = 3, len(verts)
    gen = torch.Generator().manual_seed(4)
    xyz = torch.tensor(verts, dtype=torch.float32)[None] + 0.02 * torch.randn(T, V, 3, generator=gen)
    q = torch.nn.functional.normalize(torch.cat([0.1 * torch.randn(T, V, 3, generator=gen), torch.ones(T, V, 1)], -1), dim=-1)
    x, y, z, w = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w),
                     1 - 2 * (x * x + z * z), 2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w),
                     1 - 2 * (x * x + y * y)], -1).reshape(T, V, 3, 3)
    xc, Rc = xyz.clone().requires_grad_(True), R.clone().requires_grad_(True)
    wts = torch.tensor([1.0, -0.5, 2.0])
    Eo = torch.stack([M.arap_energy(adj, xc[t], Rc[t]) for t in range(T)])
    (Eo * wts).sum().backward()
    xg, Rg = xyz.to(dev).requires_grad_(True), R.to(dev).requires_grad_(True)
    Eh = coach.compute_arap_energy(xg, Rg)
    assert Eh.shape == (T,)
    assert (Eh.cpu() - Eo.detach()).abs().max() <= 1e-4 * Eo.detach().abs().max()
    (Eh * wts.to(dev)).sum().backward()
    assert (xg.grad.cpu() - xc.grad).abs().max() <= 1e-4 * xc.grad.abs().max()
    assert (Rg.grad.cpu() - Rc.grad).abs().max() <= 1e-4 * Rc.grad.abs().max()
    # deterministic
    xg2, Rg2 = xyz.to(dev).requires_grad_(True), R.to(dev).requires_grad_(True)
    (coach.compute_arap_energy(xg2, Rg2) * wts.to(dev)).sum().backward()
    assert torch.equal(xg.grad, xg2.grad) and torch.equal(Rg.grad, Rg2.grad)


def test_arap_through_the_geometry_accessors():
    """The system's `_compute_arap_energy`: deformed vertices + rotation matrices of the timestamps -> energy; identity
    deformation costs nothing and gradients reach the deformation network."""
    _need_gpu()
    from dreammesh4d_amd import sugar
    from dreammesh4d_amd.mesh_reg import ARAPCoach

    dev = torch.device("cuda:0")
    sc = syn.mesh_bound_scene(1200, n_nodes=60, k=4, seed=2)
    geo = sugar.DynamicSuGaR(sc["verts"], sc["faces"], sc["nodes"], sc["nbr_idx"], sc["nbr_w"],
                             deformation_kwargs=dict(resolution=(16, 16, 16, 9), multires=(1, 2)), device=dev)
    coach = ARAPCoach(geo.get_xyz_verts, geo.get_faces, dev)
    ts = torch.tensor([0.25, 0.6], device=dev)
    E0 = coach.compute_arap_energy(geo.get_timed_vertex_xyz(ts), geo.get_timed_vertex_rotation(ts, return_matrix=True))
    assert float(E0.sum()) < 1e-6                       # zero-initialised heads: rest pose
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for n, p in geo._deformation.named_parameters():
            if "_deform" in n:
                p.add_((0.05 * torch.randn(p.shape, generator=g)).to(dev))
    E = coach.compute_arap_energy(geo.get_timed_vertex_xyz(ts), geo.get_timed_vertex_rotation(ts, return_matrix=True))
    assert float(E.sum()) > 1e-4
    E.sum().backward()
    gr = [p.grad for n, p in geo._deformation.named_parameters() if "pos_deform" in n]
    assert all(v is not None and torch.isfinite(v).all() for v in gr) and sum(float(v.abs().sum()) for v in gr) > 0


def test_normal_consistency_against_the_oracle():
    """csrc/meshreg.hip::k_nc_* (MeshNormalConsistency) vs oracle/mesh_reg.py::normal_consistency (float64 autograd)
    on T deformed copies of a closed mesh plus a non-manifold fan; value and vertex gradients."""
    _need_gpu()
    from dreammesh4d_amd.mesh_reg import MeshNormalConsistency
    from oracle import mesh_reg as M

    dev = torch.device("cuda:0")
    sc = syn.mesh_bound_scene(1200, n_nodes=40, k=4, seed=3)
    verts, faces = np.asarray(sc["verts"], np.float64), np.asarray(sc["faces"], np.int64)
    V = len(verts)
    faces = np.concatenate([faces, [[0, 1, V - 1], [0, 1, V - 2]]])          # extra faces on one edge: 3+ faces share it
    g = torch.Generator().manual_seed(0)
    T = 3
    x64 = (torch.tensor(verts)[None] + 0.02 * torch.randn(T, V, 3, dtype=torch.float64, generator=g)).requires_grad_(True)
    want = M.normal_consistency(x64, faces)
    want.backward()
    nc = MeshNormalConsistency(faces, V, dev)
    assert nc.n_pairs == len(M.normal_consistency_pairs(faces)) > 1.5 * 1200 - 10
    assert np.array_equal(nc._pairs.cpu().numpy(), M.normal_consistency_pairs(faces))
    x = x64.detach().float().to(dev).requires_grad_(True)
    got = nc(x)
    got.backward()
    assert abs(float(got) - float(want)) < 2e-6 * max(1.0, abs(float(want)))
    gw = x64.grad.numpy()
    assert np.abs(x.grad.cpu().numpy() - gw).max() < 2e-4 * np.abs(gw).max()
    # deterministic (gather backward)
    x2 = x.detach().clone().requires_grad_(True)
    nc(x2).backward()
    assert torch.equal(x.grad, x2.grad)
    # a single [V,3] mesh is the T = 1 batch
    assert abs(float(nc(x.detach()[0])) - float(M.normal_consistency(x64.detach()[:1], faces))) < 2e-6


def test_laplacian_smoothing_against_the_oracle():
    _need_gpu()
    from dreammesh4d_amd.mesh_reg import MeshLaplacianSmoothing
    from oracle import mesh_reg as M

    dev = torch.device("cuda:0")
    sc = syn.mesh_bound_scene(900, n_nodes=30, k=4, seed=4)
    verts, faces = np.asarray(sc["verts"], np.float64), np.asarray(sc["faces"], np.int64)
    V = len(verts)
    g = torch.Generator().manual_seed(0)
    x64 = (torch.tensor(verts)[None] + 0.03 * torch.randn(3, V, 3, dtype=torch.float64, generator=g)).requires_grad_(True)
    want = M.laplacian_smoothing(x64, faces)
    want.backward()
    ls = MeshLaplacianSmoothing(faces, V, dev)
    x = x64.detach().float().to(dev).requires_grad_(True)
    got = ls(x)
    got.backward()
    assert abs(float(got) - float(want)) < 2e-6 * max(1.0, abs(float(want)))
    gw = x64.grad.numpy()
    assert np.abs(x.grad.cpu().numpy() - gw).max() < 2e-4 * np.abs(gw).max()
    x2 = x.detach().clone().requires_grad_(True)
    ls(x2).backward()
    assert torch.equal(x.grad, x2.grad)


def test_quat_to_matrix_hip_equals_the_torch_expression():
    """ops.quat_xyzw_to_matrix (pypose convention) on the HIP device: csrc/meshreg.hip::k_quat_matrix_fwd / _bwd against the torch
    expression the CPU path evaluates (dynamic_sugar.py:640-655: pypose SO3.matrix(); backward (sum_i R e_i x G[:, i], 0))."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import ops

    g = torch.Generator().manual_seed(5)
    q = torch.nn.functional.normalize(torch.randn(3, 1001, 4, generator=g), dim=-1)
    G = torch.randn(3, 1001, 3, 3, generator=g)
    qc = q.clone().requires_grad_(True)
    Rc = ops.quat_xyzw_to_matrix(qc, "pypose")
    Rc.backward(G)
    qd = q.to("cuda:0").requires_grad_(True)
    Rd = ops.quat_xyzw_to_matrix(qd, "pypose")
    Rd.backward(G.to("cuda:0"))
    assert Rd.shape == (3, 1001, 3, 3)
    assert torch.equal(Rd.detach().cpu(), Rc.detach())                      # the same float32 operations in the same order
    assert float((qd.grad.cpu() - qc.grad).abs().max()) <= 1e-6 * float(qc.grad.abs().max())
    assert float(qd.grad[..., 3].abs().max()) == 0.0
