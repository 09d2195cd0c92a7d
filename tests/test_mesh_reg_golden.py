"""The ARAP oracle (oracle/mesh_reg.py) against the vectors the reference's own ARAPCoach produced
(tests/golden/arap_small.npz, generator tests/golden/make_golden.py::arap)."""
import os

import numpy as np
import torch

from oracle import mesh_reg as M

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "arap_small.npz")


def test_arap_energy_and_gradients_match_the_reference():
    g = np.load(GOLD)
    adj = M.build(g["verts"], g["faces"])
    xyz = torch.tensor(g["xyz_prime"], requires_grad=True)
    R = torch.tensor(g["rotations"], requires_grad=True)
    E = M.arap_energy(adj, xyz, R)
    assert abs(float(E) - float(g["energy"])) <= 2e-6 * abs(float(g["energy"]))
    E.backward()
    assert np.abs(xyz.grad.numpy() - g["g_xyz"]).max() <= 2e-5 * np.abs(g["g_xyz"]).max()
    assert np.abs(R.grad.numpy() - g["g_rot"]).max() <= 2e-5 * np.abs(g["g_rot"]).max()
    # rigid translation costs nothing (closed-form pin of SURVEY.md section 8c)
    V = len(g["verts"])
    E0 = M.arap_energy(adj, torch.tensor(g["verts"]) + 0.3, torch.eye(3)[None].repeat(V, 1, 1))
    assert float(E0) < 1e-9 and float(g["energy_rigid"]) < 1e-9


def test_adjacency_is_symmetric_and_reverse_edges_are_consistent():
    g = np.load(GOLD)
    adj = M.build(g["verts"], g["faces"])
    src, nbr, rev = adj["src"], adj["nbr"], adj["rev"]
    assert np.array_equal(src[rev], nbr) and np.array_equal(nbr[rev], src) and np.array_equal(rev[rev], np.arange(len(rev)))
    assert np.allclose(adj["w"][rev], adj["w"])                 # W + W^T is symmetric
    assert np.allclose(adj["e"][rev], -adj["e"])


def test_normal_consistency_oracle_closed_forms():
    """oracle/mesh_reg.py::normal_consistency restates pytorch3d.loss.mesh_normal_consistency (parity unpinned: the
    package is neither vendored by the reference nor installed); these are the cases with a known answer."""
    import itertools

    from oracle import mesh_reg as M

    # two triangles over the edge (0, 1): flat -> 0, folded by 90 degrees -> 1, folded flat onto each other -> 2
    faces = np.array([[0, 1, 2], [1, 0, 3]])
    for v3, want in (((0, -1, 0), 0.0), ((0, 0, 1), 1.0), ((0, 1, 0), 2.0)):
        v = torch.tensor([[[0, 0, 0], [1, 0, 0], [0, 1, 0], v3]], dtype=torch.float64)
        assert abs(float(M.normal_consistency(v, faces)) - want) < 1e-12
    # a cube (12 triangles, outward orientation): 18 edges = 6 face diagonals (coplanar, 0) + 12 cube edges (90 deg, 1)
    corners = np.array(list(itertools.product((0, 1), repeat=3)), np.float64)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    cube = np.array([t for a, b, c, d in quads for t in ((a, b, c), (a, c, d))])
    pr = M.normal_consistency_pairs(cube)
    assert pr.shape == (18, 4) and (pr[:, 0] < pr[:, 1]).all()
    v = torch.tensor(corners[None])
    assert abs(float(M.normal_consistency(v, cube)) - 12.0 / 18.0) < 1e-12
    # invariant under rigid motion and uniform scale, batched mean over meshes
    R = torch.linalg.qr(torch.randn(3, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(0)))[0]
    v2 = torch.cat([v, 2.5 * v @ R.T + 0.3])
    assert abs(float(M.normal_consistency(v2, cube)) - 12.0 / 18.0) < 1e-12
    # an edge shared by three faces contributes all three pairs
    fan = np.array([[0, 1, 2], [0, 1, 3], [0, 1, 4]])
    assert M.normal_consistency_pairs(fan).shape == (3, 4)


def test_laplacian_smoothing_oracle_closed_forms():
    """oracle/mesh_reg.py::laplacian_smoothing (pytorch3d's uniform Laplacian loss, parity unpinned): known answers."""
    from oracle import mesh_reg as M

    # a regular hexagon fan: the centre is the mean of its ring (term 0); a rim vertex has neighbours centre + two rim
    ang = np.arange(6) * np.pi / 3
    v = np.concatenate([[[0, 0, 0]], np.stack([np.cos(ang), np.sin(ang), np.zeros(6)], 1)])
    faces = np.array([[0, 1 + k, 1 + (k + 1) % 6] for k in range(6)])
    rim = np.linalg.norm((v[0] + v[2] + v[6]) / 3 - v[1])
    want = 6 * rim / 7
    got = float(M.laplacian_smoothing(torch.tensor(v[None]), faces))
    assert abs(got - want) < 1e-12
    # lifting the centre by h adds h to its own term
    v2 = v.copy(); v2[0, 2] = 0.5
    terms_rim = np.linalg.norm((v2[0] + v[2] + v[6]) / 3 - v[1])
    assert abs(float(M.laplacian_smoothing(torch.tensor(v2[None]), faces)) - (0.5 + 6 * terms_rim) / 7) < 1e-12
    # translation invariant, scales linearly, batched mean
    b = torch.tensor(np.stack([v, 3.0 * v + 1.0]))
    assert abs(float(M.laplacian_smoothing(b, faces)) - (want + 3 * want) / 2) < 1e-12
