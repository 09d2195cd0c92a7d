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
