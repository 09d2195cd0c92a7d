import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from dreammesh4d_amd import synthetic as syn, _lib
from dreammesh4d_amd import graph_build as gb
sc = syn.mesh_bound_scene(2400, n_nodes=120, k=4, seed=3)
verts, faces = np.asarray(sc["verts"], np.float32), np.asarray(sc["faces"])
Lm, Hm, area, t, G, D = gb.heat_operators(verts, faces)
V = len(verts); dev = torch.device("cuda:0"); L_ = _lib.lib()
import scipy.sparse.linalg as spl
lu = spl.splu(Hm.tocsc())
def solve(s0, S):
    Bh = torch.zeros(V, S, dtype=torch.float64, device=dev)
    Bh[torch.arange(s0, s0 + S, device=dev), torch.arange(S, device=dev)] = 1.0
    U, it = gb._cg(L_, Hm, Bh, torch.zeros(V, S, dtype=torch.float64, device=dev), 2000, 1e-10, 10, "heat")
    return U.cpu().numpy(), it
for s0, S in ((500, 500), (0, V), (960, 64), (977, 1)):
    U, it = solve(s0, S)
    j = 977 - s0
    rhs = np.zeros(V); rhs[977] = 1
    ref = lu.solve(rhs)
    print(s0, S, "it", it, "max rel err of column 977:", np.abs(U[:, j] - ref).max() / np.abs(ref).max(), "col sum check", np.abs(Hm @ U[:, j] - rhs).max())
