"""At the bench scale (16.7k vertices, 1000 nodes) which solver of the heat-method graph build agrees with the sparse-LU restatement
(oracle/graph.py::heat_method_distances, the reference's direct solver standing in)?  A random subset of source vertices through
the oracle on the host, against the dense float64 solver and the batched conjugate gradients of the product.
(Test infrastructure: imports oracle/ -- run by hand, not part of the product.)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from dreammesh4d_amd import synthetic as syn
from dreammesh4d_amd.graph_build import heat_geodesic_knn
from oracle import graph as G

n_sub = int(sys.argv[1]) if len(sys.argv) > 1 else 600
n_faces = int(sys.argv[2]) if len(sys.argv) > 2 else 33334           # 166667: BASELINE cfg 5's mesh (83.3k vertices)
solvers = sys.argv[3].split(",") if len(sys.argv) > 3 else ["dense", "cg"]
K = 4
sc = syn.mesh_bound_scene(n_faces, n_nodes=1000, k=K, seed=0)
v, f, n = np.asarray(sc["verts"], np.float64), np.asarray(sc["faces"]), np.asarray(sc["nodes"], np.float64)
V = len(v)
sub = np.sort(np.random.default_rng(0).choice(V, n_sub, replace=False))
node_vertex = np.array([np.argmin(np.linalg.norm(v - p, axis=1)) for p in n])
t0 = time.perf_counter()
d = G.heat_method_distances(v, f, sub)[:, node_vertex]                   # [n_sub, M]
print(f"oracle: {n_sub} sources in {time.perf_counter() - t0:.1f} s")
o_idx = np.argsort(d, axis=1)[:, :K]
for solver in solvers:
    stats = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx, w = heat_geodesic_knn(sc["verts"], sc["faces"], sc["nodes"], K, "cuda:0", solver=solver, stats=stats)
    torch.cuda.synchronize()
    print(f"{solver}: V = {V}, {time.perf_counter() - t0:.2f} s, peak memory {torch.cuda.max_memory_allocated() / 1e9:.1f} GB, stages {({k: v for k, v in stats.items() if k.startswith('t_')})}")
    idx = idx.cpu().numpy()[sub]
    same = np.array([set(a) == set(b) for a, b in zip(idx.tolist(), o_idx.tolist())])
    # where they differ: how far apart (in the oracle's own distances) are the swapped nodes?
    gap = [abs(np.sort(d[i][idx[i]])[-1] - np.sort(d[i][o_idx[i]])[-1]) / max(1e-30, np.sort(d[i][o_idx[i]])[-1]) for i in np.nonzero(~same)[0]]
    print(f"{solver:5s}: identical neighbour sets on {same.mean():.4f} of {n_sub} vertices; relative distance gap of the swapped node where they differ: "
          f"median {np.median(gap) if gap else 0:.2e}, max {max(gap) if gap else 0:.2e}")
