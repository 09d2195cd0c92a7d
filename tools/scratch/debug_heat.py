import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from dreammesh4d_amd import synthetic as syn
from dreammesh4d_amd.graph_build import heat_geodesic_knn
from oracle import graph as G
K=4
sc = syn.mesh_bound_scene(2400, n_nodes=120, k=K, seed=3)
verts, faces, nodes = sc["verts"], sc["faces"], sc["nodes"]
idx, w = heat_geodesic_knn(verts, faces, nodes, K, "cuda:0", chunk=500)
idx5, w5 = heat_geodesic_knn(verts, faces, nodes, K+1, "cuda:0", chunk=500)
oi, ow, d = G.heat_graph(verts, faces, nodes, K)
idx, w, idx5 = idx.cpu().numpy(), w.cpu().numpy(), idx5.cpu().numpy()
l1 = np.abs(w-ow).sum(1)
same = (idx==oi).all(1)
bad = np.nonzero(same & (l1>1e-3))[0]
print("rows same order but weights differ:", len(bad))
for i in bad[:8]:
    order = np.argsort(d[i])[:K+3]
    print(i, "mine", idx5[i], "oracle", order, "oracle d", np.round(d[i][order],6), "d of mine", np.round(d[i][idx5[i]],6), "l1", l1[i])
nv = np.array([np.argmin(np.linalg.norm(np.asarray(verts,np.float64) - p, axis=1)) for p in np.asarray(nodes,np.float64)])
print("duplicate node vertices:", len(nv) - len(set(nv.tolist())))
ds = np.sort(d, axis=1)[:, :K + 2]
clear = same & ((ds[:, 1:] - ds[:, :-1]).min(1) > 2e-5)
worst = np.argsort(-(l1 * clear))[:5]
for i in worst:
    order = np.argsort(d[i])[:K+3]
    print("WORST", i, "mine", idx5[i], "oracle", order, "oracle d", np.round(d[i][order],6), "d of mine", np.round(d[i][idx5[i]],6), "l1", l1[i], "w", w[i], ow[i])
    v = np.asarray(verts, np.float64)[i]; n = np.asarray(nodes, np.float64)
    print("   eucl mine", np.linalg.norm(v - n[idx5[i]], axis=1), "eucl oracle", np.linalg.norm(v - n[order[:K+1]], axis=1))
