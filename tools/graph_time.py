"""Start-up cost of the deformation graph at the bench scale (16.7k vertices, 1000 nodes, K = 4): heat-method distances
(csrc/heat.hip, the shipped `dist_mode: geodisc`) against the edge-path relaxation (csrc/graph.hip); and how far apart
the two graphs are."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from dreammesh4d_amd import synthetic as syn
from dreammesh4d_amd.graph_build import build_deformation_graph, heat_geodesic_knn

n_faces = int(sys.argv[1]) if len(sys.argv) > 1 else 33334
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
sc = syn.mesh_bound_scene(n_faces, n_nodes=M, k=4, seed=0)
V = len(sc["verts"])
for name, fn in (("heat", lambda st: heat_geodesic_knn(sc["verts"], sc["faces"], sc["nodes"], 4, "cuda:0", stats=st)),
                 ("heat (cg)", lambda st: heat_geodesic_knn(sc["verts"], sc["faces"], sc["nodes"], 4, "cuda:0", stats=st, solver="cg")),
                 ("edgepath", lambda st: build_deformation_graph(sc["verts"], sc["faces"], sc["nodes"], 4, "geodisc", "cuda:0", geodesic="edgepath"))):
    res = None
    for rep in range(2):
        st = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = fn(st)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:9s} V={V} M={M} K=4: {dt:.3f} s (second call) {st}")
    if name == "heat": heat = res
    elif name == "heat (cg)":
        print(f"identical neighbour sets dense vs cg: {float((heat[0].sort(1).values == res[0].sort(1).values).all(1).float().mean()):.4f}, "
              f"max |weight difference| {float((heat[1] - res[1]).abs().max()):.2e}")
    else:
        same = (heat[0].sort(1).values == res[0].sort(1).values).all(1).float().mean()
        print(f"identical neighbour sets heat vs edge path: {float(same):.4f}")
