"""GPU parity of the deformation-graph construction against oracle/graph.py: the heat-method distance of the shipped
`dist_mode: geodisc` (csrc/heat.hip: batched conjugate gradients + a float64 GEMM) against the scipy restatement of the same
published algorithm, and the edge-path alternative (csrc/graph.hip) against an exact Dijkstra on the same edge graph; the
reference's weight formula (dynamic_sugar.py:838-861) in both."""
import numpy as np
import pytest
import torch

from dreammesh4d_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")


@pytest.mark.parametrize("n_faces,M,K", [(800, 30, 4), (6000, 150, 6)])
def test_geodesic_graph_matches_exact_dijkstra(n_faces, M, K):
    _need_gpu()
    from dreammesh4d_amd.graph_build import build_deformation_graph
    from oracle import graph as G

    sc = syn.mesh_bound_scene(n_faces, n_nodes=M, k=K, seed=1)
    verts, faces, nodes = sc["verts"], sc["faces"], sc["nodes"]
    idx, w = build_deformation_graph(verts, faces, nodes, K, "geodisc", "cuda:0", geodesic="edgepath")
    oi, ow, d = G.geodesic_graph(verts, faces, nodes, K)
    idx, w = idx.cpu().numpy(), w.cpu().numpy()
    assert idx.shape == (len(verts), K) and idx.min() >= 0 and idx.max() < M
    assert np.abs(w.sum(1) - 1).max() < 1e-5 and (w >= 0).all()
    # float32 path sums vs float64: neighbours may swap only where two distances tie to rounding
    V = len(verts)
    dsel = np.take_along_axis(d.T, idx, 1)
    dora = np.take_along_axis(d.T, oi, 1)
    assert np.abs(dsel - dora).max() < 1e-5 * max(1.0, dora.max())
    # (the seeded test mesh is a regular sphere: many path lengths agree to the last bits, so a few per cent of
    # the rows order two equidistant nodes differently; their distance multisets are equal, asserted above)
    same = (idx == oi).all(1)
    assert same.mean() > 0.9
    # rows whose K + 1 nearest nodes are separated by clear gaps must reproduce the weights
    ds = np.sort(d.T, axis=1)[:, :K + 2]
    clear = same & ((ds[:, 1:] - ds[:, :-1]).min(1) > 1e-4)
    assert clear.mean() > 0.5
    assert np.abs(w[clear] - ow[clear]).max() < 2e-5
    # vertices that coincide with nothing special still get K DISTINCT nodes
    assert all(len(set(r)) == K for r in idx.tolist())


def test_the_graph_drives_the_skinning_path():
    """The tables come out in the format ops.DeformGraph consumes (int64 [V,K], float32 [V,K] rows summing to 1)."""
    _need_gpu()
    from dreammesh4d_amd import ops
    from dreammesh4d_amd.graph_build import build_deformation_graph

    dev = torch.device("cuda:0")
    sc = syn.mesh_bound_scene(1000, n_nodes=40, k=4, seed=2)
    for mode in ("geodisc", "eucdisc"):
        idx, w = build_deformation_graph(sc["verts"], sc["faces"], sc["nodes"], 4, mode, dev)
        assert idx.dtype == torch.int64 and w.dtype == torch.float32
        graph = ops.DeformGraph(sc["verts"], idx.cpu().numpy(), w.cpu().numpy(), 40, dev)
        z = torch.zeros(40, 3, device=dev)
        x, q = ops.skin_vertices(graph, z, torch.zeros(40, 4, device=dev), torch.zeros(40, 6, device=dev),
                                 torch.zeros(40, device=dev), "hybrid")
        assert (x - torch.tensor(sc["verts"], device=dev)).abs().max() < 1e-5      # identity deformation


@pytest.mark.parametrize("solver", ["dense", "cg"])
@pytest.mark.parametrize("n_faces,M,K", [(2400, 120, 4), (800, 30, 6)])
def test_heat_method_graph_matches_the_scipy_restatement(n_faces, M, K, solver):
    """`dist_mode: geodisc` as shipped: for every vertex the heat-method distances to the nodes' nearest vertices, the K + 1
    nearest nodes, Euclidean weights (dynamic_sugar.py:819-861).  Oracle: sparse LU per system (oracle/graph.py::heat_graph);
    product: M Poisson + V heat systems on the device -- dense float64 Cholesky + GEMMs (what `auto` picks up to 40 k vertices)
    or batched conjugate gradients (csrc/heat.hip; larger meshes).  >= 99 % identical neighbour sets (the rest: two nodes
    whose distances tie to solver precision), weight rows within 1e-3 (L1) on the identical rows."""
    _need_gpu()
    from dreammesh4d_amd.graph_build import build_deformation_graph, heat_geodesic_knn
    from oracle import graph as G

    sc = syn.mesh_bound_scene(n_faces, n_nodes=M, k=K, seed=3)
    verts, faces, nodes = sc["verts"], sc["faces"], sc["nodes"]
    stats = {}
    idx, w = heat_geodesic_knn(verts, faces, nodes, K, "cuda:0", chunk=500, stats=stats, solver=solver)       # (several source chunks)
    assert stats["solver"] == solver
    idx2, w2 = build_deformation_graph(verts, faces, nodes, K, "geodisc", "cuda:0")            # the shipped mode IS the heat method
    # (another chunking stops the conjugate gradients at other iteration counts: solver-precision ties may order differently)
    assert float((idx.sort(1).values == idx2.sort(1).values).all(1).float().mean()) > 0.995
    idx3, w3 = build_deformation_graph(verts, faces, nodes, K, "geodisc", "cuda:0")
    assert torch.equal(idx2, idx3) and torch.equal(w2, w3)                                     # bit-reproducible (every rank builds the same graph)
    oi, ow, d = G.heat_graph(verts, faces, nodes, K)
    idx, w = idx.cpu().numpy(), w.cpu().numpy()
    V = len(verts)
    assert idx.shape == (V, K) and idx.min() >= 0 and idx.max() < M and np.abs(w.sum(1) - 1).max() < 1e-5
    same_set = np.array([set(a) == set(b) for a, b in zip(idx.tolist(), oi.tolist())])
    same_order = (idx == oi).all(1)
    print(f"heat graph V={V} M={M} K={K}: identical sets {same_set.mean():.4f}, identical order {same_order.mean():.4f}, "
          f"Poisson CG {stats['poisson_iterations']} it, heat CG {stats['heat_iterations']} it")
    assert same_set.mean() >= 0.99
    # where the sets differ, the swapped nodes are equidistant to solver precision in the ORACLE's own table
    for i in np.nonzero(~same_set)[0]:
        mine, ref = d[i][idx[i]], d[i][oi[i]]
        assert abs(np.sort(mine)[-1] - np.sort(ref)[-1]) < 2e-5 * max(1.0, np.abs(d[i]).max())     # CG (1e-10) against sparse LU
    # the weights also depend on the (K + 1)-th node (the normalising distance): compare where the oracle's own ranking is clear
    ds = np.sort(d, axis=1)[:, :K + 2]
    clear = same_order & ((ds[:, 1:] - ds[:, :-1]).min(1) > 2e-5)          # (the regular test sphere has many symmetric near-ties)
    assert clear.mean() > 0.8, clear.mean()
    l1 = np.abs(w[clear] - ow[clear]).sum(1)
    assert l1.max() < 1e-3 and all(len(set(r)) == K for r in idx.tolist())


def test_dense_heat_solver_at_the_bench_scale_against_sparse_lu():
    """16.7k vertices, 1000 nodes: the heat solution spans ~57 decades across the mesh and only the dense float64 factorisation
    keeps its far field (graph_build.py); 120 random source vertices through the sparse-LU oracle."""
    _need_gpu()
    from dreammesh4d_amd.graph_build import build_deformation_graph
    from oracle import graph as G

    K = 4
    sc = syn.mesh_bound_scene(33334, n_nodes=1000, k=K, seed=0)
    v, f, n = np.asarray(sc["verts"], np.float64), np.asarray(sc["faces"]), np.asarray(sc["nodes"], np.float64)
    sub = np.sort(np.random.default_rng(0).choice(len(v), 120, replace=False))
    node_vertex = np.array([np.argmin(np.linalg.norm(v - p, axis=1)) for p in n])
    d = G.heat_method_distances(v, f, sub)[:, node_vertex]
    o_idx = np.argsort(d, axis=1)[:, :K]
    idx, w = build_deformation_graph(sc["verts"], sc["faces"], sc["nodes"], K, "geodisc", "cuda:0")
    idx = idx.cpu().numpy()[sub]
    same = np.array([set(a) == set(b) for a, b in zip(idx.tolist(), o_idx.tolist())])
    assert same.mean() >= 0.97, same.mean()
    for i in np.nonzero(~same)[0]:           # the rest: ties in the oracle's own table
        assert abs(np.sort(d[i][idx[i]])[-1] - np.sort(d[i][o_idx[i]])[-1]) <= 1e-6 * max(1.0, np.abs(d[i]).max())


def test_heat_graph_at_cfg5_scale_against_sparse_lu():
    """BASELINE configs[4]'s mesh: 166,667 faces -> 83.3k vertices, 1000 nodes (custom/threestudio-dreammesh4d/geometry/
    dynamic_sugar.py:745-861 solves per vertex at any size; round 3 refused above 65,536 vertices).  The blocked in-place dense
    float64 solver (graph_build.py: ~125 GB at the peak) against the sparse-LU restatement of the heat method on 200 random source
    vertices: >= 98 % identical neighbour sets, the rest ties in the oracle's own table; and within the time bar of 30 s + the
    host-side assembly."""
    _need_gpu()
    import time

    from dreammesh4d_amd.graph_build import DENSE_MAX_VERTICES, heat_geodesic_knn
    from oracle import graph as G

    if torch.cuda.get_device_properties(0).total_memory < 200e9:
        pytest.skip("needs ~125 GB of device memory")
    K = 4
    sc = syn.mesh_bound_scene(166667, n_nodes=1000, k=K, seed=0)
    v, f, n = np.asarray(sc["verts"], np.float64), np.asarray(sc["faces"]), np.asarray(sc["nodes"], np.float64)
    assert 80_000 < len(v) < DENSE_MAX_VERTICES
    stats = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx, w = heat_geodesic_knn(sc["verts"], sc["faces"], sc["nodes"], K, "cuda:0", stats=stats)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"heat graph at V = {len(v)}: {dt:.1f} s, stages {({k: v_ for k, v_ in stats.items() if k.startswith('t_')})}, "
          f"peak {torch.cuda.max_memory_allocated() / 1e9:.0f} GB")
    assert stats["solver"] == "dense"
    sub = np.sort(np.random.default_rng(0).choice(len(v), 200, replace=False))
    node_vertex = np.array([np.argmin(np.linalg.norm(v - p, axis=1)) for p in n])
    d = G.heat_method_distances(v, f, sub)[:, node_vertex]
    o_idx = np.argsort(d, axis=1)[:, :K]
    got = idx.cpu().numpy()[sub]
    same = np.array([set(a) == set(b) for a, b in zip(got.tolist(), o_idx.tolist())])
    assert same.mean() >= 0.98, same.mean()
    for i in np.nonzero(~same)[0]:           # the rest: ties in the oracle's own table
        assert abs(np.sort(d[i][got[i]])[-1] - np.sort(d[i][o_idx[i]])[-1]) <= 1e-6 * max(1.0, np.abs(d[i]).max())
    wn = w.cpu().numpy()
    assert np.isfinite(wn).all() and np.abs(wn.sum(1) - 1.0).max() < 1e-5 and all(len(set(r)) == K for r in got.tolist())
    assert dt < 60.0, dt            # (the bar is 30 s of device work; the scipy assembly of the operators on the host is in dt too)
