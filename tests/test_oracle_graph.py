"""CPU: how far the product's deformation-graph distance (shortest EDGE path, csrc/graph.hip, pinned to oracle/graph.py's exact
Dijkstra) is from the reference's (heat-method surface geodesic of potpourri3d, dynamic_sugar.py:819-847, restated from the
paper in oracle/graph.py::heat_method_distances): the deviation as numbers, on the kind of mesh the stage runs on."""
import numpy as np

from dreammesh4d_amd import synthetic as syn
from oracle import graph as G


def test_heat_method_is_a_distance_on_a_sphere():
    v, f = syn.uv_sphere(1500, radius=0.6)
    v = np.asarray(v, np.float64)
    src = [0, 17, 400]
    d = G.heat_method_distances(v, f, src)
    for s, i in enumerate(src):
        true = 0.6 * np.arccos(np.clip((v @ v[i]) / 0.36, -1, 1))          # great-circle distance
        assert d[s, i] == 0.0 and d[s].min() > -0.02
        far = true > 0.15
        rel = np.abs(d[s][far] / true[far] - 1)
        assert rel.max() < 0.16 and rel.mean() < 0.07      # the heat method's smoothing bias with t = h^2 (worst at the UV sphere's poles)


def test_edge_path_vs_heat_method_neighbour_sets_and_weights():
    """The number DESIGN.md quotes.  Same nodes, same K: which vertices get a different set of K (+1) nearest nodes, and what
    that does to the skinning weights (which are functions of the EUCLIDEAN distances to the chosen nodes, :842-855)."""
    K, M = 4, 120
    sc = syn.mesh_bound_scene(2400, n_nodes=M, k=K, seed=3)
    verts, faces, nodes = sc["verts"], sc["faces"], sc["nodes"]
    ei, ew, _ = G.geodesic_graph(verts, faces, nodes, K)
    hi, hw, _ = G.heat_graph(verts, faces, nodes, K)
    V = len(verts)
    same_set = np.array([set(ei[i]) == set(hi[i]) for i in range(V)])
    common = np.array([len(set(ei[i]) & set(hi[i])) for i in range(V)])
    # weights as dense [V, M] rows so that differently ordered / chosen neighbours compare
    We, Wh = np.zeros((V, M)), np.zeros((V, M))
    np.put_along_axis(We, ei, ew, 1)
    np.put_along_axis(Wh, hi, hw, 1)
    l1 = np.abs(We - Wh).sum(1)
    print(f"edge-path vs heat-method graph (V={V}, M={M}, K={K}): identical neighbour sets {100 * same_set.mean():.1f} %, "
          f"mean common neighbours {common.mean():.2f} of {K}, mean L1 weight difference {l1.mean():.3f} (rows sum to 1), "
          f"rows with L1 > 0.5: {100 * (l1 > 0.5).mean():.1f} %")
    assert same_set.mean() > 0.55 and common.mean() > 3.3          # the two distances mostly pick the same nodes ...
    assert l1.mean() < 0.25                                         # ... and the skinning weights stay close on average
    assert (common >= 2).all()
