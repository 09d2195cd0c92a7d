"""How much the ORDER of the mesh faces (= of the Gaussians in memory) matters to the step: the bench scene as generated (UV-sphere
rings), with the faces sorted along a Morton curve of their centroids, and randomly shuffled."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from dreammesh4d_amd import synthetic as syn

orig = syn.mesh_bound_scene

def morton_order(c, bits=10):
    lo, hi = c.min(0), c.max(0)
    q = np.minimum(((c - lo) / (hi - lo + 1e-12) * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    code = np.zeros(len(c), np.int64)
    for b in range(bits):
        for k in range(3):
            code |= ((q[:, k] >> b) & 1) << (3 * b + k)
    return np.argsort(code, kind="stable")

def permuted(kind):
    def f(*a, **kw):
        sc = orig(*a, **kw)
        F = len(sc["faces"])
        if kind == "as generated":
            return sc
        cen = sc["verts"][sc["faces"]].mean(1)
        perm = morton_order(cen) if kind == "morton" else np.random.default_rng(0).permutation(F)
        g = (perm[:, None] * 6 + np.arange(6)[None]).reshape(-1)
        sc = dict(sc)
        sc["faces"] = sc["faces"][perm]
        for k in ("log_scales", "complex", "densities", "sh_dc"):
            sc[k] = sc[k][g]
        return sc
    return f

dev = torch.device('cuda:0')
for kind in ("as generated", "morton", "random", "as generated", "morton"):
    syn.mesh_bound_scene = permuted(kind)
    wl = bench.Workload(dev, 0, 1)
    for _ in range(300):
        wl.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        wl.step()
    torch.cuda.synchronize()
    print(f"{kind:14s} {(time.perf_counter() - t0) / n * 1e3:.4f} ms/step", flush=True)
    del wl; torch.cuda.empty_cache()
