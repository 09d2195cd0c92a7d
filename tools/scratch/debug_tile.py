import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from tests.test_views_gpu import _scene
from tests.test_tile_records_gpu import _run
from dreammesh4d_amd import views
dev = torch.device("cuda:0")
H, W = int(sys.argv[1]), int(sys.argv[2]); learn = len(sys.argv) > 3
B, M = 2, 100
sc, graph, topo, qs, scales, opac, rgb, raw, cams, vm, pm = _scene(2400, M, 4, B, H, W, dev, seed=2)
gen = torch.Generator().manual_seed(1)
gC = torch.randn(B, 6, H, W, generator=gen).to(dev)
gD = (0.1 * torch.randn(B, 1, H, W, generator=gen)).to(dev)
gA = torch.randn(B, 1, H, W, generator=gen).to(dev)
res = {}
for det in (True, False):
    r = views.ViewRenderer(graph, topo, H, W, cams[0].tanfov, method="hybrid", deterministic=det)
    st = [t.clone().requires_grad_(learn) for t in (scales, opac, rgb)]
    res[det] = _run(r, raw, qs, st, vm, pm, gC, gD, gA, dev) + (r,)
    print(det, "records", r.last_num_records)
a, b = res[True][2], res[False][2]
for k in a:
    if a[k] is None: continue
    d = np.abs(a[k] - b[k]); m = np.abs(a[k]).max()
    bad = np.argwhere(d > 1e-4 * m)
    print(k, "max diff", d.max(), "max", m, "n bad", len(bad), bad[:5].tolist())
m2a, m2b = a["m2"], b["m2"]
bad = np.unique(np.argwhere(np.abs(m2a - m2b) > 1e-4 * np.abs(m2a).max())[:, :2], axis=0)
print("bad gaussians", len(bad), "of", m2a.shape[1] * B)
for (v, i) in bad[:10]:
    print(v, i, m2a[v, i], m2b[v, i], "radius", int(res[False][0]["radii"][v, i]))
