"""Distribution of the tile-list lengths of the bench workload and how the blend backward's work (consumed cell-list
entries) is spread over tiles by list length: what the tile-record kernel's window size / tile classes have to cover."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from dreammesh4d_amd import _lib
dev = torch.device('cuda:0')
wl = bench.Workload(dev, 0, 1)
wl.step(); torch.cuda.synchronize()
r = wl.renderer
vs, ws = r.last
L = _lib.lib()
B = vs.B
stride = L.dm4d_views_geom_bytes(1, r.N, 512, 512)
geom = ws["geom"].cpu().numpy()
T = 1024
al = lambda x: (x + 255) // 256 * 256
off = 256; tc = off; off = al(off + T * 4); off = al(off + (T + 1) * 4); cc = off; off = al(off + T * 64); cd = off
n, work, wide, tot = [], [], [], []
for b in range(B):
    g = geom[b * stride:(b + 1) * stride]
    n.append(g[tc:tc + T * 4].view(np.uint32).copy())
    c = g[cc:cc + T * 64].view(np.uint32).reshape(T, 16)
    d = g[cd:cd + T * 64].view(np.uint32).reshape(T, 16)
    work.append(np.minimum(c, d).sum(1))
    tot.append(c.sum(1))
    wide.append((c >= 128).sum(1))
n, work, wide = np.concatenate(n), np.concatenate(work), np.concatenate(wide)
tot = np.concatenate(tot)
print("cell-list entries (= backward records)", tot.sum(), "consumed by the forward", work.sum(), "=", work.sum() / tot.sum())
print("tiles", n.size, "duplicates", n.sum(), "consumed cell entries", work.sum(), "max tile", n.max())
edges = [0, 1, 128, 256, 384, 496, 768, 1024, 1536, 2048, 4096, 1 << 30]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (n >= lo) & (n < hi)
    print(f"n in [{lo:5d}, {hi:10d}): tiles {m.sum():5d} ({100*m.sum()/n.size:5.1f} %)  duplicates {100*n[m].sum()/n.sum():5.1f} %  work {100*work[m].sum()/work.sum():5.1f} %  wide cells per tile {wide[m].mean() if m.any() else 0:.2f}")
