"""Distribution of the cell-list lengths (ccount / cdone) of the bench workload: how much of the blend work sits in long cells."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
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
off = 256; off = al(off + T * 4); off = al(off + (T + 1) * 4); cc = off; off = al(off + T * 64); cd = off
cnt, done = [], []
for b in range(B):
    g = geom[b * stride:(b + 1) * stride]
    cnt.append(g[cc:cc + T * 64].view(np.uint32).copy())
    done.append(g[cd:cd + T * 64].view(np.uint32).copy())
cnt, done = np.concatenate(cnt), np.concatenate(done)
print("cells", cnt.size, "entries", cnt.sum(), "consumed", done.sum())
for th in (128, 192, 256, 384, 512, 768, 1024):
    m = cnt >= th
    print(f"ccount >= {th}: cells {m.sum()}  entries {cnt[m].sum()} ({100*cnt[m].sum()/cnt.sum():.1f} %)  consumed {done[m].sum()} ({100*done[m].sum()/done.sum():.1f} %)  max {cnt.max()}")
