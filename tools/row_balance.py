"""How well the four cell lists of a blend wave are balanced: wave steps (max over its 4 rows) against the mean, and
what grouping cells of similar length (instead of the 4 cells of a quadrant) would give."""
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
for name, o in (("forward (ccount)", cc), ("backward (cdone)", cd)):
    a = np.concatenate([geom[b * stride:(b + 1) * stride][o:o + T * 64].view(np.uint32).copy() for b in range(B)]).astype(np.int64)
    a[a >= 384] = 0                       # long cells have their own blocks
    w = a.reshape(-1, 4)
    steps = w.max(axis=1).sum()
    ideal = a.sum() / 4
    s = np.sort(a)[::-1].reshape(-1, 4)
    print(f"{name}: wave steps {steps}  sum/4 {ideal:.0f}  efficiency {ideal/steps:.3f}   grouped by length: {s.max(axis=1).sum()} ({ideal/s.max(axis=1).sum():.3f})")
    # per tile grouping by length (16 cells of a tile sorted, 4 waves)
    t = np.sort(a.reshape(-1, 16), axis=1)[:, ::-1].reshape(-1, 4)
    print(f"    the 16 cells of a tile sorted by length, 4 per wave: {t.max(axis=1).sum()} ({ideal/t.max(axis=1).sum():.3f})")
