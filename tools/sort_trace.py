"""Phase timing of k_tile_sort on the bench workload (dm4d_debug_sort_trace)."""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from dreammesh4d_amd import _lib
dev = torch.device('cuda:0')
wl = bench.Workload(dev, 0, 1, views_per_frame=2)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
L = _lib.lib()
buf = torch.zeros(8 * 1024, 5, dtype=torch.int64, device=dev)
_lib.check(L.dm4d_debug_sort_trace(buf.data_ptr()))
dx, dr, ds, do = wl.net.node_outputs(wl.nodes, wl.frame_t)
o = wl.render_views(wl.renderer, dx, dr, ds, do, wl.qs, wl.scales, wl.opac, wl.rgb, wl.vm, wl.pm, wl.bg6, frame_index=wl.fidx)
torch.cuda.synchronize()
_lib.check(L.dm4d_debug_sort_trace(None))
a = buf.cpu().numpy()
a = a[a[:, 3] > 0]
t0 = a[:, 0].min()
print("tiles (LDS path, non-empty)", len(a), "span us", (a[:, 3].max() - t0) / 100.0)
for name, i, j in (("load+minmax+hist", 0, 1), ("scan+scatter+rank", 1, 2), ("finish_tile", 2, 3), ("total", 0, 3)):
    d = (a[:, j] - a[:, i]) / 100.0
    print(f"  {name:20s} mean {d.mean():7.2f} us  p50 {np.percentile(d,50):7.2f}  p99 {np.percentile(d,99):7.2f}  max {d.max():7.2f}")
n = a[:, 4]
print("  n mean", n.mean(), "max", n.max(), " tiles > 2048:", (n > 2048).sum())
for lo_, hi_ in ((1, 256), (256, 512), (512, 768), (768, 1024), (1024, 1536), (1536, 2049)):
    m_ = (n >= lo_) & (n < hi_)
    if m_.any(): print(f'  n in [{lo_},{hi_}): tiles {m_.sum()}  sort us {((a[m_,2]-a[m_,0])/100.0).mean():.1f}  finish us {((a[m_,3]-a[m_,2])/100.0).mean():.1f}')
lg = a[n > 2048]
if len(lg): print("  large tiles: sort us mean", ((lg[:,2]-lg[:,0])/100.0).mean(), "max", ((lg[:,2]-lg[:,0])/100.0).max(), " finish mean", ((lg[:,3]-lg[:,2])/100.0).mean(), " start us", ((lg[:,0]-t0)/100.0).round(0)[:12], " end us", ((lg[:,3]-t0)/100.0).round(0)[:12])
big = a[np.argsort(-n)[:5]]
print("  biggest tiles (n, phases us):", [(int(r[4]), round((r[1]-r[0])/100,1), round((r[2]-r[1])/100,1), round((r[3]-r[2])/100,1)) for r in big])
# concurrency
ev = np.concatenate([np.stack([(a[:,0]-t0)/100.0, np.ones(len(a))],1), np.stack([(a[:,3]-t0)/100.0, -np.ones(len(a))],1)])
ev = ev[np.argsort(ev[:,0])]; conc = np.cumsum(ev[:,1])
print("  mean resident WGs", ((a[:,3]-a[:,0]).sum()/100.0) / ((a[:,3].max()-t0)/100.0), "max", conc.max())
span = (a[:,3].max()-t0)/100.0
edges = np.linspace(0, span, 21)
out=[]
for i in range(20):
    mid=(edges[i]+edges[i+1])/2
    out.append(int((((a[:,0]-t0)/100.0 <= mid) & ((a[:,3]-t0)/100.0 > mid)).sum()))
print("  resident WGs over time:", out)
last = a[np.argsort(-a[:,3])[:8]]
print("  last finishing tiles (n, start, end):", [(int(r[4]), round((r[0]-t0)/100,0), round((r[3]-t0)/100,0)) for r in last])
st = (a[:,0]-t0)/100.0
print("  start time percentiles:", [round(float(np.percentile(st,p)),0) for p in (10,50,90,99,100)])
