"""Per-wave timeline of the blend kernels on the bench workload (dm4d_debug_trace)."""
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
B, blocks = 1, 8 * 4096 + 8 * 256      # backward: 8 x 256 wide blocks first, then the quadrants
buf = torch.zeros(B * blocks, 4, dtype=torch.int64, device=dev)

def analyse(name, a):
    t0, t1, hw, work = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    base = t0.min()
    span = (t1.max() - base) / 100.0          # us
    dur = (t1 - t0) / 100.0
    print(f"== {name}: waves {len(a)}  span {span:.1f} us  mean wave {dur.mean():.2f} us  max wave {dur.max():.1f} us  sum {dur.sum()/1e3:.1f} ms")
    nz = work > 0
    print(f"   non-empty {nz.sum()}  us/iteration (non-empty) {dur[nz].sum()/work[nz].sum():.4f}  mean resident waves {dur.sum()/span:.1f} (of 1024 SIMDs x occupancy)")
    # concurrency over time
    edges = np.linspace(0, span, 21)
    ev = np.concatenate([np.stack([(t0-base)/100.0, np.ones(len(a))], 1), np.stack([(t1-base)/100.0, -np.ones(len(a))], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    conc = np.cumsum(ev[:, 1])
    out = []
    for i in range(20):
        m = (ev[:, 0] >= edges[i]) & (ev[:, 0] < edges[i+1])
        out.append(int(conc[m].mean()) if m.any() else 0)
    print("   resident waves over time (20 bins):", out)
    xcc = (hw >> 32) & 0xF
    hwid = hw & 0xFFFFFFFF
    cu = (hwid >> 8) & 0xF; se = (hwid >> 13) & 0x7; sh = (hwid >> 12) & 1; simd = (hwid >> 4) & 3
    for x in range(8):
        m = xcc == x
        print(f"   xcc {x}: waves {m.sum()}  work {work[m].sum()}  last end {((t1[m].max()-base)/100.0) if m.any() else 0:.1f} us")
    key = xcc * 1000 + se * 100 + sh*50 + cu
    print("   distinct (xcc,se,sh,cu):", len(np.unique(key)), " distinct simd slots:", len(np.unique(key*4+simd)))
    # longest waves: when did they start
    idx = np.argsort(-work)[:5]
    print("   top work:", [(int(work[i]), round(float((t0[i]-base)/100.0),1), round(float(dur[i]),1)) for i in idx])

MINW = int(sys.argv[1]) if len(sys.argv) > 1 else 0
_lib.check(L.dm4d_debug_trace(buf.data_ptr(), MINW))
# forward only
out = wl.step.__func__  # noqa
dx, dr, ds, do = wl.net.node_outputs(wl.nodes, wl.frame_t)
o = wl.render_views(wl.renderer, dx, dr, ds, do, wl.qs, wl.scales, wl.opac, wl.rgb, wl.vm, wl.pm, wl.bg6, frame_index=wl.fidx)
torch.cuda.synchronize()
fw = buf.cpu().numpy().copy()
buf.zero_()
torch.autograd.backward([o["color"], o["alpha"]], [wl.gC, wl.gA])
torch.cuda.synchronize()
bw = buf.cpu().numpy().copy()
_lib.check(L.dm4d_debug_trace(None, 0))
analyse("render_fwd", fw[:8 * 4096])
analyse("render_bwd (wide blocks: one wave per long cell)", bw[:8 * 256])
analyse("render_bwd (regular blocks)", bw[8 * 256:])
analyse("render_bwd (all)", bw)
