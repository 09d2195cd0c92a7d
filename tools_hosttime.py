import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0")
wl = bench.Workload(dev, 0, 1)
for _ in range(3): wl.step()
torch.cuda.synchronize()
for it in range(3):
    t0 = time.perf_counter()
    P = wl.P
    for p in P.values(): p.grad = None
    a = (P["trans"][wl.fidx], P["d_rot"][wl.fidx], P["strain"][wl.fidx], P["d_opacity"][wl.fidx].squeeze(-1))
    t1 = time.perf_counter()
    out = wl.render_views(wl.renderer, *a, wl.qs, wl.scales, wl.opac, wl.rgb, wl.vm, wl.pm, wl.bg6)
    t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    torch.autograd.backward([out["color"], out["depth"], out["alpha"]], [wl.gC, wl.gD, wl.gA])
    t4 = time.perf_counter()
    torch.cuda.synchronize(); t5 = time.perf_counter()
    print(f"index {1e3*(t1-t0):.2f} fwd-host {1e3*(t2-t1):.2f} fwd-gpu-wait {1e3*(t3-t2):.2f} bwd-host {1e3*(t4-t3):.2f} bwd-gpu-wait {1e3*(t5-t4):.2f} ms")
