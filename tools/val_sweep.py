"""Throughput of the validation sweep (validation.sweep) on the bench scene: 32 frames x 5 azimuths, 512^2, forward only."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from dreammesh4d_amd import validation
dev = torch.device('cuda:0')
wl = bench.Workload(dev, 0, 1)
static = {"q_static": wl.qs, "scales": wl.scales, "opacities": wl.opac, "rgb": wl.rgb}
n = [0]
def sink(fr, ch):
    n[0] += ch["comp_rgb"].shape[0] * ch["comp_rgb"].shape[1]
for fpc in (2, 3):
    for rep in range(3):
        n[0] = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        validation.sweep(wl.renderer, wl.net, wl.nodes, static, wl.timestamps, frames_per_call=fpc, on_chunk=sink)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"frames_per_call {fpc}: {n[0]} views in {dt*1e3:.1f} ms = {n[0]/dt:.0f} views/s forward-only (incl. deformation + epilogue)")
