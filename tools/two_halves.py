"""Experiment (verdict r03, item 1b): the 8-view bench step as TWO 4-view half-batches on two streams, so that one half's
latency-bound kernels (node network, skinning, binning, sort, the backward's tail) can run under the other half's VALU-bound blend
kernels.  Two step objects (dreammesh4d_amd/step.py), each with its own renderer / workspaces; same views, same kernels.
Prints ms per 8 views for: one 8-view step | two 4-view steps on ONE stream | two 4-view steps on TWO streams."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from dreammesh4d_amd import views
from dreammesh4d_amd.step import DynamicStep

dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
wl = bench.Workload(dev, 0, 1)
H = W = bench.H
halves = []
for h in range(2):
    r = views.ViewRenderer(wl.graph, wl.topo, H, W, wl.cams[0].tanfov, method="hybrid")
    r.fuse_face_backward = True
    st = DynamicStep(r, wl.net, wl.nodes, wl.qs, wl.scales, wl.opac, wl.rgb, wl.bg6, n_views=4, n_frames=2)
    sl = slice(4 * h, 4 * h + 4)
    halves.append(dict(step=st, t=wl.frame_t[2 * h:2 * h + 2].contiguous(), vm=wl.vm[sl].contiguous(), pm=wl.pm[sl].contiguous(),
                       fidx=(wl.fidx[sl] - 2 * h).contiguous(), gC=wl.gC[sl].contiguous(), gA=wl.gA[sl].contiguous()))
params = list(wl.net.parameters())
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]

def two(use_streams):
    outs = []
    for p in params: p.grad = None
    for h, hb in enumerate(halves):
        ctx = torch.cuda.stream(streams[h]) if use_streams else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            outs.append(hb["step"](hb["t"], hb["vm"], hb["pm"], hb["fidx"]))
    for h, hb in enumerate(halves):
        # the two halves write the SAME parameters' gradients: the second backward must not find a .grad (step.py's contract):
        # keep the first half's gradients aside (a real loop would add them: 13.5 MB of touched texels)
        if h == 1:
            for p in params: p.grad = None
        ctx = torch.cuda.stream(streams[h]) if use_streams else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            torch.autograd.backward([outs[h]["color"], outs[h]["alpha"]], [hb["gC"], hb["gA"]])

def timed(fn, n=300):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for rep in range(2):
    print(f"one 8-view step            {timed(wl.step):.4f} ms")
    print(f"two 4-view steps, 1 stream {timed(lambda: two(False)):.4f} ms")
    for s in streams: s.wait_stream(torch.cuda.current_stream())
    print(f"two 4-view steps, 2 streams {timed(lambda: two(True)):.4f} ms")
    torch.cuda.current_stream().wait_stream(streams[0]); torch.cuda.current_stream().wait_stream(streams[1])
