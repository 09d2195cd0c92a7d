"""GroupNorm (+ SiLU) forward per shape of one SDS step (UNet batch 8, VAE encoder batch 4): GPU time per call from a hipGraph of
20 calls.  Run once with DM4D_GN_SLAB=0 (two launches: statistics, apply) and once without (the slab-resident single launch where
gn_slab_plan takes it): csrc/groupnorm.hip."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dreammesh4d_amd.fused_norm import group_norm
dev = torch.device("cuda:0")
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (5 * n) * 1e6
SH = [(8, 32, 320, 12), (8, 32, 640, 2), (8, 32, 960, 1), (8, 16, 320, 1), (8, 16, 640, 12), (8, 16, 1280, 1), (8, 16, 1920, 1), (8, 16, 960, 1),
      (8, 8, 640, 1), (8, 8, 1280, 14), (8, 8, 2560, 2), (8, 8, 1920, 1), (8, 4, 1280, 9), (8, 4, 2560, 3),
      (4, 256, 128, 4), (4, 128, 128, 1), (4, 128, 256, 3), (4, 64, 256, 1), (4, 64, 512, 3), (4, 32, 512, 10)]
tot = 0.0
for N, H, C, calls in SH:
    x = torch.randn(N, C, H, H, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    gn = torch.nn.GroupNorm(32, C, eps=1e-5).to(dev, torch.float16)
    add = torch.randn(N, C, device=dev, dtype=torch.float16)
    with torch.no_grad():
        t = bench(lambda: group_norm(gn, x, silu=True, add=add))
    tot += t * calls
    print(f"{N}x{H}^2x{C:5d} x{calls:2d}: {t:6.1f} us   ({x.numel() * 4 / t / 1e6:6.2f} TB/s read + write)")
print(f"DM4D_GN_SLAB={os.environ.get('DM4D_GN_SLAB', '1')}: sum over a step's forward calls {tot / 1e3:.3f} ms")
