import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dreammesh4d_amd import conv_mfma
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
    return (time.perf_counter() - t0) / (5 * n)
SH = [(8,4,1280,1280),(8,8,1280,1280),(8,8,2560,1280),(8,16,640,640),(8,16,1280,640),(8,32,320,320),(8,32,640,320),(4,32,512,512)]
for (N,H,Ci,Co) in SH:
    x = torch.randn(N, Ci, H, H, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    pw = conv_mfma.pack_weight(torch.randn(Co, Ci, 3, 3, device=dev, dtype=torch.float16) * 0.02)
    row = []
    for cfg in (3, 10, 11):
        os.environ["DM4D_CONV_CFG"] = str(cfg)
        row.append(f"cfg{cfg}:{bench(lambda: conv_mfma.conv3x3(x, pw))*1e6:6.1f}")
    print(f"{N}x{H}^2 {Ci}->{Co} ", " ".join(row))
