"""The direct-kernel shapes (VAE encoder levels + two ragged ones) under the tile configurations of CFGS (default 7,9): GPU time per call
from a hipGraph of 20 calls and the relative error against torch in float32."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.nn.functional as F
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
    return (time.perf_counter() - t0) / (5 * n) * 1e6
for (N,H,Ci,Co) in [(4,256,128,128),(4,128,128,256),(4,128,256,256),(4,64,256,512),(4,64,512,512),(4,32,512,512),(8,32,320,320),(8,32,640,320),(8,32,960,320),(8,32,640,640),(1,32,64,128),(2,64,96,160)]:
    x = torch.randn(N, Ci, H, H, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, Ci, 3, 3, device=dev, dtype=torch.float16) * (9*Ci)**-0.5)
    b = torch.randn(Co, device=dev, dtype=torch.float16)
    pw = conv_mfma.pack_weight(w)
    ref = F.conv2d(x.float(), w.float(), b.float(), 1, 1)
    row = []
    for cfg in [int(c) for c in os.environ.get('CFGS', '7,9').split(',')]:
        os.environ["DM4D_CONV_CFG"] = str(cfg)
        y = conv_mfma.conv3x3(x, pw, b)
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        row.append(f"cfg{cfg}: {bench(lambda: conv_mfma.conv3x3(x, pw, b)):6.1f} us err {err:.1e}")
    print(f"{N}x{H}^2 {Ci}->{Co}  " + "   ".join(row))
