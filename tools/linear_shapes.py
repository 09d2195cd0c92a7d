"""The linear layers of one UNet forward (batch 8: tokens x widths of the three attention levels + the 1x1 skip convolutions):
GPU time per call of dm4d_linear_f16 (csrc/conv_mfma.hip, one-tap implicit GEMM) against F.linear / torch.addmm (hipBLASLt),
from hipGraphs of 20 calls.  DM4D_LIN_CFG / DM4D_LIN_SPLITS force a tile configuration / split count."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.nn.functional as F
from dreammesh4d_amd import conv_mfma
from dreammesh4d_amd.fused_norm import geglu
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
rows = []
for L, C, nblk in ((1024, 320, 4), (256, 640, 4), (64, 1280, 6), (16, 1280, 2)):
    M = 8 * L
    if L >= 64:
        rows += [(f"proj_in/out, attn out {M}x{C}->{C}", M, C, C, "res", 3 * nblk if L > 16 else 0), (f"qkv {M}x{C}->{3*C}", M, C, 3 * C, None, nblk),
                 (f"ff1+geglu {M}x{C}->{8*C}", M, C, 8 * C, "geglu", nblk), (f"ff2 {M}x{4*C}->{C}", M, 4 * C, C, "res", nblk)]
rows += [("skip 8192x640->320", 8192, 640, 320, None, 2), ("skip 8192x960->320", 8192, 960, 320, None, 1), ("skip 2048x320->640", 2048, 320, 640, None, 1),
         ("skip 2048x1920->640", 2048, 1920, 640, None, 1), ("skip 2048x1280->640", 2048, 1280, 640, None, 1), ("skip 2048x960->640", 2048, 960, 640, None, 1),
         ("skip 512x640->1280", 512, 640, 1280, None, 1), ("skip 512x2560->1280", 512, 2560, 1280, None, 3), ("skip 512x1920->1280", 512, 1920, 1280, None, 1),
         ("skip 128x2560->1280", 128, 2560, 1280, None, 2)]
tot_lib = tot_own = 0.0
for name, M, K, N, ep, calls in rows:
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5
    b = torch.randn(N, device=dev, dtype=torch.float16)
    r = torch.randn(M, N, device=dev, dtype=torch.float16)
    if ep == "geglu":
        wp, bp = conv_mfma.pack_geglu(w, b)
        t_lib = bench(lambda: geglu(F.linear(x, w, b)))
        t_own = bench(lambda: conv_mfma.linear(x, wp, bp, act="geglu"))
    elif ep == "res":
        rb = r + b
        t_lib = bench(lambda: torch.addmm(rb, x, w.t()))
        t_own = bench(lambda: conv_mfma.linear(x, w, b, r))
    else:
        t_lib = bench(lambda: F.linear(x, w, b))
        t_own = bench(lambda: conv_mfma.linear(x, w, b))
    gf = 2.0 * M * K * N / 1e9
    tot_lib += t_lib * calls; tot_own += t_own * calls
    print(f"{name:38s} x{calls:2d}  {gf:6.1f} GFLOP  library {t_lib:6.1f} us  own {t_own:6.1f} us ({gf / t_own * 1e-3 * 1e3:6.0f} TFLOP/s)")
print(f"sum over a UNet forward: library {tot_lib / 1e3:.2f} ms, own {tot_own / 1e3:.2f} ms")
