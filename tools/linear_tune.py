"""dm4d_linear_f16 per shape under every (tile configuration, split count): what linear_plan (csrc/conv_mfma.hip) should pick."""
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
SH = [(8192, 320, 320), (8192, 320, 960), (8192, 1280, 320), (8192, 640, 320), (8192, 960, 320), (2048, 640, 640), (2048, 640, 1920), (2048, 2560, 640), (2048, 1920, 640),
      (512, 1280, 1280), (512, 1280, 3840), (512, 5120, 1280), (512, 2560, 1280), (512, 1920, 1280), (128, 2560, 1280), (128, 1280, 1280)]
cfgs = [int(c) for c in os.environ.get("CFGS", "3,13").split(",")]
for M, K, N in SH:
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5
    b = torch.randn(N, device=dev, dtype=torch.float16)
    lib = bench(lambda: F.linear(x, w, b))
    out = []
    for cfg in cfgs:
        os.environ["DM4D_LIN_CFG"] = str(cfg)
        for sp in (1, 2, 3, 4, 6, 8):
            if K // 32 // sp < 4: continue
            os.environ["DM4D_LIN_SPLITS"] = str(sp)
            out.append((bench(lambda: conv_mfma.linear(x, w, b)), cfg, sp))
    os.environ.pop("DM4D_LIN_CFG"); os.environ.pop("DM4D_LIN_SPLITS")
    plan = bench(lambda: conv_mfma.linear(x, w, b))
    out.sort()
    print(f"{M}x{K}->{N}: library {lib:5.1f}  plan {plan:5.1f}  best " + "  ".join(f"cfg{c}/s{s}:{t:5.1f}" for t, c, s in out[:4]))
