"""The UNet's self-attention shapes (batch 8, 8 heads): dm4d_attention_f16 (csrc/attention.hip) against
F.scaled_dot_product_attention on strided views of the same fused projection, GPU time per call from a hipGraph of 20 calls."""
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
tot_l = tot_o = 0.0
for L, D, calls in ((1024, 40, 4), (256, 80, 4), (64, 160, 6)):
    qkv = torch.randn(8, L, 3, 8, D, device=dev, dtype=torch.float16)
    def lib():
        o = F.scaled_dot_product_attention(qkv[:, :, 0].transpose(1, 2), qkv[:, :, 1].transpose(1, 2), qkv[:, :, 2].transpose(1, 2))
        return o.transpose(1, 2).reshape(8, L, -1)
    tl, to = bench(lib), bench(lambda: conv_mfma.attention_qkv(qkv))
    err = float((lib().float() - conv_mfma.attention_qkv(qkv).float()).abs().max())
    gf = 4.0 * 8 * 8 * L * L * D / 1e9
    tot_l += tl * calls; tot_o += to * calls
    print(f"L {L:5d} d {D:4d} x{calls}: library {tl:6.1f} us  own {to:6.1f} us ({gf / to * 1e-3 * 1e3:5.0f} TFLOP/s)  max |difference| {err:.2e}")
print(f"sum over a UNet forward: library {tot_l / 1e3:.3f} ms, own {tot_o / 1e3:.3f} ms")
