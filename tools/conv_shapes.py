"""The 3x3 convolutions of the Zero123 SDS step by shape (SD-1.x UNet at batch 8 / 32^2 latents, VAE encoder at batch 4 / 256^2),
NHWC fp16: time and achieved TFLOP/s of the library path (MIOpen through torch) and, when built, of the hand-written
implicit-GEMM kernel (csrc/conv_mfma.hip) -- the per-shape table of profiles/r03_zero123.md."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.nn.functional as F

PEAK = 2.5e15
UNET = [  # (N, H, Cin, Cout, count per step)  -- openaimodel.py:429-842 at model_channels 320, mult (1,2,4,4), 2 res blocks
    (8, 32, 320, 320, 5), (8, 32, 960, 320, 1), (8, 32, 640, 320, 2), (8, 32, 640, 640, 1),
    (8, 16, 320, 640, 1), (8, 16, 640, 640, 6), (8, 16, 1920, 640, 1), (8, 16, 1280, 640, 1), (8, 16, 960, 640, 1), (8, 16, 1280, 1280, 1),
    (8, 8, 640, 1280, 1), (8, 8, 1280, 1280, 7), (8, 8, 2560, 1280, 2), (8, 8, 1920, 1280, 1),
    (8, 4, 1280, 1280, 9), (8, 4, 2560, 1280, 3)]
VAE = [   # encoder forward at batch 4 (model.py Encoder, ch 128, mult (1,2,4,4)); the backward runs the same shapes as data gradients
    (4, 256, 128, 128, 4), (4, 128, 128, 256, 1), (4, 128, 256, 256, 3), (4, 64, 256, 512, 1), (4, 64, 512, 512, 3), (4, 32, 512, 512, 8)]

def bench(fn, n=20):
    """GPU time per call: n calls captured in a hipGraph and replayed (the Python wrappers cost as much as the small kernels)."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (5 * n)

dev = torch.device("cuda:0")
try:
    from dreammesh4d_amd import conv_mfma
except Exception:
    conv_mfma = None
tot_lib = tot_mine = 0.0
print(f"{'shape':34s} {'GFLOP':>7s} {'MIOpen us':>10s} {'TF/s':>7s} {'frac':>6s}" + ("   mine us    TF/s   frac   max|diff|" if conv_mfma else ""))
for name, shapes in (("UNet", UNET), ("VAE", VAE)):
    for (N, H, Ci, Co, cnt) in shapes:
        x = torch.randn(N, Ci, H, H, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Co, Ci, 3, 3, device=dev, dtype=torch.float16) * (1.0 / (3 * Ci ** 0.5))).contiguous(memory_format=torch.channels_last)
        b = torch.randn(Co, device=dev, dtype=torch.float16)
        fl = 2.0 * N * H * H * Co * Ci * 9
        with torch.no_grad():
            t = bench(lambda: F.conv2d(x, w, b, padding=1))
            line = f"{name} {N}x{H}x{H} {Ci:4d}->{Co:4d} x{cnt:<2d}".ljust(34) + f" {fl/1e9:7.1f} {t*1e6:10.1f} {fl/t/1e12:7.1f} {fl/t/PEAK:6.3f}"
            tot_lib += t * cnt
            if conv_mfma is not None and conv_mfma.supported(x, w):
                pw = conv_mfma.pack_weight(w)
                ref = F.conv2d(x, w, b, padding=1)
                out = conv_mfma.conv3x3(x, pw, b)
                err = float((out.float() - ref.float()).abs().max())
                t2 = bench(lambda: conv_mfma.conv3x3(x, pw, b))
                tot_mine += t2 * cnt
                line += f" {t2*1e6:9.1f} {fl/t2/1e12:7.1f} {fl/t2/PEAK:6.3f}   {err:.3e}"
            else:
                tot_mine += t * cnt
        print(line)
print(f"sum over a step (forward only, counts applied): library {tot_lib*1e3:.2f} ms" + (f", hand-written where supported {tot_mine*1e3:.2f} ms" if conv_mfma else ""))
