"""Time of the Zero123 SDS step (full size, fp16, random weights; 4 SDS views = UNet batch 8 at 32x32 latents + VAE encoder at
256^2 batch 4 with backward to the images) under a few PyTorch-ROCm settings."""
import sys, time, torch
sys.path.insert(0, '.')
from dreammesh4d_amd import zero123 as z
dev = torch.device('cuda:0')
L = 32
def build(cl):
    torch.manual_seed(0)
    with torch.device(dev):
        model = z.Zero123()
    g = torch.Generator(device="cpu").manual_seed(0)
    guid = z.TemporalStableZero123Guidance(model, torch.randn(L, 1, 768, generator=g), torch.randn(L, 4, 32, 32, generator=g),
                                           cond_elevation_deg=5.0, half_precision_weights=True, channels_last=cl).to(dev)
    return guid
def run(guid, n=10):
    rgb = torch.rand(4, 512, 512, 3, device=dev, requires_grad=True)
    el = torch.tensor([10., 20., 30., 40.], device=dev); az = torch.tensor([0., 90., 180., 270.], device=dev)
    fi = torch.tensor([0, 5, 9, 13], device=dev)
    for _ in range(3):
        guid(rgb, el, az, torch.full_like(el, 3.8), frame_indices=fi)["loss_sds"].backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        guid(rgb, el, az, torch.full_like(el, 3.8), frame_indices=fi)["loss_sds"].backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for name, bench_flag, cl in (("nchw, library GroupNorm", False, False), ("nhwc, HIP GroupNorm+SiLU", False, True),
                             ("nhwc + miopen benchmark", True, True)):
    torch.backends.cudnn.benchmark = bench_flag
    guid = build(cl)
    print(f"{name:18s} {run(guid):7.2f} ms per SDS step", flush=True)
    del guid; torch.cuda.empty_cache()
