"""Zero123 SDS step at the bench configuration (full-size SD-1.x UNet 860 M + VAE encoder, fp16, RANDOM weights; 4 SDS
views = UNet batch 8 at 32x32 latents, VAE encoder at 256^2 batch 4 with backward to the images): steady-state time per
step, its split, FLOPs (torch.utils.flop_counter) and achieved TFLOP/s against the 2.5 PF dense fp16 MFMA peak.
  python tools/zero123_profile.py            -> one JSON line
  python tools/zero123_profile.py --steps-only N   -> just N steady-state steps (the command rocprofv3 wraps)"""
import json, sys, time, torch
import os
# the ISOLATED step: no render to run the conditioning graph beside, so the two-graph arrangement only adds its second launch and
# the join (DESIGN.md section 3 "Round 4"); the iteration-level tools (iters_per_sec.py, bench.py) measure it where it pays
os.environ.setdefault("DM4D_SDS_PRE_GRAPH", "0")
sys.path.insert(0, '.'); sys.path.insert(0, '/root/repo')
from dreammesh4d_amd import zero123 as z
dev = torch.device('cuda:0'); L = 32
torch.manual_seed(0)
with torch.device(dev):
    model = z.Zero123()
g = torch.Generator(device="cpu").manual_seed(0)
guid = z.TemporalStableZero123Guidance(model, torch.randn(L, 1, 768, generator=g), torch.randn(L, 4, 32, 32, generator=g),
                                       cond_elevation_deg=5.0, half_precision_weights=True).to(dev)
rgb = torch.rand(4, 512, 512, 3, device=dev, requires_grad=True)
el = torch.tensor([10., 20., 30., 40.], device=dev); az = torch.tensor([0., 90., 180., 270.], device=dev); fi = torch.tensor([0, 5, 9, 13], device=dev)
def step():
    guid(rgb, el, az, torch.full_like(el, 3.8), frame_indices=fi)["loss_sds"].backward()
for _ in range(8):
    step()
torch.cuda.synchronize()
if len(sys.argv) > 2 and sys.argv[1] == "--steps-only":
    for _ in range(int(sys.argv[2])):
        step()
    torch.cuda.synchronize()
    sys.exit(0)
n = 20
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
# FLOPs of one step, counted on an EAGER copy of the same computation (graph replays are invisible to the counter)
from torch.utils.flop_counter import FlopCounterMode
guid_e = z.TemporalStableZero123Guidance(model, guid.c_crossattn.float(), guid.c_concat.float(), cond_elevation_deg=5.0,
                                         half_precision_weights=True, use_graphs=False).to(dev)
from dreammesh4d_amd import conv_mfma        # (the hand-written convolutions are invisible to torch's counter: they keep their own tally)
c0 = conv_mfma.FLOPS[0]
with FlopCounterMode(display=False) as fc:
    guid_e(rgb, el, az, torch.full_like(el, 3.8), frame_indices=fi)["loss_sds"].backward()
flops = fc.get_total_flops() + conv_mfma.FLOPS[0] - c0
# split: the UNet forward alone / the VAE encode + backward alone (eager, events)
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / n * 1e3
x = torch.randn(8, 8, 32, 32, device=dev, dtype=torch.float16); tt = torch.randint(20, 980, (8,), device=dev); ctx = torch.randn(8, 1, 768, device=dev, dtype=torch.float16)
with torch.no_grad():
    unet_ms = timed(lambda: model.model.diffusion_model(x, tt, context=ctx))
c0 = conv_mfma.FLOPS[0]
with FlopCounterMode(display=False) as fu:
    with torch.no_grad():
        model.model.diffusion_model(x, tt, context=ctx)
unet_flops = fu.get_total_flops() + conv_mfma.FLOPS[0] - c0
img = torch.rand(4, 3, 256, 256, device=dev, dtype=torch.float16, requires_grad=True)
vae_ms = timed(lambda: model.first_stage_model.encode_moments(img).float().sum().backward())
c0 = conv_mfma.FLOPS[0]
with FlopCounterMode(display=False) as fv:
    model.first_stage_model.encode_moments(img).float().sum().backward()
vae_flops = fv.get_total_flops() + conv_mfma.FLOPS[0] - c0
print(json.dumps({"ms_per_sds_step": round(ms, 3), "flops_per_step": flops, "achieved_tflops": round(flops / (ms * 1e-3) / 1e12, 2),
                  "frac_of_2.5PF_dense_fp16": round(flops / (ms * 1e-3) / 2.5e15, 4),
                  "unet_fwd_eager_ms": round(unet_ms, 3), "unet_fwd_flops": unet_flops,
                  "vae_enc_fwd_bwd_eager_ms": round(vae_ms, 3), "vae_enc_fwd_bwd_flops": vae_flops}))
