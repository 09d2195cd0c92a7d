"""dynamic-stage iterations/sec (BASELINE.json's second metric) at the bench configuration: 199,980 mesh-bound
Gaussians, 512^2, 4 frames x (1 reference view + 1 SDS view) per iteration, full-size Zero123 (SD-1.x UNet 860 M
parameters + VAE encoder, fp16, RANDOM weights -- the checkpoint is not in the tree), AdamW step included."""
import json, sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, '/root/repo')
import bench
from dreammesh4d_amd import zero123 as z, synthetic as syn
from dreammesh4d_amd.dynamic_stage import DynamicStage

dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
wl = bench.Workload(dev, 0, 1)
L, H, W = bench.N_FRAMES, bench.H, bench.W
t0 = time.time()
with torch.device(dev):
    model = z.Zero123()                       # full-size hyper-parameters (zero123.py defaults = the reference's yaml)
print("Zero123 parameters: %.1f M, built in %.1f s" % (sum(p.numel() for p in model.parameters()) / 1e6, time.time() - t0), flush=True)
g = torch.Generator(device="cpu").manual_seed(0)
guid = z.TemporalStableZero123Guidance(model, torch.randn(L, 1, 768, generator=g), torch.randn(L, 4, 32, 32, generator=g),
                                       cond_elevation_deg=5.0, half_precision_weights=True).to(dev)
static = {"q_static": wl.qs, "scales": wl.scales, "opacities": wl.opac, "rgb": wl.rgb}
cam = syn.make_camera(H, W, elev_deg=5.0, azim_deg=0.0)
ref_img = torch.rand(L, H, W, 3, generator=g).to(dev)
ref_mask = (torch.rand(L, H, W, 1, generator=g) > 0.5).float().to(dev)
stage = DynamicStage(wl.renderer, wl.net, wl.nodes, static, wl.timestamps, ref_img, ref_mask, cam, guidance=guid,
                     frames_per_step=4, random_views_per_frame=1)
for i in range(3):
    out = stage.iteration()
    torch.cuda.synchronize()
    print("warmup", i, {k: float(v) for k, v in out.items()}, flush=True)
n = 10
t0 = time.perf_counter()
for _ in range(n):
    stage.iteration()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"dynamic_stage_iters_per_sec": round(n / dt, 3), "ms_per_iteration": round(1e3 * dt / n, 2),
                  "views_per_iteration": 8, "zero123": "full size, fp16, random weights"}))
