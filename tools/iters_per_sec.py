"""dynamic-stage iterations/sec (BASELINE.json's second metric) at the bench configuration: 199,980 mesh-bound
Gaussians, 512^2, 4 frames x (1 reference view + 1 SDS view) per iteration, full-size Zero123 (SD-1.x UNet 860 M
parameters + VAE encoder, fp16, RANDOM weights -- the checkpoint is not in the tree), AdamW step included."""
import json, os, sys, time, torch
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
kw = {}
if "--reg" in sys.argv or os.environ.get("DM4D_ITER_REG"):      # + the regularisers of bench.py's leg: mesh normal consistency, key-frame ARAP
    from dreammesh4d_amd.mesh_reg import ARAPCoach, MeshNormalConsistency
    kw = dict(normal_consistency=MeshNormalConsistency(wl.sc["faces"], len(wl.sc["verts"]), dev), arap=ARAPCoach(wl.sc["verts"], wl.sc["faces"], dev),
              milestone_arap_reg=0)
stage = DynamicStage(wl.renderer, wl.net, wl.nodes, static, wl.timestamps, ref_img, ref_mask, cam, guidance=guid,
                     frames_per_step=4, random_views_per_frame=int(os.environ.get("DM4D_ITER_RND_VIEWS", "1")), **kw)
for i in range(3):
    out = stage.iteration()
    torch.cuda.synchronize()
    print("warmup", i, {k: float(v) for k, v in out.items()}, flush=True)
if "--sync-debug" in sys.argv:      # every implicit host <-> device synchronisation of one iteration, with its Python stack
    import warnings, traceback
    def show(msg, cat, fn, ln, *a):
        print(f"SYNC at {fn}:{ln}: {msg}"); traceback.print_stack(limit=9)
    warnings.showwarning = show; warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    stage.iteration()
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
n = int(os.environ.get("DM4D_ITERS", "10"))
host = 0.0
t0 = time.perf_counter()
for _ in range(n):
    h0 = time.perf_counter()
    stage.iteration()
    host += time.perf_counter() - h0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
if "--host" in sys.argv:            # time the host spends enqueueing an iteration (== the iteration when the loop is host-bound)
    print(f"host time inside iteration(): {1e3 * host / n:.2f} ms of {1e3 * dt / n:.2f} ms per iteration")
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): stage.iteration()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
print(json.dumps({"dynamic_stage_iters_per_sec": round(n / dt, 3), "ms_per_iteration": round(1e3 * dt / n, 2),
                  "views_per_iteration": 4 * (1 + int(os.environ.get("DM4D_ITER_RND_VIEWS", "1"))), "zero123": "full size, fp16, random weights"}))
