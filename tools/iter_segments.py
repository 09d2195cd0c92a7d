"""Where the GPU's time of a dynamic-stage iteration goes WITHOUT a profiler attached: hip events recorded on the stream at
the boundaries of its segments (render forward, loss glue, VAE encoder graph, SDS glue, UNet graph, SDS glue, backward = VAE
backward graph + glue + render backward, optimiser, next batch), averaged over steady-state iterations.  Event-to-event time
includes whatever idles between the kernels; compare with the kernel sums of tools/iter_sequence.sh (rocprofv3)."""
import collections, os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from dreammesh4d_amd import zero123 as z, synthetic as syn, dynamic_stage as ds
from dreammesh4d_amd.dynamic_stage import DynamicStage

dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
wl = bench.Workload(dev, 0, 1)
L, H, W = bench.N_FRAMES, bench.H, bench.W
with torch.device(dev):
    model = z.Zero123()
g = torch.Generator(device="cpu").manual_seed(0)
guid = z.TemporalStableZero123Guidance(model, torch.randn(L, 1, 768, generator=g), torch.randn(L, 4, 32, 32, generator=g),
                                       cond_elevation_deg=5.0, half_precision_weights=True).to(dev)
static = {"q_static": wl.qs, "scales": wl.scales, "opacities": wl.opac, "rgb": wl.rgb}
cam = syn.make_camera(H, W, elev_deg=5.0, azim_deg=0.0)
kw = {}
if "--reg" in sys.argv:      # the regularisers of bench.py's leg: mesh normal consistency + key-frame ARAP
    from dreammesh4d_amd.mesh_reg import ARAPCoach, MeshNormalConsistency
    kw = dict(normal_consistency=MeshNormalConsistency(wl.sc["faces"], len(wl.sc["verts"]), dev), arap=ARAPCoach(wl.sc["verts"], wl.sc["faces"], dev),
              milestone_arap_reg=0)
stage = DynamicStage(wl.renderer, wl.net, wl.nodes, static, wl.timestamps, torch.rand(L, H, W, 3, generator=g).to(dev),
                     (torch.rand(L, H, W, 1, generator=g) > 0.5).float().to(dev), cam, guidance=guid, frames_per_step=4, random_views_per_frame=1, **kw)
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
def wrap(obj, attr, before, after):
    f = getattr(obj, attr)
    def w(*a, **k):
        mark(before); r = f(*a, **k); mark(after); return r
    setattr(obj, attr, w)
wrap(ds, "render_views", "batch + node network", "render forward")
wrap(guid, "encode_images", "loss glue + interpolate", "VAE encoder (graph)")
wrap(guid, "_unet", "SDS glue (cond, noise)", "UNet (graph)")
_bw = torch.Tensor.backward
def bw(self, *a, **k):
    mark("SDS glue (grad, loss) + regularisers"); r = _bw(self, *a, **k); mark("backward (VAE graph, glue, render)"); return r
torch.Tensor.backward = bw
wrap(stage.opt, "step", "overflow flag", "optimiser")
for _ in range(4): stage.iteration()
torch.cuda.synchronize()
n = 30; marks.clear(); mark("start")
t0 = time.perf_counter()
for _ in range(n):
    stage.iteration(); mark("end of iteration (host tail)")
torch.cuda.synchronize(); dt = time.perf_counter() - t0
acc = collections.OrderedDict()
for (_, a), (nm, b) in zip(marks[:-1], marks[1:]):
    acc[nm] = acc.get(nm, 0.0) + a.elapsed_time(b)
print(f"{1e3 * dt / n:.2f} ms per iteration (wall), {sum(acc.values()) / n:.2f} ms between the first and the last event")
for k, v in acc.items(): print(f"  {v / n:7.3f} ms  {k}")
