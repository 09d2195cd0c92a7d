import os, sys, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch.profiler import profile, ProfilerActivity
from dreammesh4d_amd import zero123 as z
dev = torch.device('cuda:0'); L = 32
torch.manual_seed(0)
with torch.device(dev):
    model = z.Zero123()
g = torch.Generator(device="cpu").manual_seed(0)
guid = z.TemporalStableZero123Guidance(model, torch.randn(L, 1, 768, generator=g), torch.randn(L, 4, 32, 32, generator=g),
                                       cond_elevation_deg=5.0, half_precision_weights=True, use_graphs=False).to(dev)
rgb = torch.rand(4, 512, 512, 3, device=dev, requires_grad=True)
el = torch.tensor([10., 20., 30., 40.], device=dev); az = torch.tensor([0., 90., 180., 270.], device=dev)
fi = torch.tensor([0, 5, 9, 13], device=dev)
def step():
    out = guid(rgb, el, az, torch.full_like(el, 3.8), frame_indices=fi); out["loss_sds"].backward()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = prof.key_averages(group_by_input_shape=True, group_by_stack_n=8)
out = []
for r in rows:
    sd = getattr(r, "self_device_time_total", None)
    if sd is None: sd = getattr(r, "self_cuda_time_total", 0)
    if sd <= 0 or not r.key.startswith("aten::"): continue
    if r.key.startswith(("aten::conv", "aten::_conv", "aten::miopen", "aten::linear", "aten::addmm", "aten::mm", "aten::bmm", "aten::matmul", "aten::scaled_dot", "aten::_scaled", "aten::_flash", "aten::_efficient", "aten::layer_norm", "aten::native_layer_norm", "aten::baddbmm")): continue
    st = [x for x in (r.stack or []) if "dreammesh4d_amd" in x]
    out.append((sd, r.count, r.key, str(r.input_shapes)[:70], st[0][-75:] if st else "(autograd / other)"))
print(f"leaf aten ops outside conv/GEMM/attention/LayerNorm: {sum(o[0] for o in out)/1e3:.2f} ms")
for o in sorted(out, reverse=True)[:45]:
    print(f"{o[0]:8.1f} us {o[1]:4d}  {o[2]:22s} {o[3]:70s} {o[4]}")
