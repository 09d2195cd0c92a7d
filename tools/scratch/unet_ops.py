import os, sys, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch.profiler import profile, ProfilerActivity
from dreammesh4d_amd import zero123 as z
dev = torch.device('cuda:0'); torch.manual_seed(0)
with torch.device(dev):
    model = z.Zero123()
model = model.half().to(dev)
for p in model.parameters(): p.requires_grad_(False)
unet = model.model.diffusion_model
x = torch.randn(8, 8, 32, 32, device=dev, dtype=torch.float16); tt = torch.randint(20, 980, (8,), device=dev); ctx = torch.randn(8, 1, 768, device=dev, dtype=torch.float16)
with torch.no_grad():
    for _ in range(3): unet(x, tt, context=ctx)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        unet(x, tt, context=ctx)
        torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0, None])
for e in prof.events():
    if e.name in ("aten::add", "aten::copy_", "aten::cat", "aten::contiguous", "aten::add_", "aten::mul", "aten::silu", "aten::fill_", "aten::zero_", "aten::clone", "aten::to", "aten::_to_copy") and e.device_time_total > 0:
        st = [s for s in (e.stack or []) if "dreammesh4d_amd" in s]
        key = (e.name, str(e.input_shapes)[:80], st[0][-60:] if st else "?")
        a = acc[key]; a[0] += 1; a[1] += e.device_time_total
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[1]:8.1f} us {v[0]:4d}  {k[0]:16s} {k[1]:80s} {k[2]}")
