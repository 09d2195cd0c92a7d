"""Time of one Zero123 SDS step (full-size UNet + VAE encoder, fp16, random weights, hipGraph replay) and its loss: run with
DM4D_MFMA_CONV=0 / 1 for the A/B of the hand-written convolutions (profiles/r03_zero123.md)."""
import sys, os, time, torch
import os
# the ISOLATED step: no render to run the conditioning graph beside, so the two-graph arrangement only adds its second launch and
# the join (DESIGN.md section 3 "Round 4"); the iteration-level tools (iters_per_sec.py, bench.py) measure it where it pays
os.environ.setdefault("DM4D_SDS_PRE_GRAPH", "0")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dreammesh4d_amd import zero123 as z
dev = torch.device('cuda:0'); L = 32
torch.manual_seed(0)
with torch.device(dev):
    model = z.Zero123()
g = torch.Generator(device="cpu").manual_seed(0)
guid = z.TemporalStableZero123Guidance(model, torch.randn(L, 1, 768, generator=g), torch.randn(L, 4, 32, 32, generator=g),
                                       cond_elevation_deg=5.0, half_precision_weights=True).to(dev)
rgb = torch.rand(4, 512, 512, 3, device=dev, requires_grad=True)
el = torch.tensor([10., 20., 30., 40.], device=dev); az = torch.tensor([0., 90., 180., 270.], device=dev)
fi = torch.tensor([0, 5, 9, 13], device=dev)
for _ in range(4):
    out = guid(rgb, el, az, torch.full_like(el, 3.8), frame_indices=fi); out["loss_sds"].backward()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    out = guid(rgb, el, az, torch.full_like(el, 3.8), frame_indices=fi); out["loss_sds"].backward()
torch.cuda.synchronize()
print(f"DM4D_MFMA_CONV={os.environ.get('DM4D_MFMA_CONV','1')}: {(time.perf_counter()-t0)/20*1e3:.2f} ms per SDS step, loss {float(out['loss_sds']):.4f}, graph_error {guid._graph_error}")
