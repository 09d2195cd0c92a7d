#!/usr/bin/env python
"""bench.py -- rendered views/sec (forward + backward, 512^2, 200k Gaussians) on N MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched with
torch.distributed.run, one rank per GPU.  A "step" = every rank renders VIEWS_PER_STEP
(frame, view) units -- each unit is the reference's per-view work: RGB rasterizer pass +
normal rasterizer pass, forward and backward
(custom/threestudio-dreammesh4d/renderer/diff_sugar_rasterizer_temporal.py:169-178,202-211) --
then the ranks all-reduce the parameter-gradient buffer (the one exchange step of the path,
SURVEY.md section 8e).  Units are independent, so ranks shard them with no other collective:
per-GPU work is fixed ("weak" scaling) and `value` = units of all ranks / wall time.

Inputs are synthetic and seeded (dreammesh4d_amd/synthetic.py), resident in HBM before the
timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_GAUSS = 200_000
H = W = 512
VIEWS_PER_STEP = 8      # the reference's per-rank iteration: 4 frames x (1 SDS view + 1 ref view)
K_RENDER_BWD = 5        # kernel id of the dominant kernel (include/dm4d.h)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-views", type=int, default=4)
    return ap.parse_args()


def build_workload(dev, rank):
    from dreammesh4d_amd import synthetic as syn
    import dreammesh4d_amd.diff_gaussian_rasterization as dgr

    sc = syn.random_splat_scene(N_GAUSS, seed=0)
    rng = np.random.default_rng(1)
    nrm = rng.normal(size=(N_GAUSS, 3))
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    T = lambda a, rg=False: torch.tensor(a, device=dev).requires_grad_(rg)
    P = {"means3D": T(sc["means3D"], True), "opac": T(sc["opacities"][:, None], True), "colors": T(sc["colors"], True),
         "scales": T(sc["scales"], True), "rots": T(sc["rotations"], True), "normals": T(nrm, True)}
    # cameras: this rank's (frame, view) units; azimuths differ per rank like per-rank seeds in the reference
    cams = []
    for v in range(VIEWS_PER_STEP):
        az = -180.0 + 360.0 * ((rank * VIEWS_PER_STEP + v) * 0.61803398875 % 1.0)
        el = -10.0 + 90.0 * ((rank * VIEWS_PER_STEP + v) * 0.41421356237 % 1.0)
        cam = syn.make_camera(H, W, elev_deg=el, azim_deg=az)
        rs = dgr.GaussianRasterizationSettings(H, W, cam.tanfov, cam.tanfov, T(np.ones(3, np.float32)), 1.0,
                                               T(cam.viewmatrix), T(cam.projmatrix), 0, T(cam.campos), False, False)
        cams.append((cam, dgr.GaussianRasterizer(rs)))
    g = torch.Generator(device="cpu").manual_seed(2)
    grads = {"rgb": torch.randn(3, H, W, generator=g).to(dev), "alpha": torch.randn(1, H, W, generator=g).to(dev),
             "depth": (0.1 * torch.randn(1, H, W, generator=g)).to(dev), "normal": torch.randn(3, H, W, generator=g).to(dev)}
    return sc, nrm, P, cams, grads


def render_unit(P, rast, grads):
    """One (frame, view) unit: RGB pass + normal pass, forward and backward."""
    m2 = torch.zeros_like(P["means3D"], requires_grad=True)
    color, radii, depth, alpha = rast(means3D=P["means3D"], means2D=m2, opacities=P["opac"],
                                      colors_precomp=P["colors"], scales=P["scales"], rotations=P["rots"])
    normal, _, _, _ = rast(means3D=P["means3D"], means2D=torch.zeros_like(m2), opacities=P["opac"],
                           colors_precomp=P["normals"], scales=P["scales"], rotations=P["rots"])
    torch.autograd.backward([color, depth, alpha, normal], [grads["rgb"], grads["depth"], grads["alpha"], grads["normal"]])


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback in the product path)")
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    from dreammesh4d_amd import _lib
    import dreammesh4d_amd.diff_gaussian_rasterization as dgr
    L = _lib.lib()

    sc, nrm, P, cams, grads = build_workload(dev, rank)
    params = list(P.values())

    def step():
        for p in params:
            p.grad = None
        for _, rast in cams:
            render_unit(P, rast, grads)
        if world > 1:
            flat = torch.cat([p.grad.reshape(-1) for p in params])
            dist.all_reduce(flat)
            flat.mul_(1.0 / world)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync()
    L.dm4d_profile_enable(1 << K_RENDER_BWD)
    dgr.LAST_NUM_RENDERED.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    L.dm4d_profile_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    import ctypes
    tot_ms = ctypes.c_double(0.0)
    n_launch = L.dm4d_profile_collect(K_RENDER_BWD, ctypes.byref(tot_ms))
    D_mean = float(np.mean(dgr.LAST_NUM_RENDERED)) if dgr.LAST_NUM_RENDERED else 0.0

    if rank == 0:
        units = world * VIEWS_PER_STEP * args.steps
        value = units / elapsed
        # dominant kernel: render_bwd.  Algorithmic bytes per launch (render-bwd row):
        # 48 B per duplicate (id + attributes) + 40 B per pixel (grads, state, colour) + 44 B per Gaussian (grads)
        alg_bytes = 48.0 * D_mean + 40.0 * H * W + 44.0 * N_GAUSS
        avg_s = (tot_ms.value / max(n_launch, 1)) * 1e-3
        achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        out = {
            "metric": "rendered views/sec (fwd+bwd, 512^2, 200k Gaussians)",
            "value": round(value, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "random-splat scene, 200k Gaussians, 512x512, per view: RGB pass + normal pass "
                                   "fwd+bwd through the drop-in GaussianRasterizer (1 host sync per pass, as upstream)",
                       "views_per_step_per_gpu": VIEWS_PER_STEP, "mean_duplicates_D": round(D_mean),
                       "parallelism": f"dp{world} (units sharded, 1 grad all-reduce/step)" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "k_render_bwd", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "alg_bytes_per_launch": round(alg_bytes), "avg_launch_us": round(avg_s * 1e6, 2),
                         "launches_timed": int(n_launch)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, nrm, cams, grads, args.cpu_baseline_views)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(sc, nrm, cams, grads, n_views):
    """The oracle (CPU restatement of the reference algorithm, OpenMP over tiles / Gaussians) timed on
    this box's host cores on a bounded sample: the first `n_views` units of the same workload."""
    from oracle import raster as orc

    gC, gA = grads["rgb"].cpu().numpy(), grads["alpha"].cpu().numpy()[0]
    gD, gN = grads["depth"].cpu().numpy()[0], grads["normal"].cpu().numpy()

    def unit(cam):
        for colors, g_c, g_d, g_a in ((sc["colors"], gC, gD, gA), (nrm, gN, None, None)):
            o = orc.RasterOracle(image_height=H, image_width=W, tanfovx=cam.tanfov, tanfovy=cam.tanfov, bg=(1, 1, 1),
                                 scale_modifier=1.0, viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix,
                                 campos=cam.campos)
            o.forward(sc["means3D"], sc["opacities"], colors_precomp=colors, scales=sc["scales"],
                      rotations=sc["rotations"])
            o.backward(g_c, g_d, g_a)

    unit(cams[0][0])  # warm-up (page-in, thread pool)
    t0 = time.perf_counter()
    for v in range(n_views):
        unit(cams[v % len(cams)][0])
    dt = time.perf_counter() - t0
    return {"value": round(n_views / dt, 4), "unit": "views/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{n_views} of the same (frame, view) units (RGB+normal pass fwd+bwd), C oracle with OpenMP "
                      f"over tiles/Gaussians, {os.cpu_count()} host threads"}


if __name__ == "__main__":
    main()
