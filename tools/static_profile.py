"""Where a static-stage iteration's time goes (BASELINE configs[1], bench.py's static_stage_iterations setup): wall time of
an iteration with / without the Zero123 SDS term, and the torch profiler's per-kernel and per-host-op tables for the
non-SDS iteration.  Runs on the GPU box:  python tools/static_profile.py [--no-guidance] [--trace]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402


def build(dev, with_guidance):
    from dreammesh4d_amd import renderer as R, sugar, synthetic as syn, zero123 as z
    from dreammesh4d_amd.mesh_reg import MeshLaplacianSmoothing, MeshNormalConsistency
    from dreammesh4d_amd.static_stage import StaticStage

    H = W = bench.H
    sc = syn.mesh_bound_scene(8334, n_nodes=50, k=4, seed=0)
    V = len(sc["verts"])
    geo = sugar.SuGaR(sc["verts"], sc["faces"], vertex_colors=np.random.default_rng(0).random((V, 3)), device=dev, position_lr=0.00048,
                      scaling_lr=0.005, feature_lr=0.001, opacity_lr=0.02, rotation_lr=0.001, spatial_lr_scale=1.0)
    g = torch.Generator(device="cpu").manual_seed(0)
    guid = None
    if with_guidance:
        guid = z.StableZero123Guidance(bench._zero123_model(dev), torch.randn(1, 1, 768, generator=g), torch.randn(1, 4, 32, 32, generator=g),
                                       cond_elevation_deg=5.0, half_precision_weights=True).to(dev)
    ref_img = torch.rand(1, H, W, 3, generator=g).to(dev)
    ref_mask = (torch.rand(1, H, W, 1, generator=g) > 0.5).float().to(dev)
    return StaticStage(geo, R.DiffSuGaRNormal(geo), ref_img, ref_mask, H, W, guidance=guid, random_views=4,
                       normal_consistency=MeshNormalConsistency(sc["faces"], V, dev), laplacian_smoothing=MeshLaplacianSmoothing(sc["faces"], V, dev))


def timed(stage, dev, n):
    for _ in range(4):
        stage.iteration()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n):
        stage.iteration()
    torch.cuda.synchronize(dev)
    return 1e3 * (time.perf_counter() - t0) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=30)
    ap.add_argument("--with-guidance", action="store_true")
    ap.add_argument("--tables", action="store_true")
    ap.add_argument("--sync-debug", action="store_true", help="every implicit host <-> device synchronisation of one iteration, with its Python stack")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    stage = build(dev, a.with_guidance)
    print(f"ms/iteration ({'with' if a.with_guidance else 'without'} Zero123 SDS): {timed(stage, dev, a.n):.2f}")
    # host-only share: time to ENQUEUE an iteration when the queue is empty
    torch.cuda.synchronize(dev)
    hs = []
    for _ in range(10):
        t0 = time.perf_counter()
        stage.iteration()
        hs.append(1e3 * (time.perf_counter() - t0))
        torch.cuda.synchronize(dev)
    print(f"host time to return from iteration() on an empty queue: median {sorted(hs)[5]:.2f} ms")
    if a.sync_debug:
        import traceback
        import warnings

        def show(msg, cat, fn, ln, *rest):
            print(f"SYNC at {fn}:{ln}: {msg}")
            traceback.print_stack(limit=10)
        warnings.showwarning = show
        warnings.simplefilter("always")
        torch.cuda.set_sync_debug_mode("warn")
        stage.iteration()
        torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize(dev)
    if a.tables:
        from torch.profiler import ProfilerActivity, profile

        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for _ in range(5):
                stage.iteration()
            torch.cuda.synchronize(dev)
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=70))
        print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=70))


if __name__ == "__main__":
    main()
