#!/usr/bin/env python
"""bench.py -- rendered views/sec (forward + backward, 512^2, 200k Gaussians) on N MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched with
torch.distributed.run, one rank per GPU.  A "step" = every rank renders FRAMES_PER_STEP x VIEWS_PER_FRAME
= 4 x (4 SDS + 1 reference view) = 20 (frame, view) units of the sugar_dynamic_dg scene (BASELINE.json configs[3]'s per-GPU share) -- each unit is the reference's per-view work:
sparse-control skinning of the mesh at the frame's timestamp, face->Gaussian transform
(custom/threestudio-dreammesh4d/geometry/dynamic_sugar.py:487-613,657-706), RGB rasterizer pass
+ normal rasterizer pass (.../renderer/diff_sugar_rasterizer_temporal.py:169-178,202-211),
forward and backward -- then the ranks all-reduce the parameter-gradient buffer (the one
exchange step of the path).  Units are independent, so ranks shard the
frames with no other collective: per-GPU work is fixed ("weak" scaling) and
`value` = units of all ranks / wall time.

Inputs are synthetic and seeded (dreammesh4d_amd/synthetic.py), resident in HBM before the
timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")   # two OpenMP runtimes (torch, oracle) must not spin against each other

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_FACES = 33_334        # configs[3]: 33,334 faces x 6 Gaussians/face = ~200k mesh-bound Gaussians
N_NODES, K_NBR = 1000, 4
N_FRAMES = 32
H = W = 512
# The partition BASELINE.json configs[3] / SURVEY.md section 8(e) name: rank r takes frames {4r .. 4r+3} x (4 SDS views + the frame's
# reference view) = 20 (frame, view) units per GPU per step -- the headline `value`.  The shipped YAML's own iteration (4 frames x
# (1 SDS + 1 ref view) = 8 units, configs/sugar_dynamic_dg.yaml:9-11,24; what rounds 1-5 timed) is timed beside it: `step_8_views`.
FRAMES_PER_STEP = 4
VIEWS_PER_FRAME = 5          # 4 SDS views + 1 reference view per frame
VIEWS_PER_FRAME_YAML = 2     # 1 SDS view + 1 reference view per frame
K_RENDER_BWD = 5        # kernel id of the dominant kernel (include/dm4d.h)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)      # ~0.45 s timed: 50 steps sit inside the clock / power fluctuation of a box (+-4 %)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-iters", action="store_true", help="skip the dynamic-stage iterations/sec measurement")
    ap.add_argument("--cpu-baseline-views", type=int, default=24)     # ~12 s of host work at ~2 views/s
    ap.add_argument("--views-per-frame", type=int, default=VIEWS_PER_FRAME,
                    help="views per frame of the headline step (default 5 = configs[3]'s 4 SDS + 1 ref; the per-kernel scaling tables use 1/2/4/5)")
    ap.add_argument("--no-step8", action="store_true", help="skip the 8-view step (the shipped YAML's iteration) timed beside the headline")
    ap.add_argument("--no-variants", action="store_true", help="skip the with-depth-gradient / full-backward legs")
    return ap.parse_args()


class Workload:
    """Seeded mesh-bound scene (dreammesh4d_amd/synthetic.py) resident in HBM; this rank's (frame, view)
    units per step.  Node outputs come from the HexPlane+MLP deformation network
    (dreammesh4d_amd/deformation.py, 35.76 M parameters -- the trainable state whose 143 MB of gradients
    are all-reduced), with its zero-initialised heads perturbed (seeded) so the mesh actually moves."""

    def __init__(self, dev, rank, world, views_per_frame=VIEWS_PER_FRAME, sc=None):
        from dreammesh4d_amd import geometry as geo, ops, synthetic as syn, views

        self.dev = dev
        self.views_per_frame = VIEWS_PER_FRAME_ = int(views_per_frame)
        self.views_per_step = VIEWS_PER_STEP = FRAMES_PER_STEP * VIEWS_PER_FRAME_
        sc = self.sc = sc if sc is not None else syn.mesh_bound_scene(N_FACES, n_nodes=N_NODES, k=K_NBR, seed=0)
        T = lambda a: torch.tensor(a, device=dev)
        self.graph = ops.DeformGraph(sc["verts"], sc["nbr_idx"], sc["nbr_w"], N_NODES, dev)
        self.topo = ops.MeshTopology(sc["faces"], len(sc["verts"]), 6, dev)
        verts, faces = T(sc["verts"]), T(sc["faces"])
        self.qs = geo.quaternions(verts, faces, T(sc["complex"]), 6)
        self.scales = geo.scaling(T(sc["log_scales"]), syn.THICKNESS)
        self.opac = geo.strengths(T(sc["densities"]))
        self.rgb = geo.points_rgb(T(sc["sh_dc"]))
        self.N = self.topo.F * 6
        from dreammesh4d_amd.deformation import DeformationNetwork
        torch.manual_seed(0)
        self.net = DeformationNetwork(no_ds=False, no_dr=False, no_do=False).to(dev)     # hybrid: all four heads
        g0 = torch.Generator(device="cpu").manual_seed(11)
        with torch.no_grad():
            for name, p in self.net.named_parameters():
                if "_deform" in name:
                    p.add_((0.02 * torch.randn(p.shape, generator=g0)).to(dev))
        self.net.grads_in_place = True     # persistent HexPlane gradient planes (the step drops its gradients with set_to_none)
        self.nodes = T(sc["nodes"])
        self.timestamps = torch.linspace(0, 1, N_FRAMES + 2)[1:-1].to(dev)            # data/temporal_image.py:155-158
        # rank r renders frames {4r .. 4r+3} (mod L), views_per_frame cameras each
        self.frames = [(FRAMES_PER_STEP * rank + i) % N_FRAMES for i in range(FRAMES_PER_STEP)]
        self.cams, self.unit_frames = [], []
        for fi, fr in enumerate(self.frames):
            for v in range(VIEWS_PER_FRAME_):
                u = (rank * VIEWS_PER_STEP + fi * VIEWS_PER_FRAME_ + v)
                az = -180.0 + 360.0 * (u * 0.61803398875 % 1.0)
                el = -10.0 + 90.0 * (u * 0.41421356237 % 1.0)
                self.cams.append(syn.make_camera(H, W, elev_deg=el, azim_deg=az))
                self.unit_frames.append(fr)
        self.vm = torch.stack([T(c.viewmatrix) for c in self.cams])
        self.pm = torch.stack([T(c.projmatrix) for c in self.cams])
        # unit -> row of this step's frames (int32, what the C ABI takes: an int64 index tensor costs a conversion kernel per step)
        self.fidx = torch.tensor([self.frames.index(f) for f in self.unit_frames], device=dev, dtype=torch.int32)
        self.frame_t = self.timestamps[torch.tensor(self.frames, device=dev)]
        self.bg6 = torch.ones(6, device=dev)
        # DM4D_TILE_RECORDS=1: the experimental (Gaussian, tile) backward records summed in LDS (round 3: correct, but slower
        # on gfx950 -- profiles/r03_tile_records.md); default: the bit-reproducible (Gaussian, cell) records
        self.renderer = views.ViewRenderer(self.graph, self.topo, H, W, self.cams[0].tanfov, method="hybrid",
                                           deterministic=os.environ.get("DM4D_TILE_RECORDS", "0") != "1")
        # record gather + face backward as ONE kernel with a thread per (view, Gaussian): nothing per view is materialised between the
        # rasterizer and the mesh (round 4: -30 us per step; round 3's version looped over a frame's views per thread and lost).  A/B: =0
        self.renderer.fuse_face_backward = os.environ.get("DM4D_FUSE_FACE_BWD", "1") == "1"
        g = torch.Generator(device="cpu").manual_seed(2)
        B = VIEWS_PER_STEP
        self.gC = torch.randn(B, 6, H, W, generator=g).to(dev)
        self.gD = (0.1 * torch.randn(B, 1, H, W, generator=g)).to(dev)
        self.gA = torch.randn(B, 1, H, W, generator=g).to(dev)
        self.render_views = views.render_views
        # The shipped dynamic configuration has NO depth loss (configs/sugar_dynamic_dg.yaml:142-154: lambda_depth, lambda_depth_rel,
        # lambda_depth_tv, lambda_normal_depth_consistency = 0), so in the reference's iteration no gradient reaches the depth
        # image: the headline step feeds none either (the blend backward then writes 32-byte records).  The step WITH a depth
        # gradient is timed beside it (`with_depth_gradient`), and `roofline_full` always has one.
        self.depth_grad = os.environ.get("DM4D_BENCH_DEPTH_GRAD", "0") == "1"
        self._params = list(self.net.parameters())
        # the step object (dreammesh4d_amd/step.py, csrc/step.hip): node network + render_views as one C call each way on
        # persistent buffers -- the same kernels in the same order (tests/test_step_gpu.py: bit-identical), 0.2-0.3 ms of host
        # time per step instead of 0.8.  DM4D_BENCH_STEP_OBJECT=0: the two-operator path (A/B).
        from dreammesh4d_amd.step import DynamicStep
        self.use_step_object = os.environ.get("DM4D_BENCH_STEP_OBJECT", "1") != "0"
        self.dstep = DynamicStep(self.renderer, self.net, self.nodes, self.qs, self.scales, self.opac, self.rgb, self.bg6,
                                 n_views=VIEWS_PER_STEP, n_frames=FRAMES_PER_STEP)
        self.vm16, self.pm16 = self.vm.contiguous(), self.pm.contiguous()

    def set_static_learnable(self, flag):
        """False (the dynamic stage: static_learnable = False, dynamic_sugar.py:79-87): the blend backward neither reduces nor
        records dL/dopacity, dL/d rgb, dL/dscales (lean records).  True: the FULL backward (what sugar_static_refine trains)."""
        for t in (self.scales, self.opac, self.rgb):
            t.requires_grad_(flag)
            t.grad = None

    def step(self):
        # (Module.zero_grad walks the module tree: 0.14 ms of the host's ~0.8 ms per step; the parameter list is fixed)
        for p in self._params:
            p.grad = None
        # node attributes once per distinct timestamp of the step (cached per step in the reference,
        # dynamic_sugar.py:367-405), then broadcast to the views of that frame
        if self.use_step_object and not self.scales.requires_grad:
            out = self.dstep(self.frame_t, self.vm16, self.pm16, self.fidx)
        else:       # (the full backward of `roofline_full`: static appearance learnable)
            dx, dr, ds, do = self.net.node_outputs(self.nodes, self.frame_t)
            out = self.render_views(self.renderer, dx, dr, ds, do, self.qs, self.scales, self.opac, self.rgb, self.vm, self.pm,
                                    self.bg6, frame_index=self.fidx)
        if self.depth_grad:
            torch.autograd.backward([out["color"], out["depth"], out["alpha"]], [self.gC, self.gD, self.gA])
        else:
            torch.autograd.backward([out["color"], out["alpha"]], [self.gC, self.gA])
        return out


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn_ranks(n):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks here, one process per GPU, the way the
    reference's own launcher does (launch.py:114,166,228-235 hands `devices = N` to the trainer, which spawns them).  The
    children are this same command under torch.distributed.run (rendezvous on 127.0.0.1); rank 0 prints the JSON line."""
    import subprocess
    backend = os.environ.get("DM4D_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if backend == "nccl" and have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} HIP device(s) visible (DM4D_BENCH_BACKEND=gloo rehearses the "
                         f"control flow with ranks sharing devices; never a performance number)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            _spawn_ranks(args.gpus)          # does not return
    elif int(os.environ["WORLD_SIZE"]) != args.gpus and "--gpus" in " ".join(sys.argv[1:]):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback in the product path)")
    # DM4D_BENCH_BACKEND=gloo: functional rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks
    # (ranks share devices, the exchange goes through gloo); never a performance number
    backend = os.environ.get("DM4D_BENCH_BACKEND", "nccl")
    dev = torch.device(f"cuda:{local % torch.cuda.device_count() if backend != 'nccl' else local}")
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    from dreammesh4d_amd import _lib
    L = _lib.lib()

    from dreammesh4d_amd.distributed import GradAllReducer, touched_from_plan

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed_leg(wl, steps, warmup, settle, variants):
        """One timed region of `steps` steps of `wl` (+ the exchange), after `warmup` + `settle` untimed steps; with `variants` also
        the same step with a gradient on the depth image and with the static appearance learnable (the other two instantiations
        of the blend backward).  Returns the numbers of the leg; `elapsed` is the MAX over ranks."""
        import ctypes
        wl.step()                                          # builds the HexPlane gather plan of the (static) node set
        # 35.76 M parameters, but the spatial grids only receive gradient at the texels the static nodes touch (the
        # same on every rank): the exchanged message is the touched texels + the time planes + the MLP
        reducer = GradAllReducer(wl.net.parameters(), touched=touched_from_plan(wl.net.deformation_net.grid, wl.net._hex_plan))

        def step():
            wl.step()
            reducer()       # the one exchange step of the path: data-parallel gradient all-reduce (no-op for 1 GPU)

        def region(n):
            L.dm4d_profile_enable(1 << K_RENDER_BWD)
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            sync()
            el = time.perf_counter() - t0
            L.dm4d_profile_enable(0)
            tot = ctypes.c_double(0.0)
            nl = L.dm4d_profile_collect(K_RENDER_BWD, ctypes.byref(tot))
            # seconds of the dominant kernel per STEP (a step's views may go through several launches of it: the pipelined backward)
            return el, tot.value / n * 1e-3, int(nl)

        for _ in range(warmup):
            step()
        sync()
        # settle: a FIXED number of extra untimed steps (the same count on every rank, so the ranks stay in lockstep), so
        # that the clock / power state the device idled into while the host built the scene does not leak into the timed region
        for _ in range(settle):
            step()
        sync()
        D_views = wl.renderer.check()      # also validates the duplicate-list capacity
        elapsed, avg_s, n_launch = region(steps)
        wl.renderer.check()
        if world > 1:
            t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        leg = {"elapsed": elapsed, "steps": steps, "avg_s": avg_s, "n_launch": n_launch, "D_mean": float(np.mean(D_views)),
               "reducer": reducer, "with_depth": None, "full": None}
        if variants and world == 1 and not wl.depth_grad:      # the same step with a gradient on the depth image (48-byte records)
            wl.depth_grad = True
            for _ in range(5):
                step()
            sync()
            n_wd = min(steps, 100)
            el, av, nl = region(n_wd)
            leg["with_depth"] = (el / n_wd, av, nl)
            # ... and with the static appearance learnable: the FULL (non-lean) blend backward
            wl.set_static_learnable(True)
            for _ in range(5):
                step()
            sync()
            n_full = min(steps, 20)
            el, av, nl = region(n_full)
            leg["full"] = (el / n_full, av, nl)
            wl.set_static_learnable(False)
            wl.depth_grad = False
        return leg

    def pmc_traffic(kernel, n_views):
        """HBM traffic of the dominant kernel per launch from the committed PMC passes of this same workload (profiles/
        r06_pmc_traffic_<views>v.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of `bench.py --no-step8
        --no-variants --views-per-frame .`, FETCH_SIZE x2 per MI355X_MICROARCH.md's gfx950 correction).  The workload is seeded,
        so the figure is launch-invariant; null when no file for this shape is committed."""
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", f"r06_pmc_traffic_{n_views}v.json")))
            return round(pmc[f"dm4d::{kernel}"]["bytes_corrected"])
        except Exception:
            return None

    def leg_numbers(wl, leg):
        """views/s, the dominant kernel's roofline and the whole-view fraction of one leg."""
        N, B = wl.N, wl.views_per_step
        D_mean = leg["D_mean"]
        value = world * B * leg["steps"] / leg["elapsed"]
        # Dominant kernel: k_render_bwd<6, lean> (the wide blocks of the long cells first, then the quadrants), ONE launch per
        # step covering the step's views with the RGB and the normal pass fused.  Algorithmic bytes per launch (section 8d,
        # render-bwd row of B_b, credited per reference pass): views x 2 passes x (48 B/duplicate + 40 B/pixel + 44 B/Gaussian).
        alg_bytes = B * 2.0 * (48.0 * D_mean + 40.0 * H * W + 44.0 * N)
        achieved = alg_bytes / leg["avg_s"] / 1e9 if leg["avg_s"] > 0 else 0.0
        # whole-view algorithmic bytes: B_view = 2 (B_f + B_b) + B_skin (section 8d)
        V = len(wl.sc["verts"])
        b_view = 2 * ((104 * N + 84 * D_mean + 28 * H * W) + (228 * N + 48 * D_mean + 40 * H * W)) + 40 * V + 28 * N + 12288 * N_NODES
        kern = "k_render_bwd<6, 1>" if wl.depth_grad else "k_render_bwd<6, 2>"
        roof = {"bound": "hbm", "kernel": kern + " (entry-parallel blend backward, one launch per step batched over its views: wide blocks for the long cells first, then the quadrants; "
                                          + ("48-byte lean records)" if wl.depth_grad else "32-byte lean records: no depth gradient)"),
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": pmc_traffic(kern, B), "alg_bytes_per_launch": round(alg_bytes), "avg_launch_us": round(leg["avg_s"] * 1e6, 2),
                "launches_timed": leg["n_launch"], "launches_per_step": round(leg["n_launch"] / leg["steps"], 2),
                "per": "step: `alg_bytes_per_launch` / `avg_launch_us` / `traffic` are per STEP = per launch (ONE launch of this kernel covers the step's views; "
                       "timed with HIP events on its stream inside the library)"}
        out = {"value": round(value, 3), "ms_per_step": round(leg["elapsed"] / leg["steps"] * 1e3, 4), "views_per_step_per_gpu": B,
               "mean_duplicates_D": round(D_mean), "whole_view_frac_of_hbm_roofline": round(value / world * b_view / (HBM_PEAK_GBS * 1e9), 5),
               "roofline": roof}
        if leg["full"] is not None:
            full = leg["full"]
            fa = alg_bytes / full[1] / 1e9 if full[1] > 0 else 0.0
            out["roofline_full"] = {"bound": "hbm", "kernel": "k_render_bwd<6, 0> (static appearance learnable AND a depth gradient: dL/dopacity, dL/d rgb, dL/dscales reduced and recorded too, 64-byte records)",
                                    "achieved": round(fa, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fa / HBM_PEAK_GBS, 5),
                                    "alg_bytes_per_launch": round(alg_bytes), "avg_launch_us": round(full[1] * 1e6, 2), "launches_timed": full[2],
                                    "ms_per_step": round(full[0] * 1e3, 4), "views_per_s": round(B / full[0], 1)}
        if leg["with_depth"] is not None:
            wd = leg["with_depth"]
            wa = alg_bytes / wd[1] / 1e9 if wd[1] > 0 else 0.0
            out["with_depth_gradient"] = {"kernel": "k_render_bwd<6, 1> (the same step with a gradient on the depth image: 48-byte lean records)",
                                          "ms_per_step": round(wd[0] * 1e3, 4), "views_per_s": round(B / wd[0], 1),
                                          "avg_launch_us": round(wd[1] * 1e6, 2), "frac": round(wa / HBM_PEAK_GBS, 5), "launches_timed": wd[2]}
        return out

    vpf = int(args.views_per_frame)
    wl = Workload(dev, rank, world, views_per_frame=vpf)
    settle = 400 if vpf <= 2 else 200       # ~0.4-0.8 s of untimed GPU work either way
    leg = timed_leg(wl, args.steps, args.warmup, settle, variants=not args.no_variants)
    reducer = leg["reducer"]
    head = leg_numbers(wl, leg)
    # the shipped YAML's iteration (8 views per step: what rounds 1-5 reported as `value`), on the same scene and the same box
    step8 = None
    if world == 1 and not args.no_step8 and vpf != VIEWS_PER_FRAME_YAML:
        wl8 = Workload(dev, rank, world, views_per_frame=VIEWS_PER_FRAME_YAML, sc=wl.sc)
        leg8 = timed_leg(wl8, args.steps, args.warmup, 400, variants=not args.no_variants)
        step8 = leg_numbers(wl8, leg8)
        step8["note"] = ("4 frames x (1 SDS + 1 reference view) = 8 units per step: the shipped configuration's own iteration "
                         "(configs/sugar_dynamic_dg.yaml:9-11,24) and the step rounds 1-5 reported as `value`; same scene, same box, same run")
        del wl8, leg8

    if rank == 0:
        N = wl.N
        units = "4 SDS views + 1 reference view" if vpf == 5 else ("1 SDS view + 1 reference view" if vpf == 2 else f"{vpf} views")
        out = {
            "metric": "rendered views/sec (fwd+bwd, 512^2, 200k Gaussians)",
            "value": head["value"], "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if backend == "nccl" else f"synthetic (REHEARSAL over {backend}: not a performance number)",
            "config": {"workload": f"sugar_dynamic_dg, BASELINE configs[3] per-GPU share (SURVEY.md 8e): {FRAMES_PER_STEP} frames x ({units}) = "
                                   f"{wl.views_per_step} (frame, view) units per GPU per step; mesh-bound {N} Gaussians "
                                   f"({wl.topo.F} faces x 6), {N_NODES} graph nodes K={K_NBR} hybrid LBS/DQS skinning, "
                                   f"512x512; per view: skinning + face->Gaussian + RGB pass + normal pass, fwd+bwd"
                                   + ("" if wl.depth_grad else "; no gradient on the depth image, as in the shipped configuration (no depth loss)"),
                       "comparability": "rounds 1-5 reported the 8-unit step (4 frames x 2 views: the shipped YAML's iteration) as `value`: it is `step_8_views` "
                                        "below, timed in this same run; from round 6 on `value` is the 20-unit step BASELINE configs[3] names.  Rounds 1-2 timed "
                                        "WITH a gradient on the depth image (48-byte records): they compare with `step_8_views.with_depth_gradient`",
                       "views_per_step_per_gpu": wl.views_per_step, "untimed_settle_steps": settle, "frames_per_step_per_gpu": FRAMES_PER_STEP,
                       "views_per_frame": vpf,
                       "mean_duplicates_D": head["mean_duplicates_D"], "allreduce_bytes_per_step": reducer.nbytes if world > 1 else 0,
                       "allreduce_message_bytes": reducer.nbytes, "dense_gradient_bytes": 4 * reducer.dense_elements,
                       "whole_view_frac_of_hbm_roofline": head["whole_view_frac_of_hbm_roofline"],
                       "parallelism": f"dp{world} (frames sharded, 1 grad all-reduce/step)" if world > 1 else "single GPU"},
            "roofline": head["roofline"],
        }
        for k in ("roofline_full", "with_depth_gradient"):
            if k in head:
                out[k] = head[k]
        if step8 is not None:
            out["step_8_views"] = step8
        if world == 1 and not args.no_iters:
            # BASELINE.json's second metric, reported beside the headline one (never used for `value`)
            try:
                out["config"].update(dynamic_stage_iterations(wl, dev))
            except Exception as e:     # the headline line must not depend on it
                out["config"]["dynamic_stage_iters_per_sec"] = None
                out["config"]["dynamic_stage_error"] = repr(e)[:200]
            # ... and at the partition of the headline step (BASELINE configs[3]: 4 SDS views + the reference view per frame = 20 views per
            # iteration, UNet batch 32 / VAE encoder batch 16)
            try:
                out["config"].update(dynamic_stage_iterations(wl, dev, n=20, sds_views_per_frame=VIEWS_PER_FRAME - 1, key="dynamic_stage_20_units"))
            except Exception as e:
                out["config"]["dynamic_stage_20_units_iters_per_sec"] = None
                out["config"]["dynamic_stage_20_units_error"] = repr(e)[:200]
        if world == 1 and not args.no_iters:
            try:
                out["config"].update(static_stage_iterations(dev))
            except Exception as e:
                out["config"]["static_stage_iters_per_sec"] = None
                out["config"]["static_stage_error"] = repr(e)[:200]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_baseline_views)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def dynamic_stage_iterations(wl, dev, n=50, sds_views_per_frame=VIEWS_PER_FRAME_YAML - 1, key="dynamic_stage"):
    """dynamic-stage iterations/sec at the same scene: 4 frames x (1 reference + `sds_views_per_frame` SDS views) per iteration, HexPlane
    network, render, losses, full-size Zero123 SDS (SD-1.x UNet 860 M + VAE encoder, fp16, RANDOM weights: the
    checkpoint is not in the tree), backward, AdamW over the 35.76 M parameters
    (custom/threestudio-dreammesh4d/system/sugar_4dgen.py:397-429)."""
    from dreammesh4d_amd import synthetic as syn, zero123 as z
    from dreammesh4d_amd.dynamic_stage import DynamicStage

    model = _zero123_model(dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    guid = z.TemporalStableZero123Guidance(model, torch.randn(N_FRAMES, 1, 768, generator=g),
                                           torch.randn(N_FRAMES, 4, 32, 32, generator=g), cond_elevation_deg=5.0,
                                           half_precision_weights=True).to(dev)
    static = {"q_static": wl.qs, "scales": wl.scales, "opacities": wl.opac, "rgb": wl.rgb}
    ref_img = torch.rand(N_FRAMES, H, W, 3, generator=g).to(dev)
    ref_mask = (torch.rand(N_FRAMES, H, W, 1, generator=g) > 0.5).float().to(dev)
    from dreammesh4d_amd.mesh_reg import ARAPCoach, MeshNormalConsistency

    stage = DynamicStage(wl.renderer, wl.net, wl.nodes, static, wl.timestamps, ref_img, ref_mask,
                         syn.make_camera(H, W, elev_deg=5.0, azim_deg=0.0), guidance=guid, frames_per_step=FRAMES_PER_STEP,
                         random_views_per_frame=sds_views_per_frame,
                         normal_consistency=MeshNormalConsistency(wl.sc["faces"], len(wl.sc["verts"]), dev),
                         arap=ARAPCoach(wl.sc["verts"], wl.sc["faces"], dev), milestone_arap_reg=0)
    # the first iteration runs STRICT: a 3x3 convolution, q/k/v projection, supported attention, GroupNorm / add / GEGLU of the
    # guidance step that falls back to the library raises (zero123._library_fallback, fused_norm.expect_fused) instead of
    # silently costing the step its hand-written kernels
    os.environ["DM4D_STRICT_FUSED"] = "1"
    try:
        stage.iteration()
    finally:
        os.environ.pop("DM4D_STRICT_FUSED", None)
    for _ in range(3):
        stage.iteration()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n):
        stage.iteration()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return {f"{key}_iters_per_sec": round(n / dt, 3), f"{key}_ms_per_iteration": round(1e3 * dt / n, 2),
            f"{key}_note": f"{FRAMES_PER_STEP * (1 + sds_views_per_frame)} views/iteration (4 frames x (1 reference + {sds_views_per_frame} SDS views)), full-size Zero123 fp16 with random weights, mesh normal consistency + key-frame ARAP, AdamW step "
                                  "(the reference's effective betas (0.9, 0.999), no decay) included; no backward of the normal pass: no loss of the shipped "
                                  "configuration reads the normal image (the headline step above keeps both passes' backward)"}


_ZERO123 = {}


def _zero123_model(dev):
    """The full-size random-weight Zero123 (860 M UNet + VAE encoder), built once and shared by both stage legs."""
    from dreammesh4d_amd import zero123 as z

    if "m" not in _ZERO123:
        with torch.device(dev):
            _ZERO123["m"] = z.Zero123()
    return _ZERO123["m"]


def static_stage_iterations(dev, n=50):
    """static-stage iterations/sec at BASELINE configs[1] (sugar_static_refine: 50,004 mesh-bound Gaussians, 512 x 512): one
    reference view + `random_camera.batch_size` = 4 random views per iteration (configs/sugar_static_refine.yaml:18-28), every
    SuGaR parameter learnable -- the FULL blend backward, k_render_bwd<6, 0> -- rgb / mask MSE, full-size Zero123 SDS on the 4 random
    views, mesh normal consistency + Laplacian smoothing, rgb / depth / normal total variation (so a gradient reaches the depth
    image), AdamW over the six parameter groups (custom/threestudio-dreammesh4d/system/sugar_static.py:110-340)."""
    from dreammesh4d_amd import renderer as R, sugar, synthetic as syn, zero123 as z
    from dreammesh4d_amd.mesh_reg import MeshLaplacianSmoothing, MeshNormalConsistency
    from dreammesh4d_amd.static_stage import StaticStage

    sc = syn.mesh_bound_scene(8334, n_nodes=50, k=4, seed=0)
    V = len(sc["verts"])
    geo = sugar.SuGaR(sc["verts"], sc["faces"], vertex_colors=np.random.default_rng(0).random((V, 3)), device=dev, position_lr=0.00048,
                      scaling_lr=0.005, feature_lr=0.001, opacity_lr=0.02, rotation_lr=0.001, spatial_lr_scale=1.0)
    g = torch.Generator(device="cpu").manual_seed(0)
    guid = z.StableZero123Guidance(_zero123_model(dev), torch.randn(1, 1, 768, generator=g), torch.randn(1, 4, 32, 32, generator=g),
                                   cond_elevation_deg=5.0, half_precision_weights=True).to(dev)
    ref_img = torch.rand(1, H, W, 3, generator=g).to(dev)
    ref_mask = (torch.rand(1, H, W, 1, generator=g) > 0.5).float().to(dev)
    stage = StaticStage(geo, R.DiffSuGaRNormal(geo), ref_img, ref_mask, H, W, guidance=guid, random_views=4,
                        normal_consistency=MeshNormalConsistency(sc["faces"], V, dev), laplacian_smoothing=MeshLaplacianSmoothing(sc["faces"], V, dev))
    for _ in range(4):
        stage.iteration()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n):
        stage.iteration()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return {"static_stage_iters_per_sec": round(n / dt, 3), "static_stage_ms_per_iteration": round(1e3 * dt / n, 2),
            "static_stage_note": f"sugar_static_refine (configs[1]): {geo.n_gaussians} Gaussians, 1 reference + 4 random views of 512 x 512 per iteration, "
                                 "every SuGaR parameter learnable (full blend backward), full-size Zero123 fp16 with random weights, normal consistency + "
                                 "Laplacian + rgb / depth / normal TV, AdamW step included"}


def cpu_baseline(wl, n_views):
    """The oracle (CPU restatement of the reference algorithm: PyTorch-CPU skinning, C/OpenMP rasterizer)
    timed on this box's host cores on a bounded sample: the first `n_views` units of the same workload."""
    from oracle import raster as orc, skinning as sk

    torch.set_num_threads(min(16, os.cpu_count()))   # skinning tensors are small; more threads only contend
    sc = wl.sc
    t = lambda a: torch.tensor(np.asarray(a))
    verts, faces, idx, w = t(sc["verts"]), t(sc["faces"]), t(sc["nbr_idx"]), t(sc["nbr_w"])
    qs = sk.static_quaternions(verts.double(), faces, t(sc["complex"]).double()).float()
    scales, opac, rgb = sk.static_attributes(t(sc["log_scales"]), t(sc["densities"]), t(sc["sh_dc"]), 3.8e-6)
    gC, gD, gA = wl.gC.cpu().numpy(), wl.gD.cpu().numpy(), wl.gA.cpu().numpy()

    with torch.no_grad():
        raw = [x.detach().cpu() for x in wl.net.node_outputs(wl.nodes, wl.frame_t)]

    def unit(u):
        cam, f = wl.cams[u], int(wl.fidx[u])
        leaves = [raw[0][f].clone().requires_grad_(True), raw[1][f].clone().requires_grad_(True),
                  raw[2][f].clone().requires_grad_(True), raw[3][f].clone().unsqueeze(-1).requires_grad_(True)]
        trans, q, S, op = sk.node_attributes(*leaves)
        xyz, vrot = sk.skin_vertices(verts, idx, w, trans, q, S, op, "hybrid")
        means, rots, normals = sk.face_gaussians(xyz, vrot, faces, qs)
        g_means, g_rots, g_nrm = 0, 0, None
        for colors, g_c, g_d, g_a in ((rgb.numpy(), gC[u, :3], gD[u, 0] if wl.depth_grad else None, gA[u, 0]), (normals.detach().numpy(), gC[u, 3:], None, None)):
            o = orc.RasterOracle(image_height=H, image_width=W, tanfovx=cam.tanfov, tanfovy=cam.tanfov, bg=(1, 1, 1),
                                 scale_modifier=1.0, viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix,
                                 campos=cam.campos)
            o.forward(means.detach().numpy(), opac.view(-1).numpy(), colors_precomp=colors, scales=scales.numpy(),
                      rotations=rots.detach().numpy())
            g = o.backward(g_c, g_d, g_a)
            g_means = g_means + g["dL_dmeans3D"]
            g_rots = g_rots + g["dL_drots"]
            g_nrm = g["dL_dcolors"]
        torch.autograd.backward([means, rots, normals], [t(g_means), t(g_rots), t(g_nrm)])

    unit(0)  # warm-up (page-in, thread pools)
    t0 = time.perf_counter()
    for v in range(n_views):
        unit(v % len(wl.cams))
    dt = time.perf_counter() - t0
    return {"value": round(n_views / dt, 4), "unit": "views/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{n_views} of the same (frame, view) units, HexPlane query excluded (skinning + face->Gaussian in PyTorch-CPU on "
                      f"{min(16, os.cpu_count())} threads, RGB + normal pass fwd+bwd in the C/OpenMP oracle on "
                      f"{os.cpu_count()} threads)"}


if __name__ == "__main__":
    main()
