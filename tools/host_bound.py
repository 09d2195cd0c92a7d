"""Is the bench step bound by the host?  The host's time to ENQUEUE a burst of steps onto an EMPTY stream (10 steps ~ 210
launches: far below the depth of the hardware queue, so no back-pressure from the device leaks into the number -- over 400 steps
the launch calls block on the full queue and "enqueue time" converges to the GPU's own pace whatever the host costs) beside the
wall time per step of 400 back-to-back steps.  Both host paths: the step object (dreammesh4d_amd/step.py) and the two-operator
path (node_outputs + render_views).  --profile: cProfile of the burst."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
wl = bench.Workload(dev, 0, 1)
K = 10
for name, flag in (("step object", True), ("two operators", False), ("step object", True)):
    wl.use_step_object = flag
    for _ in range(100): wl.step()
    torch.cuda.synchronize()
    bursts = []
    for rep in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K): wl.step()
        bursts.append((time.perf_counter() - t0) / K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(400): wl.step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 400
    bursts.sort()
    print(f"{name:14s} host enqueue (empty queue, median of 20 bursts of {K}) {1e3*bursts[10]:.4f} ms/step [min {1e3*bursts[0]:.4f}], wall {1e3*wall:.4f} ms/step")
if "--profile" in sys.argv:
    import cProfile, pstats
    for flag in (True, False):
        wl.use_step_object = flag
        for _ in range(20): wl.step()
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        for rep in range(10):
            torch.cuda.synchronize()
            pr.enable()
            for _ in range(K): wl.step()
            pr.disable()
        print("==== step object" if flag else "==== two operators")
        pstats.Stats(pr).sort_stats("tottime").print_stats(18)
