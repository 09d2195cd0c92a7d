"""Is the bench step bound by the host?  Enqueue time of 400 steps (before the device sync) beside the wall time of the same 400
steps: enqueue ~ wall means the Python / launch path is the bottleneck and GPU savings do not show in the step time."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
wl = bench.Workload(dev, 0, 1)
for _ in range(300): wl.step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(400): wl.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0)/400:.4f} ms/step, wall {1e3*(t2-t0)/400:.4f} ms/step")
