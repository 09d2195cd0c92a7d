"""Single-GPU stand-in for the N > 1 step: step + pack + RCCL all-reduce (world 1) + unpack."""
import os, sys, time, torch
sys.path.insert(0, '/root/repo')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch.distributed as dist
import bench
from dreammesh4d_amd.distributed import GradAllReducer, touched_from_plan
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
wl = bench.Workload(dev, 0, 1)
wl.step()
for name, touched in (("dense", None), ("touched", touched_from_plan(wl.net.deformation_net.grid, wl.net._hex_plan))):
    red = GradAllReducer(wl.net.parameters(), touched=touched)
    def full():
        wl.step(); red.pack(); dist.all_reduce(red.flat); red.unpack(1.0)
    for _ in range(20): full()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): full()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(100): wl.step()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: message {red.nbytes/1e6:.1f} MB  step+exchange {(t1-t0)*10:.3f} ms  step only {(t2-t1)*10:.3f} ms")
dist.destroy_process_group()
