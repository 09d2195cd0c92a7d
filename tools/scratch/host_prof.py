import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
wl = bench.Workload(dev, 0, 1)
from dreammesh4d_amd.distributed import GradAllReducer, touched_from_plan
wl.step()
reducer = GradAllReducer(wl.net.parameters(), touched=touched_from_plan(wl.net.deformation_net.grid, wl.net._hex_plan))
for _ in range(300): wl.step(); reducer()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): wl.step(); reducer()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(30)
