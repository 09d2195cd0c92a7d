import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device('cuda:0')
wl = bench.Workload(dev, 0, 1)
for _ in range(10):
    wl.step()
torch.cuda.synchronize()
ts = []
for _ in range(200):
    t0 = time.perf_counter(); wl.step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("per-step synced ms: min %.3f p50 %.3f p90 %.3f max %.3f" % (ts.min(), np.percentile(ts, 50), np.percentile(ts, 90), ts.max()))
t0 = time.perf_counter()
for _ in range(200):
    wl.step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("unsynced: enqueue %.3f ms/step total %.3f ms/step" % ((t1 - t0) * 5, (t2 - t0) * 5))
import torch.cuda
print(torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9)
