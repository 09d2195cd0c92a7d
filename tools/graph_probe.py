"""Does the whole step capture into a HIP graph, and what does replay buy?"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
wl = bench.Workload(dev, 0, 1)
for _ in range(20): wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): wl.step()
torch.cuda.synchronize(); t1 = time.perf_counter()
print("eager  %.4f ms/step" % ((t1 - t0) * 5))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): wl.step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = wl.step()
torch.cuda.synchronize()
for _ in range(20): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): g.replay()
torch.cuda.synchronize(); t1 = time.perf_counter()
print("graph  %.4f ms/step" % ((t1 - t0) * 5))
# same result?
ref = wl.step(); torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("color equal:", torch.equal(out["color"], ref["color"]))
