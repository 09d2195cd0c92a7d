"""cProfile of the bench step's host side (where do the ~2.3 ms of CPU per step go?)."""
import cProfile, pstats, sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device('cuda:0')
wl = bench.Workload(dev, 0, 1)
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    wl.step()
t1 = time.perf_counter()          # host time to ENQUEUE 30 steps (no sync)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/30:.3f} ms/step, total {1e3*(t2-t0)/30:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    wl.step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
