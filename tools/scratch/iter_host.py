import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["x"]
import runpy, io, contextlib
ns = runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools/iters_per_sec.py"))
stage, torch = ns["stage"], ns["torch"]
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(20): stage.iteration()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0)/20:.2f} ms/iteration, wall {1e3*(t2-t0)/20:.2f} ms/iteration")
pr = cProfile.Profile(); pr.enable()
for _ in range(20): stage.iteration()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
