import sys, torch
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device('cuda:0')
print("start", torch.cuda.memory_allocated() / 1e9)
wl = bench.Workload(dev, 0, 1)
print("after workload", torch.cuda.memory_allocated() / 1e9)
for i in range(6):
    wl.step(); torch.cuda.synchronize()
    print("step", i, "alloc %.3f GB reserved %.3f GB" % (torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9))
r = wl.renderer
print("capacity", r.capacity, "record_capacity", r.record_capacity, "pool", len(r._ws_pool), [k for k, _ in r._ws_pool])
from dreammesh4d_amd import _lib
L = _lib.lib()
print("views binning bytes GB", L.dm4d_views_binning_bytes(8, r.capacity) / 1e9, "grad GB", L.dm4d_views_grad_bytes(8, r.record_capacity) / 1e9, "geom GB", L.dm4d_views_geom_bytes(8, r.N, 512, 512) / 1e9)
import gc
n = gc.collect(); torch.cuda.synchronize()
print("after gc.collect (%d objects): alloc %.3f GB" % (n, torch.cuda.memory_allocated() / 1e9))
for i in range(3):
    wl.step(); torch.cuda.synchronize()
    print("step", i, "alloc %.3f GB" % (torch.cuda.memory_allocated() / 1e9))
# which tensors are alive?
import collections
sizes = collections.Counter()
for o in gc.get_objects():
    try:
        if torch.is_tensor(o) and o.is_cuda:
            sizes[(tuple(o.shape), str(o.dtype))] += 1
    except Exception:
        pass
for k, v in sorted(sizes.items(), key=lambda kv: -kv[1])[:15]:
    print(v, k)
seen = {}
for o in gc.get_objects():
    try:
        if torch.is_tensor(o) and o.is_cuda:
            st = o.untyped_storage()
            seen[st.data_ptr()] = (st.nbytes(), tuple(o.shape))
    except Exception:
        pass
tot = sum(v[0] for v in seen.values())
print("unique live storages: %d, %.3f GB (allocator says %.3f GB)" % (len(seen), tot / 1e9, torch.cuda.memory_allocated() / 1e9))
big = collections.Counter()
for nb, shp in seen.values():
    big[(nb, shp)] += 1
for (nb, shp), v in sorted(big.items(), key=lambda kv: -kv[0][0] * kv[1])[:12]:
    print(v, "x", nb / 1e6, "MB", shp)
