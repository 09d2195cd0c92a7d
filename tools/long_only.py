"""Bench steps with the regular blend kernels' waves disabled (dm4d_debug_trace min_work = huge): what the long-cell
kernels cost when they run alone.  Run under rocprofv3 --kernel-trace --stats."""
import sys, torch
sys.path.insert(0, '/root/repo')
import bench
from dreammesh4d_amd import _lib
dev = torch.device('cuda:0')
wl = bench.Workload(dev, 0, 1, views_per_frame=2)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
L = _lib.lib()
buf = torch.zeros(8 * 4096 * 8, 4, dtype=torch.int64, device=dev)
_lib.check(L.dm4d_debug_trace(buf.data_ptr(), int(sys.argv[1]) if len(sys.argv) > 1 else 1000000))
for _ in range(10):
    wl.step()
torch.cuda.synchronize()
_lib.check(L.dm4d_debug_trace(None, 0))
