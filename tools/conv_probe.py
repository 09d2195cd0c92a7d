"""One convolution shape per line through csrc/conv_mfma.hip under the tile configurations of DM4D_CONV_CFG (GPU time from a
hipGraph of 20 calls).  With a library built with -DDM4D_CONV_PROBE the probe switches of DM4D_CONV_PROBE (1 no stores, 2 no MFMA,
4 no DMA, 8 return at once, 16 one k-tile, 32 no barrier, 64 no LDS reads) split the kernel time (profiles/r03_zero123.md)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dreammesh4d_amd import conv_mfma
dev = torch.device("cuda:0")
SH = [(4,256,128,128),(4,128,256,256),(4,64,512,512),(8,32,640,640),(8,16,640,640)]
for (N,H,Ci,Co) in SH:
    x = torch.randn(N, Ci, H, H, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, Ci, 3, 3, device=dev, dtype=torch.float16) * 0.02)
    pw = conv_mfma.pack_weight(w)
    fl = 2.0*N*H*H*Ci*Co*9
    for cfg in [int(c) for c in os.environ.get('CFGS', '3,7,9').split(',')]:
        os.environ["DM4D_CONV_CFG"] = str(cfg)
        row = []
        for probe in [int(c) for c in os.environ.get('PROBES', '0').split(',')]:
            os.environ["DM4D_CONV_PROBE"] = str(probe)
            for _ in range(3): conv_mfma.conv3x3(x, pw)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20): conv_mfma.conv3x3(x, pw)
            g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): g.replay()
            torch.cuda.synchronize(); t = (time.perf_counter()-t0)/100
            row.append(f"p{probe}:{t*1e6:7.1f}")
        print(f"{N}x{H}^2 {Ci}->{Co} cfg{cfg} ", " ".join(row), f"  ({fl/1e9:.1f} GFLOP; p0 = {fl/float(row[0][3:])/1e6:.0f} TF/s)")
