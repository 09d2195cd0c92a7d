"""One convolution shape through csrc/conv_mfma.hip, repeated (for rocprofv3 counter passes): conv_one.py N H Cin Cout [reps]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dreammesh4d_amd import conv_mfma
N, H, Ci, Co = (int(a) for a in sys.argv[1:5]); reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda:0")
x = torch.randn(N, Ci, H, H, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(Co, Ci, 3, 3, device=dev, dtype=torch.float16) * 0.02).contiguous(memory_format=torch.channels_last)
pw = conv_mfma.pack_weight(w)
for _ in range(reps): y = conv_mfma.conv3x3(x, pw, None)
torch.cuda.synchronize()
