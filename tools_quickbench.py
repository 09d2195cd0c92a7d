import math, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from dreammesh4d_amd import synthetic as syn
import dreammesh4d_amd.diff_gaussian_rasterization as dgr
dev = torch.device("cuda:0")
n, H, W = 200_000, 512, 512
sc = syn.random_splat_scene(n, seed=0)
cam = syn.make_camera(H, W)
T = lambda a, rg=False: torch.tensor(a, device=dev).requires_grad_(rg)
m3, op = T(sc["means3D"], True), T(sc["opacities"][:, None], True)
col, scl, rot = T(sc["colors"], True), T(sc["scales"], True), T(sc["rotations"], True)
rs = dgr.GaussianRasterizationSettings(H, W, cam.tanfov, cam.tanfov, T(np.ones(3, np.float32)), 1.0, T(cam.viewmatrix), T(cam.projmatrix), 0, T(cam.campos), False, False)
rast = dgr.GaussianRasterizer(rs)
gC = torch.randn(3, H, W, device=dev); gA = torch.randn(1, H, W, device=dev)
def step():
    m2 = torch.zeros_like(m3, requires_grad=True)
    color, radii, depth, alpha = rast(means3D=m3, means2D=m2, opacities=op, colors_precomp=col, scales=scl, rotations=rot)
    torch.autograd.backward([color, alpha], [gC, gA])
    return radii
for _ in range(5): r = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for _ in range(K): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"fwd+bwd one pass: {dt*1e3:.3f} ms  visible={(r>0).sum().item()}")
