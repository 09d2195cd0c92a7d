"""configs[4] sanity: ~1 M mesh-bound Gaussians, 1024^2, DQS skinning, 2 views -- runs, no overflow, finite grads."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from dreammesh4d_amd import geometry as geo, ops, synthetic as syn, views
dev = torch.device('cuda:0')
F_, M, H, W, B = 166_667, 1000, 1024, 1024, 2
t0 = time.time()
sc = syn.mesh_bound_scene(F_, n_nodes=M, k=4, seed=0)
print("scene built in %.1f s: faces %d verts %d" % (time.time() - t0, len(sc["faces"]), len(sc["verts"])), flush=True)
T = lambda a: torch.tensor(a, device=dev)
graph = ops.DeformGraph(sc["verts"], sc["nbr_idx"], sc["nbr_w"], M, dev)
topo = ops.MeshTopology(sc["faces"], len(sc["verts"]), 6, dev)
verts, faces = T(sc["verts"]), T(sc["faces"])
qs = geo.quaternions(verts, faces, T(sc["complex"]), 6)
scales = geo.scaling(T(sc["log_scales"]), syn.THICKNESS); opac = geo.strengths(T(sc["densities"])); rgb = geo.points_rgb(T(sc["sh_dc"]))
ts, motion = syn.node_motion(M, B, seed=0)
raw = {k: torch.stack([T(m[k]) for m in motion]).requires_grad_(True) for k in ("trans", "d_rot", "strain", "d_opacity")}
cams = [syn.make_camera(H, W, elev_deg=10 + 20 * b, azim_deg=40 * b) for b in range(B)]
vm = torch.stack([T(c.viewmatrix) for c in cams]); pm = torch.stack([T(c.projmatrix) for c in cams])
r = views.ViewRenderer(graph, topo, H, W, cams[0].tanfov, method="dqs")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = views.render_views(r, raw["trans"], raw["d_rot"], raw["strain"], raw["d_opacity"].squeeze(-1), qs, scales, opac, rgb, vm, pm, torch.ones(6, device=dev))
    (out["color"].sum() + out["alpha"].sum()).backward()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("iter", it, "%.2f ms" % (1e3 * dt), "D", r.check(), "R", r.last_num_records, flush=True)
print("finite grads:", all(torch.isfinite(v.grad).all().item() for k, v in raw.items() if v.grad is not None), "mem GB", torch.cuda.max_memory_allocated() / 1e9)
print("alpha coverage", float((out["alpha"] > 0.5).float().mean()))
