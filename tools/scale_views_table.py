"""profiles/r06_view_scaling.md from what tools/scale_views.sh wrote (gpurun_out/scale_<tag>/): the step's kernels at 4 frames x {1, 2, 4, 5}
views = 4 / 8 / 16 / 20 (frame, view) units -- which kernels amortise over a frame's views and which are paid per view.   usage: python tools/scale_views_table.py r06b"""
import csv, json, os, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"scale_{tag}")
ks = [k for k in (1, 2, 4, 5) if os.path.exists(os.path.join(src, f"v{k}_kernel_stats.csv"))]
tabs, lines = {}, {}
for k in ks:
    rows = list(csv.DictReader(open(os.path.join(src, f"v{k}_kernel_stats.csv"))))
    calls = max(int(r["Calls"]) for r in rows if "render_bwd" in r["Name"])
    tabs[k] = {r["Name"].replace("dm4d::", "").split("(")[0].replace("void ", ""): float(r["AverageNs"]) / 1e3 * int(r["Calls"]) / calls for r in rows if "dm4d::" in r["Name"]}
    l = [x for x in open(os.path.join(src, f"v{k}.json")) if x.startswith("{")]
    lines[k] = json.loads(l[-1]) if l else None
names = sorted(tabs[ks[-1]], key=lambda n: -tabs[ks[-1]][n])
out = [f"# r06 the step's kernels against the views per frame (4 frames x 1 / 2 / 4 / 5 views; `tools/scale_views.sh {tag}` on one box)\n\n",
       "Per step, microseconds under `rocprofv3 --kernel-trace --stats` (50 steps; a traced launch is a few per cent longer than an un-traced one), and what a kernel costs per "
       "(frame, view) unit at 20 and at 8 units.  `k_tile_sort<1024>` and `k_render_fwd_long` run on helper streams beside `k_tile_sort<256>` / `k_render_fwd`.\n\n",
       "| kernel | " + " | ".join(f"{4 * k} units" for k in ks) + " | us / unit @ 20 | us / unit @ 8 |\n|---|" + "---|" * (len(ks) + 2) + "\n"]
for n in names:
    if tabs[ks[-1]][n] < 0.05:
        continue
    out.append(f"| `{n}` | " + " | ".join(f"{tabs[k].get(n, 0):.1f}" for k in ks) + f" | {tabs[5][n] / 20 if 5 in tabs else 0:.2f} | {tabs[2].get(n, 0) / 8 if 2 in tabs else 0:.2f} |\n")
out.append("| sum of the kernels | " + " | ".join(f"{sum(tabs[k].values()):.0f}" for k in ks) + " | | |\n")
out.append("\nThe same shapes un-traced (`python bench.py --views-per-frame k --no-step8 --no-variants`):\n\n| units per step | ms per step | views / s | whole-view fraction of the HBM roofline | blend backward us (HIP events) | its fraction |\n|---|---|---|---|---|---|\n")
for k in ks:
    d = lines[k]
    if d:
        out.append(f"| {4 * k} | {d['ms_per_step']} | {d['value']} | {d['config']['whole_view_frac_of_hbm_roofline']} | {d['roofline']['avg_launch_us']} | {d['roofline']['frac']} |\n")
open(os.path.join(root, "profiles", "r06_view_scaling.md"), "w").write("".join(out))
print("".join(out))
