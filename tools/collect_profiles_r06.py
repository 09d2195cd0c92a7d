"""Copies the summaries produced by tools/refresh_profiles_r06.sh (gpurun_out/profiles_<tag>/) into profiles/ as r06_*:
   r06_bench.json, r06_bench_kernel_stats_{20v,8v}.csv, r06_pmc_traffic_{20v,8v}.{md,json} (bench.py reads the .json files for
   `roofline.traffic`), and rewrites profiles/README.md.     usage: python tools/collect_profiles_r06.py r06b"""
import collections, csv, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"profiles_{tag}")
dst = os.path.join(root, "profiles")
line = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1]
open(os.path.join(dst, "r06_bench.json"), "w").write(line)
bench = json.loads(line)


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "dm4d::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


readme = ["# profiles/\n\nrocprofv3 summaries of `bench.py` on one MI355X (gfx950), copied from `gpurun_out/` by `tools/collect_profiles_r06.py` after "
          f"`tools/refresh_profiles_r06.sh {tag}` ran on the GPU box.  Round 6: the headline step is BASELINE configs[3]'s per-GPU share, 4 frames x "
          "(4 SDS views + 1 reference view) = 20 (frame, view) units; the 8-unit step of rounds 1-5 (the shipped YAML's iteration) is profiled beside it.  The "
          "kernel names are the same at both shapes, so each shape has its own rocprofv3 runs (`bench.py --views-per-frame 5 | 2 --no-step8 --no-variants`).\n\n"
          f"* `r06_bench.json` -- the JSON line of `python bench.py` (un-profiled, with `step_8_views` and `cpu_baseline`): {bench['value']} {bench['unit']}, "
          f"{bench['ms_per_step']} ms per 20-unit step, whole-view fraction {bench['config']['whole_view_frac_of_hbm_roofline']}, roofline.frac {bench['roofline']['frac']}; "
          f"8-unit step {bench['step_8_views']['value']} views/s, {bench['step_8_views']['ms_per_step']} ms, whole-view fraction {bench['step_8_views']['whole_view_frac_of_hbm_roofline']}\n"]
for vpf, name in ((5, "20v"), (2, "8v")):
    shutil.copy(os.path.join(src, f"stats_v{vpf}", "bench_kernel_stats.csv"), os.path.join(dst, f"r06_bench_kernel_stats_{name}.csv"))
    fetch = per_kernel(os.path.join(src, f"pmc_fetch_v{vpf}", "p_counter_collection.csv"), "FETCH_SIZE")
    write = per_kernel(os.path.join(src, f"pmc_write_v{vpf}", "p_counter_collection.csv"), "WRITE_SIZE")
    out, rows = {}, []
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        out[k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "bytes_raw": (f + w) * 1024, "bytes_corrected": (2 * f + w) * 1024}
        rows.append(f"| `{k}` | {f:.0f} | {w:.0f} | {f/1024:.1f} / {2*f/1024:.1f} | {w/1024:.1f} | {(2*f+w)/1024:.1f} |")
    json.dump(out, open(os.path.join(dst, f"r06_pmc_traffic_{name}.json"), "w"), indent=1)
    open(os.path.join(dst, f"r06_pmc_traffic_{name}.md"), "w").write(
        f"# r06 HBM traffic per launch, {4 * vpf}-view step (rocprofv3 --pmc, separate passes for FETCH_SIZE and WRITE_SIZE)\n\n"
        f"Command: `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python bench.py --views-per-frame {vpf} --no-step8 --no-variants --steps 3 --warmup 1 "
        "--no-cpu-baseline --no-iters` (and the same with `WRITE_SIZE`); `tools/refresh_profiles_r06.sh` + `tools/collect_profiles_r06.py`.\n"
        f"One launch of every kernel covers the {4 * vpf} views of a step (mesh-bound 199,980 Gaussians, 512^2).  `FETCH_SIZE` / `WRITE_SIZE` are in KB.  "
        "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by exactly 2x; the gather-heavy patterns here are uncalibrated, so both "
        "the raw and the x2-corrected read volume are listed; `traffic` in bench.py uses the corrected (upper) figure.\n\n"
        "| kernel | FETCH_SIZE KB | WRITE_SIZE KB | read MB raw / x2 | write MB | total MB (x2 reads) |\n|---|---|---|---|---|---|\n" + "\n".join(rows) + "\n")
    stats = list(csv.DictReader(open(os.path.join(dst, f"r06_bench_kernel_stats_{name}.csv"))))
    calls = max(int(r["Calls"]) for r in stats if "render_bwd" in r["Name"])
    tot = sum(float(r["TotalDurationNs"]) for r in stats if "dm4d::" in r["Name"])
    top = "\n".join(f"| `{r['Name'].replace('void ', '')[:70]}` | {int(r['Calls']) / calls:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['AverageNs']) / 1e3 / (4 * vpf):.2f} | {r['Percentage']} |"
                    for r in stats[:20] if "dm4d::" in r["Name"])
    readme.append(f"\n## {4 * vpf}-view step\n\n* `r06_bench_kernel_stats_{name}.csv` -- `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --views-per-frame {vpf} "
                  f"--no-step8 --no-variants --steps 50 --warmup 10 --no-cpu-baseline --no-iters`\n* `r06_pmc_traffic_{name}.{{md,json}}` -- HBM bytes per launch (bench.py's `roofline.traffic`)\n\n"
                  f"Sum of the `dm4d::` kernel durations per step: {tot / calls / 1e6:.3f} ms (`k_render_fwd_long` and `k_tile_sort<1024>` run BESIDE `k_tile_sort<256>` / `k_render_fwd` on helper "
                  "streams, so the sum exceeds the step time; traced launches are a few per cent longer than un-traced ones).\n\n"
                  f"| kernel | calls/step | avg us | us per view | % |\n|---|---|---|---|---|\n{top}\n")
extra = os.path.join(dst, "README_extra.md")
if os.path.exists(extra):
    readme.append("\n" + open(extra).read())
open(os.path.join(dst, "README.md"), "w").write("".join(readme))
print(line[:400])
