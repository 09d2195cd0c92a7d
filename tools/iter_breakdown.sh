#!/bin/bash
# Kernels of ONE steady-state dynamic-stage iteration (tools/iters_per_sec.py under rocprofv3 --kernel-trace): the segment between
# two consecutive k_preprocess launches (one per iteration), aggregated by kernel, with the GPU-busy sum and the wall span.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/itb
rocprofv3 --kernel-trace --output-format csv -d /tmp/itb -o k -- python $REPO/tools/iters_per_sec.py 2>/dev/null | tail -1
python - <<PY
import csv, glob, collections
f = glob.glob('/tmp/itb/**/k_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'k_preprocess' in r['Kernel_Name']]
i0, i1 = marks[-3], marks[-2]
seg = rows[i0:i1]
span = (int(rows[i1]['Start_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1e6
acc = collections.defaultdict(lambda: [0.0, 0])
for r in seg:
    n = r['Kernel_Name'].replace('void ', '')[:260]
    acc[n][0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    acc[n][1] += 1
busy = sum(v[0] for v in acc.values()) / 1e3
print(f"iteration: wall span {span:.2f} ms, sum of kernel time {busy:.2f} ms, {len(seg)} launches (graph replays are traced kernel by kernel)")
def grp(n):
    if n.startswith('dm4d::k_conv') or 'Cijk' in n or 'igemm' in n or 'attn' in n or 'groupnorm' in n or 'layer_norm' in n or 'geglu' in n or 'k_add' in n or 'grouped_conv' in n or 'bwd_kernel' in n: return 'zero123'
    if n.startswith('dm4d::'): return 'render/network (dm4d)'
    return 'torch elementwise / optimizer / other'
g = collections.defaultdict(float)
for n, v in acc.items(): g[grp(n)] += v[0]
for k, v in sorted(g.items(), key=lambda x: -x[1]): print(f"  {v/1e3:7.2f} ms  {k}")
print("top kernels outside the zero123 group:")
for n, v in sorted(((n, v) for n, v in acc.items() if grp(n) != 'zero123'), key=lambda x: -x[1][0])[:28]:
    print(f"  {v[0]:8.1f} us {v[1]:4d}  {n}")
PY
