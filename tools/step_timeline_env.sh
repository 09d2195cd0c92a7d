#!/bin/bash
# tools/step_timeline.sh for ONE build under several environment settings: tools/step_timeline_env.sh "VAR=a" "VAR=b" ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  rm -rf /tmp/stl
  env $e rocprofv3 --kernel-trace --output-format csv -d /tmp/stl -o k -- python $REPO/bench.py --no-cpu-baseline --no-iters --steps 30 --warmup 5 > /dev/null 2>&1
  echo "== $e"
  python - <<PY
import csv, glob
f = glob.glob('/tmp/stl/**/k_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'k_nodenet_fwd' in r['Kernel_Name']]
head = [k for k in range(len(starts) - 1) if any('k_render_bwd<6, 2>' in r['Kernel_Name'] for r in rows[starts[k]:starts[k + 1]])]
k = head[len(head) // 2]
i0, i1 = starts[k], starts[k + 1]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1]:
    n = r['Kernel_Name'].replace('void ', '').split('(')[0]
    if 'dm4d' not in n: continue
    print(f"  {(int(r['Start_Timestamp'])-t0)/1e3:8.1f} -> {(int(r['End_Timestamp'])-t0)/1e3:8.1f} us  q{r.get('Queue_Id','?')}  {n[:80]}")
print(f"  step: {(int(rows[i1]['Start_Timestamp'])-t0)/1e3:.1f} us")
PY
done
