#!/bin/bash
# Start / end of every kernel of ONE steady-state bench step relative to the step's first kernel (rocprofv3 --kernel-trace),
# for several builds of libdm4d_hip.so on the same box:   tools/step_timeline.sh a.so b.so ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cp $REPO/dreammesh4d_amd/libdm4d_hip.so /tmp/libdm4d_keep.so; cp $REPO/dreammesh4d_amd/libdm4d_hip.so $REPO/build_cur.so 2>/dev/null
cd /tmp && export TMPDIR=/tmp
ALL=${ALL:-0}
for v in "$@"; do
  cp $REPO/$v $REPO/dreammesh4d_amd/libdm4d_hip.so
  rm -rf /tmp/stl
  rocprofv3 --kernel-trace --output-format csv -d /tmp/stl -o k -- python $REPO/bench.py --no-cpu-baseline --no-iters --no-step8 --no-variants --steps 30 --warmup 5 $BENCH_ARGS > /dev/null 2>&1
  echo "== $v"
  python - <<PY
import csv, glob
f = glob.glob('/tmp/stl/**/k_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# steps start at k_nodenet_fwd; take the 20th from the end
starts = [i for i, r in enumerate(rows) if 'k_nodenet_fwd' in r['Kernel_Name']]
# a HEADLINE step (32-byte records: k_render_bwd<6, 2>), the middle one (bench.py ends with the with-depth and full-backward legs)
head = [k for k in range(len(starts) - 1) if any('k_render_bwd<6, 2>' in r['Kernel_Name'] for r in rows[starts[k]:starts[k + 1]])]
k = head[len(head) // 2]
i0, i1 = starts[k], starts[k + 1]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1]:
    n = r['Kernel_Name'].replace('void ', '').split('(')[0]
    if 'dm4d' not in n and '$ALL' != '1': continue
    print(f"  {(int(r['Start_Timestamp'])-t0)/1e3:8.1f} -> {(int(r['End_Timestamp'])-t0)/1e3:8.1f} us  {n[:90]}")
print(f"  step: {(int(rows[i1]['Start_Timestamp'])-t0)/1e3:.1f} us")
PY
done
cp /tmp/libdm4d_keep.so $REPO/dreammesh4d_amd/libdm4d_hip.so
