#!/bin/bash
# How much of a dynamic-stage iteration is GPU time: sum of all kernel durations per iteration (rocprofv3 --kernel-trace --stats of
# tools/iters_per_sec.py: 3 warm-up + 10 timed iterations + graph captures) beside the wall time the script reports.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/igb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/igb -o k -- python $REPO/tools/iters_per_sec.py 2>/dev/null | tail -1
python - <<PY
import csv, glob
f = glob.glob('/tmp/igb/**/k_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"sum of kernel time: {tot/1e6:.1f} ms over the run (13 iterations + capture warm-ups)")
z = sum(float(r['TotalDurationNs']) for r in rows if 'dm4d::k_' not in r['Name'] or 'groupnorm' in r['Name'] or 'k_add_bias' in r['Name'] or 'k_geglu' in r['Name'])
print(f"  torch / library / Zero123 operator kernels: {z/1e6:.1f} ms; dm4d render / network kernels: {(tot-z)/1e6:.1f} ms")
for r in rows[:12]:
    print(f"  {float(r['TotalDurationNs'])/1e6:8.1f} ms {int(r['Calls']):6d}  {r['Name'][:100]}")
PY
