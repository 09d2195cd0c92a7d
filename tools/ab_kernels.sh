#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats of bench.py) for several builds of libdm4d_hip.so on the SAME box:
#   tools/ab_kernels.sh a.so b.so ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cp $REPO/dreammesh4d_amd/libdm4d_hip.so /tmp/libdm4d_keep.so
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  cp $REPO/$v $REPO/dreammesh4d_amd/libdm4d_hip.so
  rm -rf /tmp/abk
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -o k -- python $REPO/bench.py --no-cpu-baseline --no-iters --no-step8 --no-variants --steps 30 --warmup 5 $BENCH_ARGS > /dev/null 2>&1
  echo "== $v"
  python - <<PY
import csv, glob
f = glob.glob('/tmp/abk/**/k_kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"  {float(r['AverageNs'])/1e3:8.1f} us x {r['Calls']:>5s}  {r['Name'][:70]}")
PY
done
cp /tmp/libdm4d_keep.so $REPO/dreammesh4d_amd/libdm4d_hip.so
