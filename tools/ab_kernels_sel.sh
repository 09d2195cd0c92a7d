#!/bin/bash
# tools/ab_kernels.sh restricted to kernels whose name matches a pattern: PATTERN='nodenet|mlp_bwd' tools/ab_kernels_sel.sh a.so b.so ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cp $REPO/dreammesh4d_amd/libdm4d_hip.so /tmp/libdm4d_keep.so
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  cp $REPO/$v $REPO/dreammesh4d_amd/libdm4d_hip.so
  rm -rf /tmp/abk
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -o k -- python $REPO/bench.py --no-cpu-baseline --no-iters --steps ${STEPS:-60} --warmup 5 > /dev/null 2>&1
  echo "== $v"
  PATTERN="${PATTERN:-.}" python - <<PY
import csv, glob, os, re
f = glob.glob('/tmp/abk/**/k_kernel_stats.csv', recursive=True)[0]
pat = re.compile(os.environ["PATTERN"])
for r in list(csv.DictReader(open(f))):
    if pat.search(r['Name']):
        print(f"  {float(r['AverageNs'])/1e3:8.1f} us x {r['Calls']:>5s}  {r['Name'][:70]}")
PY
done
cp /tmp/libdm4d_keep.so $REPO/dreammesh4d_amd/libdm4d_hip.so
