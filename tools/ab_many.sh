#!/bin/bash
# A/B/C... of several builds of libdm4d_hip.so by bench.py's own step time on ONE box: tools/ab_many.sh rounds a.so b.so c.so ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
R=$1; shift
cp $REPO/dreammesh4d_amd/libdm4d_hip.so /tmp/libdm4d_keep.so
for r in $(seq 1 $R); do
  for v in "$@"; do
    cp $REPO/$v $REPO/dreammesh4d_amd/libdm4d_hip.so
    echo "== $v round $r: $(python $REPO/bench.py --no-cpu-baseline --no-iters 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step, bwd", d["roofline"]["avg_launch_us"], "us; depth", d["with_depth_gradient"]["avg_launch_us"], "full", d["roofline_full"]["avg_launch_us"])')"
  done
done
cp /tmp/libdm4d_keep.so $REPO/dreammesh4d_amd/libdm4d_hip.so
