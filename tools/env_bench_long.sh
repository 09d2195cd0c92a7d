#!/bin/bash
# as env_bench.sh with a longer timed region (400 steps ~ 0.45 s: the 50-step default sits inside the device's clock / power
# fluctuation, +-4 % from run to run on one box): tools/env_bench_long.sh rounds "VAR=1" "VAR=2" ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
R=$1; shift
for r in $(seq 1 $R); do
  for e in "$@"; do
    echo "== [$e] round $r: $(env $e python $REPO/bench.py --no-cpu-baseline --no-iters --steps 400 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d["value"], "views/s, bwd", d["roofline"]["avg_launch_us"], "us")')"
  done
done
