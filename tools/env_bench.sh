#!/bin/bash
# bench.py under several environment settings on the SAME GPU box: tools/env_bench.sh rounds "VAR=1 VAR2=x" "VAR=2" ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
R=$1; shift
for r in $(seq 1 $R); do
  for e in "$@"; do
    echo "== [$e] round $r: $(env $e python $REPO/bench.py --no-cpu-baseline --no-iters 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d["value"], "views/s, bwd", d["roofline"]["avg_launch_us"], "us")')"
  done
done
