#!/bin/bash
# Same-box A/B of two TREES (each with its own bench.py and built library), alternating:  tools/ab_rounds.sh <rounds> <treeA> <treeB>
# e.g. tools/ab_rounds.sh 3 build_ab/r03tree .   (build_ab/r03tree: `git worktree add` of the previous round's head, built in place)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
R=$1; shift
for r in $(seq 1 $R); do
  for t in "$@"; do
    echo "== [$t] round $r: $(python $REPO/$t/bench.py --no-cpu-baseline --no-iters --steps 400 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d["value"], "views/s, bwd", d["roofline"]["avg_launch_us"], "us; with depth", d["with_depth_gradient"]["ms_per_step"], "full", d["roofline_full"]["ms_per_step"])')"
  done
done
