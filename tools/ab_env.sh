#!/bin/bash
# A/B of ENVIRONMENT switches of one build on one box by bench.py's own step time:  tools/ab_env.sh rounds "VAR=a" "VAR=b OTHER=c" ...
# (BENCH_ARGS: extra bench.py arguments)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    echo "== [$v] round $r: $(env $v python $REPO/bench.py --no-cpu-baseline --no-iters $BENCH_ARGS 2>/dev/null | tail -1 | python -c '
import sys,json
d=json.loads(sys.stdin.read())
s=d.get("step_8_views") or {}
g=lambda x,*k: (g(x.get(k[0],{}),*k[1:]) if k else x) if isinstance(x,dict) else None
print(d["config"]["views_per_step_per_gpu"],"v:",d["ms_per_step"],"ms/step frac",d["config"]["whole_view_frac_of_hbm_roofline"],"bwd",d["roofline"]["avg_launch_us"],"us depth",g(d,"with_depth_gradient","avg_launch_us"),"full",g(d,"roofline_full","avg_launch_us"),"| 8v:",s.get("ms_per_step"),"ms/step frac",s.get("whole_view_frac_of_hbm_roofline"),"bwd",g(s,"roofline","avg_launch_us"),"depth",g(s,"with_depth_gradient","avg_launch_us"),"full",g(s,"roofline_full","avg_launch_us"))')"
  done
done
