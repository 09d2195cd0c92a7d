#!/bin/bash
# Runs ON THE GPU BOX: per-kernel scaling of the step with the views per frame (4 frames x {1, 2, 4, 5} views = 4 / 8 / 16 / 20 units).
#   usage: tools/scale_views.sh <tag>    -> gpurun_out/scale_<tag>/v<k>_kernel_stats.csv + v<k>.json (un-traced line of the same shape)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/scale_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in ${VIEWS:-1 2 4 5}; do
  python $REPO/bench.py --views-per-frame $k --no-step8 --no-variants --no-cpu-baseline --no-iters > $OUT/v$k.json 2> $OUT/v$k.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v$k -o b -- python $REPO/bench.py --views-per-frame $k --steps 50 --warmup 5 --no-step8 --no-variants --no-cpu-baseline --no-iters > $OUT/v$k.log 2>&1
  cp $OUT/v$k/b_kernel_stats.csv $OUT/v${k}_kernel_stats.csv 2>/dev/null || find $OUT/v$k -name '*kernel_stats.csv' -exec cp {} $OUT/v${k}_kernel_stats.csv \;
  rm -rf $OUT/v$k
done
tail -c 600 $OUT/v*.json
