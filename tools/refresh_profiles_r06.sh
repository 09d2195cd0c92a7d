#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): regenerates what profiles/ holds for round 6 -- the bench line, kernel stats and PMC traffic of the
# 20-view step (the headline) and of the 8-view step, each as its own run (the kernel names are the same at both shapes).
#   usage: tools/refresh_profiles_r06.sh r06   -> gpurun_out/profiles_<tag>/
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err
for V in 5 2; do
  A="--views-per-frame $V --no-step8 --no-variants --no-cpu-baseline --no-iters"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_v$V -o bench -- python $REPO/bench.py $A --steps 50 --warmup 10 > $OUT/stats_v$V.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_v$V -o p -- python $REPO/bench.py $A --steps 3 --warmup 1 > $OUT/pmc_fetch_v$V.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_v$V -o p -- python $REPO/bench.py $A --steps 3 --warmup 1 > $OUT/pmc_write_v$V.log 2>&1
done
find $OUT -name '*.csv' | head -20
