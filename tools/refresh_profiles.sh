#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): regenerates everything profiles/ holds for the current round.
#   usage: tools/refresh_profiles.sh r01
# Outputs under gpurun_out/profiles_<tag>/ ; copy them into profiles/ afterwards (tools/collect_profiles.py).
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $REPO/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-iters > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-iters > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-iters > $OUT/pmc_write.log 2>&1
ls -R $OUT | head -30
