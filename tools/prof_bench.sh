#!/bin/bash
# usage: tools/prof_bench.sh <tag>  -- rocprofv3 kernel stats of bench.py into gpurun_out/prof_<tag>/
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_$1 -o bench -- python /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-iters > /root/repo/gpurun_out/prof_$1.log 2>&1
