#!/bin/bash
# Runs ON THE GPU BOX: everything profiles/ holds for round 2 (bench + rocprofv3 stats + PMC traffic via refresh_profiles.sh,
# SQ counters of the blend / sort / gather kernels, the VALU issue-cost micro-benchmark).  Outputs under gpurun_out/.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash tools/refresh_profiles.sh r02 > /dev/null 2>&1
rm -rf gpurun_out/profiles_r02/stats/*/*.db gpurun_out/profiles_r02/pmc_*/*/*.db 2>/dev/null
bash tools/pmc_sq.sh > /dev/null 2>&1
python tools/pmc_sq.py 'k_render_fwd<6>' 'k_render_fwd_long<6>' 'k_render_bwd<6, true>' 'k_render_bwd<6, false>' 'k_tile_sort<256>' 'k_tile_sort<1024>' 'k_gather_bwd<3>' 'k_preprocess' > gpurun_out/profiles_r02/pmc_sq_summary.txt 2>&1
rm -rf gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b gpurun_out/pmc_sq_c
./tools/ubench/valu > gpurun_out/profiles_r02/ubench_valu.txt 2>&1
./tools/ubench/stream > gpurun_out/profiles_r02/ubench_stream.txt 2>&1
du -sh gpurun_out; tail -c 1500 gpurun_out/profiles_r02/bench.json
