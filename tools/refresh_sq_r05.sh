#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the round-4 bench kernels -> gpurun_out/sq_r05/bench_sq.txt (tools/pmc_sq.sh / pmc_sq.py)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/sq_r05
bash tools/pmc_sq.sh > /dev/null 2>&1
python tools/pmc_sq.py 'k_render_fwd<6>' 'k_render_fwd_long<6>' 'k_scatter' 'k_render_bwd<6, 2>' 'k_tile_sort<256>' 'k_tile_sort<1024>' 'k_nodenet_bwdB' 'k_nodenet_fwd' 'k_gather_face_bwd<2>' 'k_preprocess' 'k_face_bwd_vertex' > gpurun_out/sq_r05/bench_sq.txt 2>&1
rm -rf gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b gpurun_out/pmc_sq_c
tail -60 gpurun_out/sq_r05/bench_sq.txt
