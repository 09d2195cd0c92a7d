#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the round-3 kernels -> gpurun_out/sq_r03/ (bench kernels: pmc_sq.sh / pmc_sq.py; the two
# convolution kernels on one representative shape each: conv_pmc.sh)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/sq_r03
bash tools/pmc_sq.sh > /dev/null 2>&1
python tools/pmc_sq.py 'k_render_fwd<6>' 'k_render_bwd<6, 2>' 'k_render_bwd<6, 1>' 'k_tile_sort<256>' 'k_gather_bwd<2>' 'k_preprocess' 'k_nodenet_fwd' 'k_nodenet_bwd3' > gpurun_out/sq_r03/bench_sq.txt 2>&1
rm -rf gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b gpurun_out/pmc_sq_c
echo "== k_conv3x3<2,2,2,2,4> (cfg 3), UNet 8x16x16 640->640" > gpurun_out/sq_r03/conv_sq.txt
bash tools/conv_pmc.sh 3 8 16 640 640 >> gpurun_out/sq_r03/conv_sq.txt 2>&1
echo "== k_conv3x3_direct<2,9,4> (cfg 7), VAE 4x64x64 512->512" >> gpurun_out/sq_r03/conv_sq.txt
bash tools/conv_pmc.sh 7 4 64 512 512 >> gpurun_out/sq_r03/conv_sq.txt 2>&1
cat gpurun_out/sq_r03/conv_sq.txt | tail -40
