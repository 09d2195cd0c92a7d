#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the 20-view step's kernels -> gpurun_out/sq_r06/bench_sq.txt (tools/pmc_sq.py reads gpurun_out/pmc_sq_{a,b,c})
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
A="--steps 3 --warmup 1 --no-cpu-baseline --no-iters --no-step8 --no-variants"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_sq_a -o p -- python $R/bench.py $A > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc_sq_b -o p -- python $R/bench.py $A > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_sq_c -o p -- python $R/bench.py $A > /dev/null 2>&1
mkdir -p $REPO/gpurun_out/sq_r06
cd $REPO
python tools/pmc_sq.py 'k_render_fwd<6>' 'k_render_fwd_long<6>' 'k_scatter' 'k_render_bwd<6, 2>' 'k_tile_sort<256>' 'k_tile_sort<1024>' 'k_nodenet_bwdB' 'k_nodenet_fwd' 'k_gather_face_bwd<2>' 'k_preprocess' 'k_face_bwd_vertex' > gpurun_out/sq_r06/bench_sq.txt 2>&1
rm -rf gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b gpurun_out/pmc_sq_c
tail -30 gpurun_out/sq_r06/bench_sq.txt
