#!/bin/bash
# SQ counter passes over bench.py (3 steps) -> gpurun_out/pmc_sq_{a,b,c}/ ; summarise with tools/pmc_sq.py
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_sq_a -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-iters > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc_sq_b -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-iters > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_sq_c -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-iters > /dev/null 2>&1
ls $R/gpurun_out/pmc_sq_a $R/gpurun_out/pmc_sq_b $R/gpurun_out/pmc_sq_c
