#!/bin/bash
# PMC passes for the head-tower conv micro-benchmark (run on the GPU box through gpurun).
# usage: pmc_head_conv.sh <variant list> ; writes gpurun_out/pmc_<set>/
#        PMC_PRODUCTION=1 pmc_head_conv.sh : the same three passes over two steps of the headline bench (512 frames per step: the
#        production launches of the tower kernel) instead of the micro-benchmark
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V="${@:-0}"
rm -rf gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_tcc
if [ "$PMC_PRODUCTION" = "1" ]; then
  CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_sq1 -o p -- $CMD > gpurun_out/pmc_sq1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_DATA_FIFO_FULL --output-format csv -d gpurun_out/pmc_sq2 -o p -- $CMD > gpurun_out/pmc_sq2.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_TA_BUSY TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_tcc -o p -- $CMD > gpurun_out/pmc_tcc.log 2>&1
  ls gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_tcc
  exit 0
fi
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_sq1 -o p -- python3 tests/tools/bench_head_conv.py $V > gpurun_out/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_DATA_FIFO_FULL --output-format csv -d gpurun_out/pmc_sq2 -o p -- python3 tests/tools/bench_head_conv.py $V > gpurun_out/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_TA_BUSY TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_tcc -o p -- python3 tests/tools/bench_head_conv.py $V > gpurun_out/pmc_tcc.log 2>&1
ls gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_tcc
