#!/bin/bash
# Every profile of a round in one call (run on the GPU box through gpurun): writes gpurun_out/<prefix>_* -- copy them under profiles/.
#   kernel trace + stats of the bench command, HBM traffic (FETCH / WRITE PMC passes), SQ / TCC counters and the phase clock of the
#   tower kernel on its PRODUCTION launches (512 frames per step), the training step's kernel trace, the per-op roofline tables
#   (tests/tools/op_table.py) of the headline and the parity-mode forward.
# usage: profile_round.sh <prefix, e.g. round4> "<label>"
P=${1:-round}; LABEL="${2:-}"
cd $GRAFT_REPO_ROOT
tests/tools/profile_bench.sh "$LABEL" > gpurun_out/${P}_profile_bench.log 2>&1
cp gpurun_out/bench_kernel_trace.txt gpurun_out/${P}_bench_kernel_trace.txt; cp gpurun_out/bench_kernel_stats.csv gpurun_out/${P}_bench_kernel_stats.csv
tests/tools/pmc_traffic.sh > gpurun_out/${P}_pmc_traffic.log 2>&1
cp gpurun_out/head_conv_pmc.json gpurun_out/${P}_head_conv_pmc.json
PMC_PRODUCTION=1 tests/tools/pmc_head_conv.sh > gpurun_out/${P}_pmc_counters.log 2>&1
python3 tests/tools/pmc_head_summary.py gpurun_out "bench.py --steps 2: the production launches, 512 frames per step, tower layers 1-3" > gpurun_out/${P}_head_conv_counters.json
rm -rf gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_tcc
( echo "# phase clock of the tower kernel (variant 90: s_memtime stamps, wave 0) on the PRODUCTION shapes: 512 frames, N = 10, 512x512 -- $LABEL";
  B=512 python3 tests/tools/bench_head_conv.py 0:1 90:1 0:2 90:2 0:3 90:3 0:4 90:4 0:0 2>&1 | grep "phase clock\|round 1" ) > gpurun_out/${P}_phase_clock.txt
tests/tools/profile_train.sh "$LABEL" 512 512 3 101 > gpurun_out/${P}_profile_train.log 2>&1
cp gpurun_out/train_kernel_trace.txt gpurun_out/${P}_train_step_kernel_trace.txt
tests/tools/op_table.sh "$LABEL" > gpurun_out/${P}_op_table.log 2>&1
cp gpurun_out/op_table.txt gpurun_out/${P}_op_table.txt
tests/tools/op_table.sh "$LABEL; parity mode f16mx" --precision f16mx --batch 256 > gpurun_out/${P}_op_table_mx.log 2>&1
cp gpurun_out/op_table.txt gpurun_out/${P}_op_table_f16mx.txt
tests/tools/op_table.sh "$LABEL; f16mx4 (e2m1 cross terms, opt-in)" --precision f16mx4 --batch 256 > gpurun_out/${P}_op_table_mx4.log 2>&1
cp gpurun_out/op_table.txt gpurun_out/${P}_op_table_f16mx4.txt
tests/tools/op_table.sh "$LABEL; parity mode of rounds 2-4" --precision bf16x3 --batch 256 > gpurun_out/${P}_op_table_x3.log 2>&1
cp gpurun_out/op_table.txt gpurun_out/${P}_op_table_bf16x3.txt
# the f16mx tower kernel: phase clock (with / without the loop's LDS-DMA) and SQ counters on its production launches (256 frames per step);
# the stamped variants exist for the hx instantiation only, hence BOD_MX_CLS_H4=0 (all three heads on hx rows: the three-head launch)
( echo "# f16mx tower kernel (conv_igemm_mx_kernel<1>), 256 frames, N = 10, 512x512: phase clock, variant 90; variant 91 = the same without the loop's LDS-DMA -- $LABEL";
  BOD_MX_CLS_H4=0 PRECISION=f16mx B=256 python3 tests/tools/bench_head_conv.py 0:1 90:1 91:1 0:3 90:3 0:0 2>&1 | grep "phase clock\|round 1\|Error";
  echo "# f16mx4 tower kernel (conv_igemm_mx_kernel<3>), same shapes: launch times per layer";
  PRECISION=f16mx4 B=256 python3 tests/tools/bench_head_conv.py 0:0 0:1 0:2 0:3 2>&1 | grep "round 1" ) > gpurun_out/${P}_f16mx_phase_clock.txt
( echo "# SQ counters of conv_igemm_mx_kernel over bench.py --precision f16mx --batch 256 --steps 2 (per-launch means over both instantiations) -- $LABEL";
  tests/tools/pmc_kernel.sh conv_igemm_mx_kernel --precision f16mx --batch 256 2>&1 | tail -20 ) > gpurun_out/${P}_f16mx_counters.txt
tests/tools/pmc_posterior.sh 512 > gpurun_out/${P}_pmc_posterior.log 2>&1
cp gpurun_out/posterior_pmc.json gpurun_out/${P}_posterior_pmc.json
ls -la gpurun_out/${P}_*
cat gpurun_out/${P}_head_conv_counters.json gpurun_out/${P}_phase_clock.txt
