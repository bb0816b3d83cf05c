#!/bin/bash
# rocprofv3 kernel trace of the parity mode (bf16x3) bench command -> gpurun_out/bench_x3_kernel_trace.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_bench_x3
CMD="python3 bench.py --precision bf16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench_x3 -o p -- $CMD > gpurun_out/prof_bench_x3.log 2>&1
DB=$(find gpurun_out/prof_bench_x3 -name "*.db" | head -1)
python3 tests/tools/rocprof_summary.py $DB "rocprofv3 --kernel-trace --stats -- $CMD   ($1)" > gpurun_out/bench_x3_kernel_trace.txt
tail -1 gpurun_out/prof_bench_x3.log | cut -c1-400
head -14 gpurun_out/bench_x3_kernel_trace.txt
rm -rf gpurun_out/prof_bench_x3
