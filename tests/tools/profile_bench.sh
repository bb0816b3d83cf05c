#!/bin/bash
# rocprofv3 kernel trace of the bench command (run on the GPU box through gpurun) -> gpurun_out/bench_kernel_trace.txt (+ rocprofv3's
# own kernel_stats.csv); copy both under profiles/ with the round's prefix.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_bench
CMD="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o p -- $CMD > gpurun_out/prof_bench.log 2>&1
DB=$(find gpurun_out/prof_bench -name "*.db" | head -1)
python3 tests/tools/rocprof_summary.py $DB "rocprofv3 --kernel-trace --stats -- $CMD   ($1)" > gpurun_out/bench_kernel_trace.txt
rm -rf gpurun_out/prof_bench_csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench_csv -o p -- $CMD > gpurun_out/prof_bench_csv.log 2>&1
cp $(find gpurun_out/prof_bench_csv -name "*kernel_stats.csv" | head -1) gpurun_out/bench_kernel_stats.csv
tail -1 gpurun_out/prof_bench.log | cut -c1-600
head -12 gpurun_out/bench_kernel_trace.txt
rm -rf gpurun_out/prof_bench gpurun_out/prof_bench_csv
