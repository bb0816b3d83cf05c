#!/bin/bash
# rocprofv3 kernel trace of the training-step bench (run on the GPU box through gpurun) -> gpurun_out/train_kernel_trace.txt
# usage: profile_train.sh "<label>" [H W batch depth]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LABEL="$1"; shift
ARGS="${@:-512 512 3 101}"
rm -rf gpurun_out/prof_train
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o p -- python3 tests/tools/bench_train.py $ARGS > gpurun_out/prof_train.log 2>&1
DB=$(find gpurun_out/prof_train -name "*.db" | head -1)
python3 tests/tools/rocprof_summary.py $DB "rocprofv3 --kernel-trace --stats -- python3 tests/tools/bench_train.py $ARGS   ($LABEL)" > gpurun_out/train_kernel_trace.txt
tail -1 gpurun_out/prof_train.log | cut -c1-300
rm -rf gpurun_out/prof_train
