#!/bin/bash
# Per-op roofline table of the headline forward (run on the GPU box through gpurun) -> gpurun_out/op_table.txt
#   usage: op_table.sh "<label>" [extra bench.py flags]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LABEL="$1"; shift
rm -rf gpurun_out/prof_ops
export BOD_DUMP_OPS=1
CMD="python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary $*"
rocprofv3 --kernel-trace -d gpurun_out/prof_ops -o p -- $CMD > gpurun_out/prof_ops.out 2> gpurun_out/prof_ops.err
DB=$(find gpurun_out/prof_ops -name "*.db" | head -1)
python3 tests/tools/op_table.py $DB gpurun_out/prof_ops.err "rocprofv3 --kernel-trace -- $CMD   ($LABEL)" > gpurun_out/op_table.txt
rm -rf gpurun_out/prof_ops
tail -1 gpurun_out/prof_ops.out | cut -c1-300
cat gpurun_out/op_table.txt
