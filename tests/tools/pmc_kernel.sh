#!/bin/bash
# SQ counters of ONE kernel (name substring) over a short forward-only bench (run on the GPU box through gpurun) -> stdout
#   usage: pmc_kernel.sh <kernel name substring> [bench.py flags]      e.g.  pmc_kernel.sh stem_pool_fused --mc 1 --forward-only
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
K="$1"; shift
CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary $*"
rm -rf gpurun_out/pmc_k1 gpurun_out/pmc_k2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_k1 -o p -- $CMD > gpurun_out/pmc_k1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_DATA_FIFO_FULL --output-format csv -d gpurun_out/pmc_k2 -o p -- $CMD > gpurun_out/pmc_k2.log 2>&1
python3 - "$K" <<'PY'
import csv, glob, sys, collections
want = sys.argv[1]
for d in ("gpurun_out/pmc_k1", "gpurun_out/pmc_k2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print("%-28s launches=%d mean=%.4g" % (k, len(v), sum(v) / len(v)))
PY
rm -rf gpurun_out/pmc_k1 gpurun_out/pmc_k2
