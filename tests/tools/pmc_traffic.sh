#!/bin/bash
# HBM-traffic PMC passes over the bench command (run on the GPU box through gpurun); see pmc_traffic.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_fetch_b gpurun_out/pmc_write_b
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_b -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/pmc_fetch_b.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_b -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/pmc_write_b.log 2>&1
F=$(find gpurun_out/pmc_fetch_b -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/pmc_write_b -name "*counter_collection.csv" | head -1)
python3 tests/tools/pmc_traffic.py $F $W ${PMC_BATCH:-512} 10 512 512 > gpurun_out/head_conv_pmc.json
cat gpurun_out/head_conv_pmc.json
# keep the merge small: the raw csv files are large
rm -rf gpurun_out/pmc_fetch_b gpurun_out/pmc_write_b
