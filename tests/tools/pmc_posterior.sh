#!/bin/bash
# HBM bytes the UNFUSED posterior really moves (round-4 review: bench.py quoted SURVEY 8d's 49.5 MB per image over the stage's time --
# 9.2 TB/s, above HBM's 8 -- although, since the compaction kernel, the per-anchor fusion reads box / covariance samples of the kept
# anchors only): FETCH_SIZE / WRITE_SIZE of the post_* kernels (separate PMC passes, MI355X_MICROARCH.md; FETCH_SIZE in KiB and
# doubled on gfx950, WRITE_SIZE in KiB) over a bench run planned WITHOUT the fused MC aggregation (raw [B,N,A,.] tensors).
#   usage: pmc_posterior.sh [batch]   -> gpurun_out/posterior_pmc.json (copy to profiles/roundN_posterior_pmc.json)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B=${1:-512}
CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --batch $B"
rm -rf gpurun_out/pmc_pf gpurun_out/pmc_pw
BOD_FUSE_AGGREGATION=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_pf -o p -- $CMD > gpurun_out/pmc_pf.log 2>&1
BOD_FUSE_AGGREGATION=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_pw -o p -- $CMD > gpurun_out/pmc_pw.log 2>&1
python3 - $B <<'PY' > gpurun_out/posterior_pmc.json
import csv, glob, json, sys, collections
B = int(sys.argv[1])
def per_kernel(d, counter):
    acc = collections.defaultdict(lambda: [0.0, set()])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "post_" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[k][0] += float(r["Counter_Value"]); acc[k][1].add(r["Dispatch_Id"])
    return {k: (v[0], len(v[1])) for k, v in acc.items()}
fe, wr = per_kernel("gpurun_out/pmc_pf", "FETCH_SIZE"), per_kernel("gpurun_out/pmc_pw", "WRITE_SIZE")
kern, total = {}, 0.0
for k in sorted(set(fe) | set(wr)):
    f, nf = fe.get(k, (0.0, 1)); w, nw = wr.get(k, (0.0, 1))
    rd, wt = f / max(nf, 1) * 1024 * 2, w / max(nw, 1) * 1024            # bytes per dispatch (FETCH_SIZE x2: gfx950 correction)
    kern[k] = {"hbm_read_bytes_corrected": int(rd), "hbm_write_bytes": int(wt), "dispatches": nf}
    total += rd + wt
P = sum(((512 + (1 << l) - 1) >> l) ** 2 for l in range(3, 8))
print(json.dumps({"config": {"height": 512, "width": 512, "mc_samples": 10, "batch": B, "plan": "BOD_FUSE_AGGREGATION=0 (raw [B,N,A,.] tensors)"},
                  "kernels": kern, "hbm_bytes_per_step": int(total),
                  "algorithmic_bytes_per_step_survey_8d": int((10 * (4 + 10 + 8) * 4 + 16 + (4 + 16 + 8) * 4) * B * 9 * P)}, indent=1))
PY
cat gpurun_out/posterior_pmc.json
rm -rf gpurun_out/pmc_pf gpurun_out/pmc_pw
