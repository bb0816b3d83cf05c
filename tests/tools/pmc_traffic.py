#!/usr/bin/env python3
"""HBM traffic of the head-tower convolutions from two rocprofv3 PMC passes over bench.py
(one --pmc FETCH_SIZE, one --pmc WRITE_SIZE; MI355X_MICROARCH.md: separate passes, FETCH_SIZE is in KiB and
under-counts wide coalesced reads by 2x on gfx950, WRITE_SIZE in KiB).

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> B N H W [raw] > profiles/roundX_head_conv_pmc.json
("raw": the plan without the fused MC aggregation, BOD_FUSE_AGGREGATION=0 -- the last layers then write the fp32 [B,N,A,.] tensors)

The head launches of a step are the four conv_igemm dispatches that precede post_sample_kernel."""
import csv, json, sys

def dispatches(path, counter):
    rows = {}
    with open(path) as fp:
        for r in csv.DictReader(fp):
            if r["Counter_Name"] != counter:
                continue
            d = int(r["Dispatch_Id"])
            e = rows.setdefault(d, {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "value": 0.0})
            e["value"] += float(r["Counter_Value"])
    return [rows[k] for k in sorted(rows)]

fused_agg = not (len(sys.argv) > 7 and sys.argv[7] == "raw")
NHEAD = 5 if fused_agg else 4          # plans with the fused aggregation run tower layer 2 as two launches (round 4)


def head_launches(ds):
    out = []
    for i, d in enumerate(ds):
        if d["name"].startswith("void post_sample_kernel"):
            convs = [x for x in ds[:i] if "conv_igemm_kernel" in x["name"]][-NHEAD:]
            out.append(convs)
    return out

fetch, write = dispatches(sys.argv[1], "FETCH_SIZE"), dispatches(sys.argv[2], "WRITE_SIZE")
B, N, H, W = (int(v) for v in sys.argv[3:7])
P = sum(((H + (1 << l) - 1) >> l) * ((W + (1 << l) - 1) >> l) for l in range(3, 8))
A = 9 * P
hf, hw = head_launches(fetch), head_launches(write)
steps = min(len(hf), len(hw))
assert steps >= 1, "no post_sample_kernel dispatch found"
act = lambda rows, heads: rows * 256 * 2 * heads
wts = lambda heads: 9 * 256 * 256 * 2 * heads
raw = {"cls": B * N * A * 8 * 4, "box": B * N * A * 4 * 4, "cov": B * N * A * 10 * 4}
agg = {"cls": B * A * 8 * 4, "box": B * A * 16 * 4, "cov": B * A * 10 * 4}       # per-anchor MC statistics (fused aggregation)
out = agg if fused_agg else raw
tail = "fused 1x1 + MC aggregation -> per-anchor statistics" if fused_agg else "fused 1x1 -> fp32 [B,N,A,.] outputs"
algo = [  # (label, read bytes, write bytes)
    ("head layer 0 (de-duplicated, %d-way dropout fan-out)" % N, act(B * P, 1) + wts(3), act(B * N * P, 3)),
    ("head layer 1", act(B * N * P, 3) + wts(3), act(B * N * P, 3)),
]
if fused_agg:
    algo += [("head layer 2, classification + covariance towers (plain tiles)", act(B * N * P, 2) + wts(2), act(B * N * P, 2)),
             ("head layer 2, regression tower ends: %s" % tail, act(B * N * P, 1) + wts(1), out["box"])]
else:
    algo += [("head layer 2 (regression tower ends: %s)" % tail, act(B * N * P, 3) + wts(3), act(B * N * P, 2) + out["box"])]
algo += [("head layer 3 (cls + cov: %s)" % tail, act(B * N * P, 2) + wts(2), out["cls"] + out["cov"])]
launches = []
for k in range(NHEAD):
    f = sum(hf[s][k]["value"] for s in range(steps)) / steps
    w = sum(hw[s][k]["value"] for s in range(steps)) / steps
    launches.append({"launch": algo[k][0], "kernel": hf[0][k]["name"][:80], "grid_threads": hf[0][k]["grid"],
                     "FETCH_SIZE_KB_raw": round(f), "WRITE_SIZE_KB": round(w),
                     "hbm_read_bytes_corrected": round(f * 1024 * 2), "hbm_write_bytes": round(w * 1024),
                     "algorithmic_read_bytes": algo[k][1], "algorithmic_write_bytes": algo[k][2]})
tot = sum(l["hbm_read_bytes_corrected"] + l["hbm_write_bytes"] for l in launches)
print(json.dumps({
    "what": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary",
    "mc_aggregation_fused": fused_agg,
    "config": {"height": H, "width": W, "mc_samples": N, "batch": B},
    "correction": "FETCH_SIZE x 1024 x 2 (gfx950 half-count of wide coalesced reads), WRITE_SIZE x 1024",
    "steps_averaged": steps, "launches": launches,
    "hbm_bytes_per_step_head_convs": tot, "avg_hbm_bytes_per_launch": round(tot / NHEAD),
    "algorithmic_bytes_per_step_head_convs": sum(a[1] + a[2] for a in algo)}, indent=1))
