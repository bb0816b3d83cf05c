#!/usr/bin/env python3
"""Sums the PMC counters of the row-reuse tower kernel over the dispatches of tests/tools/pmc_head_conv.sh's three
passes (rocprofv3 --pmc ... --output-format csv) and prints the derived ratios quoted in DESIGN.md section 5.1.
usage: pmc_head_summary.py <dir with pmc_sq1/ pmc_sq2/ pmc_tcc/> ["label of the run"]  > profiles/roundN_head_conv_counters.json"""
import csv, glob, json, os, sys

root = sys.argv[1]
tot = {}
# per launch position of a step (round 5): the tower kernel's four launches per step are, in order, layer 1 (three heads, plain tiles),
# layer 2 of the two continuing heads (plain), layer 2 of the regression head (sample-complete tiles, fused 1x1 + MC aggregation) and
# layer 3 (two heads, aggregating): the same symbol, different epilogues -- which flavour owns the LDS bank conflicts?
POS = ["layer 1 (3 heads, plain)", "layer 2 cls+cov (plain)", "layer 2 reg (aggregating)", "layer 3 cls+cov (aggregating)"]
pos = [dict() for _ in POS]
for sub in ("pmc_sq1", "pmc_sq2", "pmc_tcc"):
    for path in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fp:
            rows = [r for r in csv.DictReader(fp) if "4, 0, true" in r.get("Kernel_Name", "")]      # conv_igemm_kernel<256, 256, 2, 4, 0, true, false>: the tower kernel
        order = {d: i for i, d in enumerate(sorted({int(r["Dispatch_Id"]) for r in rows}))}
        for row in rows:
            tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            p_ = pos[order[int(row["Dispatch_Id"])] % len(POS)]
            p_[row["Counter_Name"]] = p_.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
label = sys.argv[2] if len(sys.argv) > 2 else "tests/tools/bench_head_conv.py, layer 1, B=8"
out = {"kernel": "conv_igemm_kernel<256,256,2,4,0,true> (%s)" % label, "counters": tot}
g = tot.get("GRBM_GUI_ACTIVE")
if g:
    # GRBM_* are reported summed over the 8 XCDs (GUI_ACTIVE / 8 / dispatches = the launch duration in shader clocks:
    # 2.78 M cycles = 1.33 ms at 2.09 GHz); SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs
    out["xcds"] = 8
    out["mfma_busy_fraction_of_1024_simds"] = tot.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (g / 8 * 1024)
    out["ta_busy_fraction"] = tot.get("GRBM_TA_BUSY", 0.0) / g
    if tot.get("SQ_INSTS_MFMA"):
        out["gui_active_cycles_per_mfma_instruction_per_simd"] = (g / 8 * 1024) / tot["SQ_INSTS_MFMA"]
h, m = tot.get("TCC_HIT_sum", 0.0), tot.get("TCC_MISS_sum", 0.0)
if h + m:
    out["l2_hit_rate"] = h / (h + m)
if tot.get("SQ_LDS_IDX_ACTIVE"):
    out["lds_bank_conflict_fraction_of_lds_cycles"] = tot.get("SQ_LDS_BANK_CONFLICT", 0.0) / tot["SQ_LDS_IDX_ACTIVE"]
per = {}
for name, c in zip(POS, pos):
    e = {}
    if c.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_fraction_of_lds_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    if c.get("GRBM_GUI_ACTIVE"):
        e["mfma_busy_fraction_of_1024_simds"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
        e["lds_active_fraction_of_256_cus"] = c.get("SQ_LDS_IDX_ACTIVE", 0.0) / (c["GRBM_GUI_ACTIVE"] / 8 * 256)
    if c.get("SQ_WAVE_CYCLES"):
        e["wait_any_fraction_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_INSTS_LDS"):
        e["lds_instructions"] = c["SQ_INSTS_LDS"]; e["lds_bank_conflict_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0)
    per[name] = e
out["per_launch_of_a_step"] = per
print(json.dumps(out, indent=1))
