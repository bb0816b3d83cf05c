#!/usr/bin/env python3
"""Kernel symbols of one forward, in launch order (development): run under `rocprofv3 --kernel-trace --output-format csv -d DIR -o p -- python3
tests/tools/kernel_list.py run B`, then `python3 tests/tools/kernel_list.py show DIR` prints the launches of the last forward."""
import csv, glob, os, re, sys
if sys.argv[1] == "run":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from bayes_od_rc_amd import synthetic
    from bayes_od_rc_amd.engine import Engine, make_config
    B = int(sys.argv[2])
    e = Engine(make_config((512, 512), batch=B, mc_samples=2))
    e.load_weights(synthetic.make_weights()); e.upload_images(synthetic.make_frames(B, 512, 512, seed=12))
    for _ in range(3):
        e.forward(None)
    e.synchronize(); e.close()
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    starts = [i for i, n in enumerate(names) if "stem" in n]
    last = starts[-1] if starts else 0
    # the last forward: from its stem kernel on
    stems = [i for i in starts]
    first_of_last = max(i for i in stems if all("stem" in names[j] or j == i for j in range(i, min(i + 1, len(names)))))
    for k, r in enumerate(rows[first_of_last:]):
        n = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "")
        print(k, n, "vgpr", r.get("VGPR_Count"), "agpr", r.get("Accum_VGPR_Count"), "lds", r.get("LDS_Block_Size"), "wg", r.get("Workgroup_Size"), "grid", r.get("Grid_Size"),
              "us", round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1))
