#!/usr/bin/env python3
"""Summarises a `rocprofv3 --marker-trace --kernel-trace --output-format csv` run of a BOD_ROCTX=1 program: per roctx range the
number of occurrences, the mean host time of the range (the ENQUEUE of the stage) and the kernels dispatched inside it with
their mean device time.  usage: marker_summary.py <dir with *_marker_api_trace.csv and *_kernel_trace.csv> "<command line>" """
import csv, glob, os, sys
from collections import OrderedDict, defaultdict

root, what = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
mk = glob.glob(os.path.join(root, "**", "*marker_api_trace.csv"), recursive=True)[0]
kt = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)[0]
ranges = [(r["Function"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Thread_Id"])) for r in csv.DictReader(open(mk))
          if r["Function"].startswith("bod:")]
kernels = sorted((int(r["Correlation_Id"]), r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt)))
# the dispatches are matched to ranges through the order of their correlation ids relative to the ranges' (the marker rows carry
# correlation ids of the same per-process counter: a kernel enqueued inside a range has an id between the range's and the next range's)
mrows = sorted((int(r["Correlation_Id"]), r["Function"]) for r in csv.DictReader(open(mk)) if r["Function"].startswith("bod:"))
inner = [m for m in mrows if m[1] not in ("bod:infer", "bod:infer_async", "bod:collect", "bod:upload")]
stat = OrderedDict()
for name, t0, t1, _ in ranges:
    s = stat.setdefault(name, {"n": 0, "host_ns": 0, "kernels": defaultdict(lambda: [0, 0])})
    s["n"] += 1; s["host_ns"] += t1 - t0
ids = [m[0] for m in inner]
import bisect
for cid, kname, dur in kernels:
    i = bisect.bisect_right(ids, cid) - 1
    if i < 0:
        continue
    k = stat[inner[i][1]]["kernels"][kname.split("(")[0][:90]]
    k[0] += 1; k[1] += dur
print("# %s" % what)
print("# roctx ranges of the engine (BOD_ROCTX=1): occurrences, mean host time of the range = enqueue of the stage, kernels dispatched inside")
for name, s in stat.items():
    print("%-28s x%-3d host %8.1f us" % (name, s["n"], s["host_ns"] / s["n"] / 1e3))
    for kname, (n, dur) in sorted(s["kernels"].items(), key=lambda kv: -kv[1][1]):
        print("    %-92s x%-4d device %9.1f us each" % (kname, n, dur / n / 1e3))
