#!/usr/bin/env python3
"""Per-op roofline table of the forward pass: joins the plan dump of the library (BOD_DUMP_OPS=1, stderr of the profiled command)
with a rocprofv3 --kernel-trace results .db (rocpd sqlite) by launch order.

    usage: op_table.py results.db bench.stderr "header" > gpurun_out/op_table.txt

A forward starts at a stem kernel; its conv-family launches follow in plan order (the max-pool op launches nothing when the stem is
the fused kernel, flavour-1 ("raw") ops are not run by bod_infer, a split-K reduce belongs to the launch before it).  Times are the
MEAN over the last forwards that match the plan; GB/s = algorithmic bytes / time (input pixels, outputs, shortcut and weights once),
`hbm` = the time those bytes take at 4.6 TB/s (the copy rate tests/tools/hbm_rw_probe.hip measures on this part), `mfma` = the
op's flops at 2.5 PFLOP/s.  The ratio column is time / max(hbm, mfma): what a perfect kernel for that op would still have to gain;
gap_us = idle time of the queue in front of the launch.
"""
import re
import sqlite3
import sys

HBM = 4.6e12
MFMA = 2.5e15


def short(name):
    m = re.match(r"void (\w+)<([^>]*)>", name)
    if m:
        return "%s<%s>" % (m.group(1).replace("_kernel", ""), m.group(2).replace(" ", ""))
    return name.split("(")[0].replace("_kernel", "")


def main():
    db, ops_file = sys.argv[1], sys.argv[2]
    ops = []
    for line in open(ops_file, errors="replace"):
        if line.startswith("# op "):
            f = line.split()
            ops.append(dict(index=int(f[2]), name=f[3], kind=int(f[4]), flavour=int(f[5]), M=int(f[6]), taps=int(f[7]), cin=int(f[8]),
                            cout=int(f[9]), groups=int(f[10]), fan=int(f[11]), res=int(f[12]), nxt=int(f[13]), flops=float(f[14]),
                            bytes=float(f[15])))
    if not ops:
        sys.exit("no '# op' lines in %s (BOD_DUMP_OPS=1?)" % ops_file)
    # the plan is dumped once per handle: keep the LAST dump (the bench's timed handle is created last among equal shapes)
    last0 = max(i for i, o in enumerate(ops) if o["index"] == 0)
    ops = ops[last0:]
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall()
    fam = ("conv_igemm", "pw_conv", "slide3x3", "stem_", "conv_splitk_reduce")
    fwd, cur = [], None
    for r in rows:
        n = r[0]
        if "stem_conv" in n or "stem_pool_fused" in n:
            cur = [r]
            fwd.append(cur)
        elif cur is not None and any(k in n for k in fam):
            cur.append(r)
        elif cur is not None and ("post_" in n or "nms" in n or "cluster" in n):
            cur = None
    # idle time of the stream in front of each forward: the stem's start minus the end of the last kernel before it (any kernel of the
    # process that is not on a side queue would do; the trace is in start order, so take the latest end among the earlier kernels)
    step_gaps, latest_end = [], None
    for r in rows:
        if ("stem_conv" in r[0] or "stem_pool_fused" in r[0]) and latest_end is not None:
            step_gaps.append((r[1] - latest_end) / 1e3)
        if not any(k in r[0] for k in ("nms", "cluster")):        # (side-stream kernels of the previous batch overlap the next forward)
            latest_end = r[2] if latest_end is None else max(latest_end, r[2])
    fused_stem = any("stem_pool_fused" in r[0] for r in rows)
    want = [o for o in ops if o["flavour"] != 1 and not (o["kind"] == 1 and fused_stem)]
    good = []
    for f in fwd:
        launches = []
        for r in f:
            if "conv_splitk_reduce" in r[0] and launches:
                launches[-1] = launches[-1] + [r]
            else:
                launches.append([r])
        if len(launches) == len(want):
            good.append(launches)
    for line in sys.argv[3:]:
        print("# " + line)
    if not good:
        print("# no forward in the trace matches the plan (%d launching ops); launches per forward seen: %s" % (len(want), sorted({len(f) for f in fwd})))
        return
    good = good[-6:]
    print("# mean of the last %d forwards; bytes = algorithmic (in + out + shortcut + weights, once); hbm @ %.1f TB/s, mfma @ %.1f PFLOP/s" % (len(good), HBM / 1e12, MFMA / 1e15))
    print("%-36s %-44s %9s %8s %8s %8s %8s %6s %7s" % ("op", "kernel", "us", "GB/s", "TFLOP/s", "hbm_us", "mfma_us", "ratio", "gap_us"))
    tot = tot_floor = 0.0
    front = front_floor = front_gap = 0.0
    for i, o in enumerate(want):
        us = sum(sum(r[2] - r[1] for r in g[i]) for g in good) / len(good) / 1e3
        # idle time of the queue in front of this launch: its start minus the previous launch's end
        gap = sum(max(0, g[i][0][1] - g[i - 1][-1][2]) for g in good) / len(good) / 1e3 if i > 0 else 0.0
        k = short(good[-1][i][0][0])
        hbm_us = o["bytes"] / HBM * 1e6
        mfma_us = o["flops"] / MFMA * 1e6
        floor = max(hbm_us, mfma_us, 1e-9)
        print("%-36s %-44s %9.1f %8.0f %8.1f %8.1f %8.1f %6.2f %7.1f" % (o["name"][:36], k[:44], us, o["bytes"] / us / 1e3 if us else 0, o["flops"] / us / 1e6 if us else 0,
                                                                     hbm_us, mfma_us, us / floor, gap))
        tot += us
        tot_floor += floor
        if "head_" not in o["name"] and not o["name"].startswith(("cls", "reg", "cov")):
            front += us
            front_floor += floor
            front_gap += gap
    print("# stem + backbone + FPN launches: %.1f us, floors %.1f us; all launches %.1f us, floors %.1f us" % (front, front_floor, tot, tot_floor))
    if step_gaps:
        print("# stream idle in front of a forward (stem start - end of the last main-stream kernel before it), last steps: %s us" %
              ", ".join("%.0f" % g for g in step_gaps[-5:]))
    print("# queue idle between the stem + backbone + FPN launches (start of a launch - end of the one before): %.1f us in all" % front_gap)


if __name__ == "__main__":
    main()
