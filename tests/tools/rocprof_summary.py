#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace results .db (rocpd sqlite) into the text table committed
under profiles/.   usage: rocprof_summary.py results.db "header line" > profiles/xxx.txt"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
rows = cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
                   "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
for line in sys.argv[2:]:
    print("# " + line)
print("%-72s %6s %12s %7s %10s %10s %10s" % ("kernel", "calls", "total_ms", "pct", "avg_us", "min_us", "max_us"))
for r in rows:
    print("%-72s %6d %12.3f %6.1f%% %10.1f %10.1f %10.1f" % (r[0][:72], r[1], r[2], 100 * r[2] / tot, r[3], r[4], r[5]))
print()
print("# largest launches by grid (the head-tower 3x3 convs are the grid.z>=2 conv_igemm launches):")
rows = cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), avg(end-start)/1e3, vgpr_count, "
                   "accum_vgpr_count, lds_size from kernels group by name, grid_x, grid_y, grid_z "
                   "order by 7 desc limit 10").fetchall()
for r in rows:
    print("%-48s grid=(%d,%d,%d) wg=%d calls=%d avg_us=%.1f vgpr=%s agpr=%s lds=%s" % ((r[0][:48],) + tuple(r[1:])))

print()
print("# launches grouped by (kernel, grid), by total time:")
kcols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
sid = "stream_id" if "stream_id" in kcols else "0"
rows = cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), sum(end-start)/1e6, avg(end-start)/1e3, %s "
                   "from kernels group by name, grid_x, grid_y, grid_z, %s order by 7 desc limit 40" % (sid, sid)).fetchall()
for r in rows:
    print("%-48s grid=(%d,%d,%d) wg=%d calls=%d total_ms=%.2f avg_us=%.1f stream=%s" % ((r[0][:48],) + tuple(r[1:])))
print()
print("# per stream and kernel name:")
for r in cur.execute("select %s, name, count(*), sum(end-start)/1e6 from kernels group by %s, name order by 1, 4 desc" % (sid, sid)).fetchall():
    if r[3] >= 0.5: print("stream=%s %-64s calls=%d total_ms=%.2f" % (r[0], r[1][:64], r[2], r[3]))

cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
for key in ("stream_id", "queue_id"):
    if key in cols:
        print()
        print("# kernel time per %s:" % key)
        for r in cur.execute("select %s, count(*), sum(end-start)/1e6, (max(end)-min(start))/1e6 from kernels group by %s order by 3 desc" % (key, key)).fetchall():
            print("%s=%s launches=%d busy_ms=%.2f span_ms=%.2f" % (key, r[0], r[1], r[2], r[3]))
