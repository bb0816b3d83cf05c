"""CPU: the per-op roofline table tool (tests/tools/op_table.py) on a synthetic rocprofv3 database and plan dump -- the join by
launch order (fused stem: the pool op launches nothing; raw-flavour ops are skipped; a split-K reduce belongs to the launch before
it), the per-op figures, the queue-idle column and the stream idle in front of a forward."""
import os
import sqlite3
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TOOL = os.path.join(HERE, "tools", "op_table.py")


def _make_db(path, forwards):
    con = sqlite3.connect(path)
    con.execute("create table kernels (name text, start integer, end integer, grid_x integer, grid_y integer, grid_z integer, workgroup_x integer)")
    t = 0
    for f in range(forwards):
        t += 5000                                    # 5 us of idle stream in front of every forward
        for name, dur in (("stem_pool_fused_kernel<2>(float const*)", 1000000), ("void pw_conv_kernel<64, 64, false, 1, 256, 2>(ConvArgs, int, int, int)", 500000),
                          ("void conv_igemm_kernel<128, 128, 2, 2, 0, false, false>(ConvArgs)", 200000), ("conv_splitk_reduce_kernel(ConvArgs)", 10000),
                          ("void conv_igemm_kernel<256, 256, 2, 4, 0, true, false>(ConvArgs)", 4000000)):
            con.execute("insert into kernels values (?,?,?,?,?,?,?)", (name, t, t + dur, 256, 1, 1, 256))
            t += dur + (2000 if "true, false" not in name else 0)            # 2 us gaps except in front of the posterior
        con.execute("insert into kernels values (?,?,?,?,?,?,?)", ("void post_sample_kernel<8>(PostCfg, PostBuffers, int)", t, t + 1000, 256, 1, 1, 256))
        t += 1000
    con.commit()
    con.close()


def test_op_table_joins_plan_and_trace(tmp_path):
    db = str(tmp_path / "r.db")
    _make_db(db, 3)
    plan = tmp_path / "err.txt"
    plan.write_text("noise\n"
                    "# ops: index name kind flavour M taps cin cout groups fan has_res fused_next flops bytes\n"
                    "# op 0 conv1(stem) 0 0 0 0 0 0 0 0 0 0 1e12 4.6e9\n"
                    "# op 1 pool1 1 0 100 0 0 0 1 1 0 0 0 0\n"
                    "# op 2 res2a_branch1 2 0 100 1 64 256 1 1 0 1 5e11 2.3e9\n"
                    "# op 3 P6 2 0 100 9 2048 256 1 1 0 0 2.5e11 4.6e8\n"
                    "# op 4 head_tower_layer_2(raw) 2 1 100 9 256 256 2 1 0 0 1e13 1e9\n"
                    "# op 5 head_tower_layer_2(aggregating) 2 2 100 9 256 256 2 1 0 0 1e13 9.2e9\n")
    r = subprocess.run([sys.executable, TOOL, db, str(plan), "header line"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[0] == "# header line" and "mean of the last 3 forwards" in lines[1]
    rows = {l.split()[0]: l.split() for l in lines if l and not l.startswith("#") and not l.startswith("op ")}
    assert set(rows) == {"conv1(stem)", "res2a_branch1", "P6", "head_tower_layer_2(aggregating)"}          # pool and raw flavour launch nothing
    # stem: 1000 us for 4.6 GB = 4 600 GB/s, 1e12 flops = 1 000 TFLOP/s; byte floor 1000 us at 4.6 TB/s -> ratio 1.00
    s = rows["conv1(stem)"]
    assert s[1].startswith("stem_pool_fused") and abs(float(s[2]) - 1000.0) < 0.1 and abs(float(s[3]) - 4600) < 1 and abs(float(s[-2]) - 1.0) < 0.01
    # P6: the split-K reduce is part of the launch (200 + 10 us); 2 us of queue idle in front of it
    p6 = rows["P6"]
    assert abs(float(p6[2]) - 210.0) < 0.1 and abs(float(p6[-1]) - 2.0) < 0.05
    # the tower launch: MFMA floor 4 000 us at 2.5 PFLOP/s = its time -> ratio 1.00
    tw = rows["head_tower_layer_2(aggregating)"]
    assert abs(float(tw[2]) - 4000.0) < 0.1 and abs(float(tw[-2]) - 1.0) < 0.01
    tail = "\n".join(lines[-3:])
    assert "stem + backbone + FPN launches: 1710.0 us" in tail
    assert "stream idle in front of a forward" in tail and "5, 5 us" in tail              # forwards 2 and 3 (the first has no predecessor)
    assert "4.0 us in all" in tail                                                         # 2 us in front of res2a_branch1 and of P6


def test_op_table_reports_a_plan_mismatch(tmp_path):
    db = str(tmp_path / "r.db")
    _make_db(db, 1)
    plan = tmp_path / "err.txt"
    plan.write_text("# op 0 conv1(stem) 0 0 0 0 0 0 0 0 0 0 1e12 4.6e9\n# op 1 res2a 2 0 100 1 64 256 1 1 0 0 5e11 2.3e9\n")
    r = subprocess.run([sys.executable, TOOL, db, str(plan)], capture_output=True, text=True)
    assert r.returncode == 0 and "no forward in the trace matches the plan (2 launching ops)" in r.stdout
