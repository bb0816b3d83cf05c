#!/usr/bin/env python3
"""Derives the `s_waitcnt lgkmcnt(n)` tables of the mid-tile-barrier tower loop (csrc/conv_igemm.hip, ABL = 6) by replaying its LDS
issue order: LDS reads return in order, so the wait in front of a step's MFMAs may leave exactly the reads issued AFTER the youngest
operand in flight.  Per K-tile of 16 steps: A(s + AHEAD) goes out in step s (s + AHEAD <= 15), the k-step-1 B set behind steps 2..5,
the next K-tile's k-step-0 B set behind steps 9..12 (or, at a group's last K-tile, behind the barrier in front of step 14: two per
step), the next K-tile's first AHEAD A fragments behind the barrier (AHEAD - 1 in step 14, one in step 15)."""
import sys


def sim(ahead, prev_last, last):
    q = []

    def tail(group_last):
        s14 = ["A'(%d)" % i for i in range(ahead - 1)]
        s15 = ["A'(%d)" % (ahead - 1)]
        if group_last:
            s14 += ["Bn0", "Bn1"]
            s15 += ["Bn2", "Bn3"]
        return s14, s15
    s14, s15 = tail(prev_last)
    q += [x.replace("A'", "A").replace("Bn", "Bc") for x in s14 + s15]
    waits = []
    for st in range(16):
        if st == 14:
            q.clear()                                    # lgkmcnt(0) in front of the barrier
            s14, s15 = tail(last)
        if st < 14:
            if st + ahead <= 15:
                q.append("A(%d)" % (st + ahead))
            if 2 <= st <= 5:
                q.append("Bk%d" % (st - 2))
            if not last and 9 <= st <= 12:
                q.append("Bn%d" % (st - 9))
        else:
            q += s14 if st == 14 else s15
        need = ["A(%d)" % st] + (["Bc%d" % j for j in range(4)] if st == 0 else []) + (["Bk%d" % j for j in range(4)] if st == 8 else [])
        idx = [q.index(n) for n in need if n in q]
        waits.append(len(q) - 1 - max(idx) if idx else -1)
    return waits


if __name__ == "__main__":
    ahead = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    print("AHEAD", ahead)
    print("W_SAME (kxc < 2) :", sim(ahead, False, False))
    print("W_LAST (kxc == 2):", sim(ahead, False, True))
    print("group-first K-tile (previous = group-last), step 0 uses its value, later steps the smaller W_SAME:", sim(ahead, True, False))
