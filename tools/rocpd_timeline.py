#!/usr/bin/env python3
"""Print a slice of the kernel timeline (all streams) of a rocprofv3 rocpd database.

    python tools/rocpd_timeline.py results.db [first_kernel_substring] [n_rows]
"""
import sqlite3
import sys


def main(path, needle="target_fused", n=60):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end, stream_id, queue_id from kernels order by start").fetchall()
    k0 = len(rows) // 2
    for i in range(k0, len(rows)):
        if needle in rows[i][0]:
            k0 = i
            break
    base = rows[k0][1]
    for r in rows[max(0, k0 - 4):k0 + int(n)]:
        print(f"{(r[1] - base) / 1000:9.2f} {(r[2] - base) / 1000:9.2f} dur={(r[2] - r[1]) / 1000:7.2f} "
              f"stream={r[3]} queue={r[4]} {r[0][:60]}")


if __name__ == "__main__":
    main(*sys.argv[1:])
