#!/usr/bin/env python3
"""Print a slice of the kernel timeline (all streams) of a rocprofv3 rocpd database.

    python tools/rocpd_timeline.py results.db [first_kernel_substring] [n_rows]
    python tools/rocpd_timeline.py results.db --last-call     # everything from the last
                                                              # sample_indices_kernel launch on
"""
import sqlite3
import sys


def rows_of(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end, stream_id, queue_id from kernels order by start").fetchall()
    try:
        cp = cur.execute("select name, start, end from memory_copies order by start").fetchall()
        rows += [(f"[copy] {n}", s, e, -1, -1) for n, s, e in cp]
        rows.sort(key=lambda r: r[1])
    except sqlite3.Error:
        pass
    return rows


def show(rows, base):
    prev_end = {}
    for r in rows:
        gap = (r[1] - prev_end.get(r[3], r[1])) / 1000
        prev_end[r[3]] = r[2]
        print(f"{(r[1] - base) / 1000:9.2f} {(r[2] - base) / 1000:9.2f} dur={(r[2] - r[1]) / 1000:7.2f} "
              f"gap={gap:6.2f} stream={r[3]} queue={r[4]} {r[0][:60]}")


def main(path, needle="target_fused", n=60):
    rows = rows_of(path)
    if needle == "--last-call":
        k0 = max(i for i, r in enumerate(rows) if ("sample_indices_kernel" in r[0] or "learn_prologue_kernel" in r[0]))
        k0 = max(0, k0 - 3)
        print(f"last learn() call: {len(rows) - k0} device operations, "
              f"{(rows[-1][2] - rows[k0][1]) / 1000:.1f} us first start -> last end")
        show(rows[k0:], rows[k0][1])
        return
    k0 = len(rows) // 4
    for i in range(k0, len(rows)):
        if needle in rows[i][0]:
            k0 = i
            break
    show(rows[max(0, k0 - 4):k0 + int(n)], rows[k0][1])


if __name__ == "__main__":
    main(*sys.argv[1:])
