#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel stats + one step's timeline.

    python tools/rocpd_summary.py gpurun_out/prof/dqn_results.db > profiles/rXX_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute(
        "select name, count(*), avg(end-start)/1000.0, min(end-start)/1000.0, "
        "max(end-start)/1000.0, sum(end-start)/1e6, max(vgpr_count), max(accum_vgpr_count), "
        "max(lds_size), max(grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z)), "
        "max(workgroup_x) from kernels group by name order by sum(end-start) desc").fetchall()
    tot = sum(r[5] for r in rows)
    print(f"{'kernel':62s} {'calls':>6s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'tot_ms':>8s} "
          f"{'%':>5s} {'vgpr':>5s} {'lds':>7s} {'wgs':>5s} {'wgsz':>5s}")
    for r in rows:
        print(f"{r[0][:62]:62s} {r[1]:6d} {r[2]:8.2f} {r[3]:8.2f} {r[4]:8.2f} {r[5]:8.2f} "
              f"{100 * r[5] / tot:5.1f} {r[6]:5d} {r[8]:7d} {r[9]:5d} {r[10]:5d}")
    rows = cur.execute("select name, start, end, stream_id from kernels order by start").fetchall()
    k0 = len(rows) // 2
    for i in range(k0, len(rows)):
        if "target_fused" in rows[i][0]:
            k0 = i
            break
    base = rows[k0][1]
    print("\ntimeline around one learner step (us, relative to a target_fused launch):")
    for r in rows[max(0, k0 - 3):k0 + 14]:
        print(f"{(r[1] - base) / 1000:9.2f} {(r[2] - base) / 1000:9.2f} dur={(r[2] - r[1]) / 1000:7.2f} "
              f"stream={r[3]} {r[0][:70]}")


if __name__ == "__main__":
    main(sys.argv[1])
