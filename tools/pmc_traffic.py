#!/usr/bin/env python3
"""HBM traffic of target_fused_kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE),
per transition, with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-byte
requests at 64 bytes: double it).  The passes must come from the single-stream loop
(PEARL_AMD_OVERLAP=0), where one launch covers 10 240 transitions.

    python tools/pmc_traffic.py fetch/dqn_counter_collection.csv write/dqn_counter_collection.csv \
        > profiles/r01_pmc_target.json
"""
import csv
import json
import sys

TRANSITIONS_PER_LAUNCH = 10240   # target_update_freq (10) x batch (1024)


def avg(path, counter):
    tot, n, grids = 0.0, 0, set()
    with open(path) as f:
        for row in csv.DictReader(f):
            if "target_fused_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                if int(row["Grid_Size"]) != 2560 * 512:   # full windows only
                    continue
                tot += float(row["Counter_Value"])
                n += 1
    return tot / max(n, 1), n


def main(fetch_csv, write_csv):
    fetch_kb, nf = avg(fetch_csv, "FETCH_SIZE")
    write_kb, nw = avg(write_csv, "WRITE_SIZE")
    hbm = (2.0 * fetch_kb + write_kb) * 1024.0
    algorithmic = 256 * 4 + 16 * 16 * 4 + 16 + 4 + 1 + 4   # U row, one-hot table, mask, reward, term, y
    print(json.dumps({
        "kernel": "target_fused_kernel<32>", "launches": [nf, nw],
        "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
        "correction": "2 x FETCH_SIZE (gfx950: 128-byte requests tallied at 64 bytes) + WRITE_SIZE",
        "hbm_bytes_per_launch": hbm, "transitions_per_launch": TRANSITIONS_PER_LAUNCH,
        "hbm_bytes_per_transition": hbm / TRANSITIONS_PER_LAUNCH,
        "algorithmic_bytes_per_transition": algorithmic,
    }, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
