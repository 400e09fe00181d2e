#!/usr/bin/env python3
"""HBM traffic of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), per launch and
per transition, with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-byte
requests at 64 bytes: double it).  The passes must come from the single-stream loop
(PEARL_AMD_OVERLAP=0), where one target launch covers a whole window (10 240 transitions).

    python tools/pmc_traffic.py fetch/..._counter_collection.csv write/..._counter_collection.csv \
        --kernel target_split_kernel --transitions 10240 --algorithmic 1033 > profiles/r03_pmc_target.json

Only launches of the kernel's most frequent grid size are averaged (full windows).
"""
import argparse
import collections
import csv
import json


def rows_of(path, kernel, counter):
    out = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                out.append((int(row["Grid_Size"]), float(row["Counter_Value"])))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--kernel", default="target_split_kernel")
    ap.add_argument("--transitions", type=int, default=10240,
                    help="transitions covered by one launch of the selected grid size")
    ap.add_argument("--algorithmic", type=float, default=None, help="algorithmic bytes per transition")
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    fr, wr = rows_of(a.fetch_csv, a.kernel, "FETCH_SIZE"), rows_of(a.write_csv, a.kernel, "WRITE_SIZE")
    grid = collections.Counter(g for g, _ in fr).most_common(1)[0][0] if fr else 0
    fsel = [v for g, v in fr if g == grid]
    wsel = [v for g, v in wr if g == grid]
    fetch_kb = sum(fsel) / max(len(fsel), 1)
    write_kb = sum(wsel) / max(len(wsel), 1)
    hbm = (2.0 * fetch_kb + write_kb) * 1024.0
    print(json.dumps({
        "kernel": a.kernel, "grid_size": grid, "launches": [len(fsel), len(wsel)],
        "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
        "correction": "2 x FETCH_SIZE (gfx950: 128-byte requests tallied at 64 bytes) + WRITE_SIZE",
        "hbm_bytes_per_launch": hbm, "transitions_per_launch": a.transitions,
        "hbm_bytes_per_transition": hbm / a.transitions,
        "algorithmic_bytes_per_transition": a.algorithmic, "note": a.note,
    }, indent=1))


if __name__ == "__main__":
    main()
