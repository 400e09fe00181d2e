#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files (one or more passes).

    python tools/pmc_summary.py gpurun_out/pmc_*/dqn_counter_collection.csv
"""
import collections
import csv
import sys


def main(paths):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for p in paths:
        with open(p) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].split("(")[0][-46:]
                c = acc[name][row["Counter_Name"]]
                c[0] += float(row["Counter_Value"])
                c[1] += 1
    counters = sorted({c for k in acc.values() for c in k})
    print(f"{'kernel':46s} " + " ".join(f"{c[-16:]:>16s}" for c in counters))
    for name, cs in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", [0, 1])[0]):
        if not name.startswith(("pa::", "void pa::")) and "pa::" not in name:
            continue
        print(f"{name:46s} " + " ".join(
            f"{(cs[c][0] / cs[c][1]) if c in cs and cs[c][1] else float('nan'):16.1f}" for c in counters))


if __name__ == "__main__":
    main(sys.argv[1:])
