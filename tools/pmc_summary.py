#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc CSV output (one directory per pass)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    table = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values
    for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "")
                if not k.startswith(("pa::", "void pa::")):
                    continue
                table[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(table, key=lambda k: -sum(table[k].get("GRBM_GUI_ACTIVE", [0]))):
        print(k[:100])
        for c, v in sorted(table[k].items()):
            print(f"    {c:28s} n={len(v):5d} avg={sum(v) / len(v):16.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
