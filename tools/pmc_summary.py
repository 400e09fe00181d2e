#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files (one or more passes).

    python tools/pmc_summary.py gpurun_out/pmc_*/dqn_counter_collection.csv
    python tools/pmc_summary.py --chain-json profiles/rNN_pmc_chain.json <csv ...>
        also writes the matrix-pipe busy fraction of the DQN loop's kernels for bench.py's
        roofline.chain: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
"""
import json
import collections
import csv
import sys


def main(paths):
    chain_json = None
    if paths and paths[0] == "--chain-json":
        chain_json, paths = paths[1], paths[2:]
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
    if chain_json:
        out = {"formula": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs), per-launch averages, "
                          "separate rocprofv3 --pmc passes of the single-stream loop", "kernels": {}}
        for name, cs in acc.items():
            key = next((k for k in ("online_rowpass_h2_kernel", "online_rowpass_kernel", "weight_grad",
                                    "target_h2_kernel", "target_split_kernel")
                        if k in name), None)
            if key is None or "SQ_VALU_MFMA_BUSY_CYCLES" not in cs or "GRBM_GUI_ACTIVE" not in cs:
                continue
            busy = cs["SQ_VALU_MFMA_BUSY_CYCLES"][0] / cs["SQ_VALU_MFMA_BUSY_CYCLES"][1]
            act = cs["GRBM_GUI_ACTIVE"][0] / cs["GRBM_GUI_ACTIVE"][1]
            out["kernels"][name.strip()] = {"mfma_busy_cycles": busy, "grbm_gui_active": act,
                                            "launches": cs["GRBM_GUI_ACTIVE"][1],
                                            "mfma_busy_frac": busy / (128.0 * act)}
        with open(chain_json, "w") as f:
            json.dump(out, f)


if __name__ == "__main__":
    main(sys.argv[1:])
