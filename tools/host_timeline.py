#!/usr/bin/env python3
"""Host API calls and kernel executions of the LAST learn() call of a rocprofv3 run on one time
axis (csv output: --kernel-trace --hip-runtime-trace --output-format csv):

    python tools/host_timeline.py <hip_api_trace.csv> <kernel_trace.csv>

Shows when the host ENQUEUED each launch and when the device RAN it — whether a late kernel is the
host's doing (late enqueue) or the device's (queue / resources)."""
import csv
import sys


def rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def main(api_path, ker_path):
    api, ker = rows(api_path), rows(ker_path)
    ks = sorted(ker, key=lambda r: int(r["Start_Timestamp"]))
    # the last call: kernels after the last learn_prologue_kernel
    start = max(i for i, r in enumerate(ks) if "learn_prologue" in r["Kernel_Name"])
    ks = ks[start:]
    t0 = int(ks[0]["Start_Timestamp"])
    by_corr = {}
    for r in api:
        if "Launch" in r["Function"] or "Memset" in r["Function"] or "Memcpy" in r["Function"] or "Synchronize" in r["Function"] or "Event" in r["Function"]:
            by_corr[r["Correlation_Id"]] = r
    first_api = None
    print(f"{'enqueued':>9s} {'api_us':>7s} {'start':>9s} {'end':>9s} {'dur':>7s}  kernel")
    for r in ks[:80]:
        a = by_corr.get(r["Correlation_Id"])
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        if a:
            q = (int(a["Start_Timestamp"]) - t0) / 1e3
            qd = (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
            first_api = q if first_api is None else first_api
            print(f"{q:9.1f} {qd:7.1f} {s:9.1f} {e:9.1f} {e - s:7.1f}  {r['Kernel_Name'][:70]}")
        else:
            print(f"{'?':>9s} {'':7s} {s:9.1f} {e:9.1f} {e - s:7.1f}  {r['Kernel_Name'][:70]}")
    # every API call of the window around the call start (events, memsets, syncs included)
    lo = t0 - 150_000
    print("\nhost API calls from 150 us before the prologue started to 120 us after:")
    for r in sorted(api, key=lambda r: int(r["Start_Timestamp"])):
        s = int(r["Start_Timestamp"])
        if lo <= s <= t0 + 120_000:
            print(f"{(s - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - s) / 1e3:7.1f}  {r['Function']}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
