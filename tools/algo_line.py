"""stdin: bench_algos.py JSON lines -> one short line per configuration."""
import json
import sys

for ln in sys.stdin:
    if ln.startswith("{"):
        d = json.loads(ln)
        ks = [(k["kernel"][:22], round(k["avg_launch_us"], 1)) for k in d.get("kernels", []) if "avg_launch_us" in k]
        print("  ", d["config"][:40], round(d["value"] / 1e6, 2), "M",
              round(d.get("ms_per_step", 0) * 1e3, 1), "us/step", ks)
