#!/usr/bin/env python3
"""Copy measurement files from gpurun_out/ into profiles/ with the HEAD they were measured on.

    python tools/stamp_profiles.py r05_z  bench_s20.log:bench_s20.json  kernel_stats.txt ...

Every copy says which commit produced it (two r04 files predated the final
code and nothing could tell): text files get a first line `# pearl_amd HEAD <sha> ...`, a JSON
object a `_stamp` key, JSON-lines files a first `{"_stamp": ...}` line.  The GPU box has no .git,
so the stamp is taken HERE, right after the call: commit first, measure, then run this — it
refuses to stamp a dirty tree unless --allow-dirty (then the stamp says so); --head=<commit> names
the commit of an earlier call whose files are copied later.
"""
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def git(*args):
    return subprocess.run(["git", "-C", REPO, *args], capture_output=True, text=True).stdout.strip()


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    allow_dirty = "--allow-dirty" in sys.argv
    prefix, items = args[0], args[1:]
    sha = git("rev-parse", "--short=12", "HEAD")
    for a in sys.argv[1:]:
        if a.startswith("--head="):     # the files come from an EARLIER call: name the commit it ran on
            sha = git("rev-parse", "--short=12", a.split("=", 1)[1])
            allow_dirty = True
    dirty = [ln for ln in git("status", "--porcelain", "--", "pearl_amd", "bench.py", "bench_algos.py",
                              "include", "tools", "oracle", "tests").splitlines() if ln.strip()]
    if dirty and not allow_dirty:
        sys.exit("stamp_profiles: the source tree differs from HEAD (commit first, or --allow-dirty):\n  "
                 + "\n  ".join(dirty[:10]))
    stamp = {"head": sha, "dirty_paths": len(dirty), "copied_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
             "note": "measured by a gpurun call on this HEAD's snapshot (one MI355X box)"}
    for it in items:
        src, _, dst = it.partition(":")
        dst = dst or os.path.basename(src)
        sp = os.path.join(REPO, "gpurun_out", src)
        dp = os.path.join(REPO, "profiles", f"{prefix}_{dst}")
        if not os.path.exists(sp):
            print(f"  (missing: gpurun_out/{src})")
            continue
        text = open(sp, errors="replace").read()
        lines = [ln for ln in text.splitlines() if ln.strip()]
        if dst.endswith(".json"):
            try:
                obj = json.loads(text)                 # a (possibly indented) JSON document
            except ValueError:
                js = [ln for ln in lines if ln.startswith("{")]      # a log whose last line is one
                obj = json.loads(js[-1]) if js else {"raw": text[-2000:]}
            obj["_stamp"] = stamp
            out = json.dumps(obj) + "\n"
        elif dst.endswith(".jsonl"):
            out = json.dumps({"_stamp": stamp}) + "\n" + "\n".join(ln for ln in lines if ln.startswith("{")) + "\n"
        else:
            out = f"# pearl_amd HEAD {sha}" + (f" + {len(dirty)} uncommitted path(s)" if dirty else "") + \
                  f", copied {stamp['copied_utc']}: {stamp['note']}\n" + text
        with open(dp, "w") as f:
            f.write(out)
        print(f"  profiles/{prefix}_{dst}  <- gpurun_out/{src}")


if __name__ == "__main__":
    main()
