#!/usr/bin/env python3
"""Instruction mix of the kernels in a hipcc -S --cuda-device-only listing.

    hipcc -O3 ... --offload-arch=gfx950 -S --cuda-device-only -o k.s file.hip
    python tools/asm_mix.py k.s weight_grad rowpass
"""
import collections
import re
import sys


def main(path, *needles):
    lines = open(path).read().split("\n")
    starts = [(i, m.group(1)) for i, l in enumerate(lines) if (m := re.match(r"^(_Z\w+):", l))]
    for n, (i, name) in enumerate(starts):
        if needles and not any(x in name for x in needles):
            continue
        end = starts[n + 1][0] if n + 1 < len(starts) else len(lines)
        ops = collections.Counter()
        for l in lines[i + 1:end]:
            l = l.strip()
            if not l or l[0] in ".;/" or l.endswith(":"):
                continue
            ops[l.split()[0]] += 1
            if l.startswith("s_endpgm"):
                pass
        print(name[:80], "total", sum(ops.values()))
        print("   ", ", ".join(f"{k}:{v}" for k, v in ops.most_common(32)))


if __name__ == "__main__":
    main(*sys.argv[1:])
