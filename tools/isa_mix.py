#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a `hipcc -S --cuda-device-only` listing: MFMA / exp / other VALU / LDS / VMEM
counts and the most frequent VALU opcodes - how the per-element select in the attention backward's loop was found.

    hipcc -S --offload-arch=gfx950 --cuda-device-only -O3 -std=c++17 -Icsrc csrc/attention_bwd.hip -o /tmp/a.s
    python tools/isa_mix.py /tmp/a.s attn_bwd_dkv_kernelILb0ELb1E [min_instructions]"""
import collections
import re
import sys


def main(path, needle, minn=80):
    lines = open(path).read().splitlines()
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and needle in l]
    if not starts:
        raise SystemExit("no symbol containing %r" % needle)
    start = starts[0]
    end = [i for i, l in enumerate(lines) if i > start and re.match(r"^_Z\w+:", l)]
    seg = lines[start:(end[0] if end else len(lines))]
    print(lines[start][:120])
    blocks, cur = [], None
    for l in seg:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = [m.group(1), collections.Counter()]
            blocks.append(cur)
        elif cur is not None:
            t = l.strip().split()
            if t and not t[0].startswith((".", ";", "//")):
                cur[1][t[0]] += 1
    for name, c in blocks:
        n = sum(c.values())
        if n < minn:
            continue
        grp = lambda p: sum(v for k, v in c.items() if k.startswith(p))
        valu = grp("v_") - grp("v_mfma")
        print("%s: %d instr | mfma %d exp %d valu %d lds %d vmem %d waitcnt %d" % (name, n, grp("v_mfma"), grp("v_exp"), valu, grp("ds_"),
                                                                                    grp("global_") + grp("buffer_"), grp("s_waitcnt")))
        print("    ", [(k, v) for k, v in c.most_common(20) if k.startswith("v_") and not k.startswith("v_mfma")][:14])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 80)
