#!/usr/bin/env python
"""Instruction mix of one kernel in a hipcc -S listing: per basic block (label) the instruction count, and for the
largest loop body the opcode histogram.   usage: python tools/isa_mix.py file.s <kernel-name-substring>"""
import collections, re, sys
src, want = open(sys.argv[1]).read().split("\n"), sys.argv[2]
start = next(i for i, l in enumerate(src) if l.endswith(":") is False and re.match(r"^_Z\w+:", l) and want in l)
end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
blocks, cur = collections.OrderedDict(), "entry"
blocks[cur] = []
for l in src[start + 1:end]:
    t = l.strip()
    if not t or t.startswith((";", ".")) and not re.match(r"^\.LBB\d+_\d+:", t):
        continue
    if re.match(r"^\.LBB\d+_\d+:", t):
        cur = t.split(":")[0]; blocks[cur] = []
        continue
    blocks[cur].append(t.split()[0])
tot = sum(len(b) for b in blocks.values())
print(f"{want}: {tot} instructions in {len(blocks)} blocks")
for k, b in blocks.items():
    if len(b) >= 40:
        c = collections.Counter(b)
        cls = collections.Counter()
        for op, n in c.items():
            key = ("mfma" if "mfma" in op else "exp/log/rcp" if re.match(r"v_(exp|log|rcp)", op) else "cvt" if op.startswith("v_cvt") else
                   "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else
                   "salu" if op.startswith("s_") else "other")
            cls[key] += n
        print(f"  {k:12s} {len(b):5d}  " + "  ".join(f"{a}={n}" for a, n in cls.most_common()))
        if len(b) >= 150:
            print("      " + "  ".join(f"{op}:{n}" for op, n in c.most_common(28)))
