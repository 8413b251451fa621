#!/usr/bin/env python
"""Per-kernel instruction census of a device-only assembly dump (hipcc --cuda-device-only -S): MFMAs, LDS-DMA pieces, stores,
barriers, s_waitcnt vmcnt(0) and scratch accesses per kernel -- the things a hand-counted vmcnt pipeline breaks on.
usage: python tools/isa_k5.py file.s [name-filter]"""
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt and flt not in name:
        continue
    lines = body.split("\n")
    cnt = lambda pat: sum(re.search(pat, l) is not None for l in lines)
    vm0 = cnt(r's_waitcnt.*vmcnt\(0\)')
    dma = cnt(r'buffer_load.* lds')
    print(f"{name[-46:]:46s} lines {len(lines):6d} mfma {cnt('v_mfma'):5d} dma {dma:4d} "
          f"store {cnt('buffer_store'):3d} barrier {cnt('s_barrier'):3d} vmcnt0 {vm0:3d} "
          f"scratch {cnt('scratch_'):3d} readlane {cnt('v_readlane'):4d} writelane {cnt('v_writelane'):4d}")
