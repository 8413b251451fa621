#!/usr/bin/env python
"""Instruction census per kernel of a device-only assembly dump (hipcc --cuda-device-only -S): how many of each class -- MFMA, packed
VALU (v_pk_*), transcendental, conversions, LDS, waits -- a kernel's body holds.  usage: python tools/isa_census.py file.s [filter]"""
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
TAB = "\t"
CLASSES = [("mfma", r"^\tv_mfma"), ("v_pk", r"^\tv_pk_"), ("valu", r"^\tv_"), ("trans", r"^\tv_(exp|log|rcp|rsq|sqrt)"), ("cvt", r"^\tv_cvt"),
           ("perm", r"^\tv_perm|^\tv_permlane|dpp|^\tds_bpermute|^\tds_swizzle"), ("ds", r"^\tds_"), ("vmem", r"^\t(buffer|global)_"),
           ("salu", r"^\ts_(?!waitcnt|nop|barrier)"), ("wait", r"^\ts_waitcnt"), ("nop", r"^\ts_nop"), ("barrier", r"^\ts_barrier")]
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt and flt not in name:
        continue
    lines = [l for l in body.split("\n") if l.startswith(TAB) and not l.startswith(TAB + ".") and not l.startswith(TAB + ";")]
    out = " ".join("%s %d" % (k, sum(re.search(p, l) is not None for l in lines)) for k, p in CLASSES)
    print("%-70s insts %d (~%d KB) %s" % (name[:70], len(lines), len(lines) * 6 // 1024, out))
