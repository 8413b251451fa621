#!/usr/bin/env python
"""Per-kernel roofline table of one recorded bench run: time per step and launch (rocprofv3 --kernel-trace --stats CSV), HBM
bytes per launch (the two --pmc passes, profiles/rNN_pmc_traffic.json), TFLOP/s of the GEMM instantiations (bench.py's own HIP-event
table, `all_gemm_variants`), and the fraction of the roof each one reaches -- 8 TB/s HBM, 2.5 PFLOP/s dense bf16 MFMA
(MI355X_MICROARCH.md).  usage: python tools/roofline_table.py profiles/r05 > profiles/r05_roofline_table.txt"""
import csv
import json
import re
import sys

pre = sys.argv[1] if len(sys.argv) > 1 else "profiles/r05"
rows = list(csv.DictReader(open(pre + "_bench_bs12_kernel_stats.csv")))
bench = json.loads(open(pre + "_bench.json").read().strip().splitlines()[-1])
traffic = json.load(open(pre + "_pmc_traffic.json"))
gemm = bench["roofline"].get("all_gemm_variants", {})
steps = max(int(r["Calls"]) for r in rows if "adam_kernel" in r["Name"])


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)I", n)
    if m:
        n = m.group(1)
    return re.sub(r"\(.*", "", n)


def lookup(table, name):
    if name in table:
        return table[name]
    base = name.split("<")[0]
    hits = [k for k in table if k.split("<")[0] == base or re.sub(r"_ZN12_GLOBAL__N_1\d+", "", k).startswith(base)]
    return table[hits[0]] if len(hits) == 1 else None


print(f"# {pre}: {bench['value']} images/s, {bench['ms_per_step']} ms per step; kernel time under the profiler "
      f"{sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e6:.2f} ms per step over {steps} steps")
print(f"{'kernel':58s} {'n/step':>6s} {'us':>8s} {'ms/step':>8s} {'MB rd':>8s} {'MB wr':>8s} {'TB/s':>6s} {'of 8':>5s} {'TFLOP/s':>8s} {'of 2500':>7s}")
tot = 0.0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    n, calls = short(r["Name"]), int(r["Calls"])
    per = calls / steps
    if per < 0.9:
        continue
    us = float(r["AverageNs"]) / 1e3
    ms = float(r["TotalDurationNs"]) / steps / 1e6
    tot += ms
    t = lookup(traffic, n)
    g = lookup(gemm, n)
    rd = wr = bw = None
    if t:
        rd, wr = t["read_bytes_per_launch"] / 1e6, t["write_bytes_per_launch"] / 1e6
        bw = (rd + wr) / us
    tf = g["TFLOP/s"] if g else None
    f = lambda v, w, p: (f"{v:{w}.{p}f}" if v is not None else " " * (w - 1) + "-")
    print(f"{n[:58]:58s} {per:6.1f} {us:8.1f} {ms:8.3f} {f(rd, 8, 1)} {f(wr, 8, 1)} {f(bw, 6, 2)} {f(bw / 8 if bw else None, 5, 2)} "
          f"{f(tf, 8, 1)} {f(tf / 2500 if tf else None, 7, 3)}")
    if ms < 0.02:
        break
print(f"# listed: {tot:.2f} ms per step")
