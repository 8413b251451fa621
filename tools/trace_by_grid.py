"""Average duration per (kernel, grid) of a rocprofv3 --kernel-trace CSV.  usage: python tools/trace_by_grid.py <csv> [name-substring]"""
import collections, csv, re, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if len(sys.argv) > 2 and sys.argv[2] not in n:
        continue
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", n)
    agg[((m.group(1) + (m.group(2) or "")) if m else n[:50], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Grid_Size_Y", ""))].append(
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f"{len(v):5d} x  med {v[len(v) // 2]:8.1f} us  min {v[0]:8.1f}   {k[0]}  grid {k[1]} x {k[2]}")
