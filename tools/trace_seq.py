"""Sequential listing of one step of a rocprofv3 --kernel-trace CSV: start offset (us), duration, gap before, short name, grid."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
ends = [i for i, n in enumerate(names) if "adam_kernel" in n]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(ends) - 2
lo, hi = ends[which - 1] + 1, ends[which] + 1
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"]); prev = t0
def short(n):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:50]
for i, r in enumerate(step):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{i:4d} {(s-t0)/1e3:9.1f} {(e-s)/1e3:7.1f} gap {(s-prev)/1e3:6.1f}  {short(r['Kernel_Name']):50s} grid {r.get('Grid_Size_X', r.get('Grid_Size',''))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size',''))}")
    prev = max(prev, e)
