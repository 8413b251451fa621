"""Per-launch HBM traffic per kernel from the two PMC passes of tools/run_pmc_bench.sh.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B, so it is doubled
(MI355X_MICROARCH.md, HBM).  usage: python tools/pmc_traffic.py <dir> [out.json]"""
import csv, json, re, sys, collections
d = sys.argv[1]
def load(prefix, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f"{d}/{prefix}_counter_collection.csv")):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"])
        n = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:60]
        a = agg[n]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return agg
f, w = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
out = {}
for n in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, [0, 0])[1] + w.get(k, [0, 0])[1])):
    cf, vf = f.get(n, [0, 0.0]); cw, vw = w.get(n, [0, 0.0])
    c = max(cf, cw, 1)
    out[n] = {"launches": c, "read_bytes_per_launch": round(2 * vf * 1024 / max(cf, 1)), "write_bytes_per_launch": round(vw * 1024 / max(cw, 1))}
for n, v in list(out.items())[:25]:
    print(f"{n:70s} {v['launches']:6d}  rd {v['read_bytes_per_launch'] / 1e6:9.2f} MB  wr {v['write_bytes_per_launch'] / 1e6:9.2f} MB per launch")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
