# effective shader clock per kernel of bench.py: GRBM_GUI_ACTIVE (cycles summed over the 8 XCDs) / kernel duration (usage: bash tools/clock_probe.sh <outdir>)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; mkdir -p $out
VPU_BENCH_GRAPH=0 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o clk -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/clk.log 2>&1
python3 - <<PY
import csv, collections, glob, re
dur = {}
for r in csv.DictReader(open(glob.glob("$out/clk_kernel_trace.csv")[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
agg = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob("$out/clk_counter_collection.csv")[0])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    d = dur.get(r["Dispatch_Id"])
    if d:
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", d[1])
        agg[(m.group(1) + (m.group(2) or "")) if m else d[1][:50]].append((float(r["Counter_Value"]), d[0]))
rows = sorted(agg.items(), key=lambda kv: -sum(b for _, b in kv[1]))
with open("$out/clock.txt", "w") as f:
    for k, v in rows[:40]:
        c = sum(a for a, _ in v); t = sum(b for _, b in v)
        line = f"{k:60s} n={len(v):4d} avg {t / len(v) / 1e3:8.1f} us  clock {c / t / 8:5.2f} GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)"
        print(line); f.write(line + "\n")
PY
rm -f $out/clk_kernel_trace.csv $out/clk_counter_collection.csv
