cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/j7; mkdir -p $out
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o clk -- python3 tools/op_bench.py attn layernorm p2cl > $out/clk.log 2>&1
python3 - <<PY
import csv, collections, glob
dur = {}
for r in csv.DictReader(open(glob.glob("$out/clk_kernel_trace.csv")[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
agg = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob("$out/clk_counter_collection.csv")[0])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    d = dur.get(r["Dispatch_Id"])
    if d: agg[d[1][:60]].append((float(r["Counter_Value"]), d[0]))
for k, v in agg.items():
    c = sum(a for a, _ in v) / len(v); t = sum(b for _, b in v) / len(v)
    print(f"{k:60s} n={len(v):3d} GUI_ACTIVE {c:12.0f}  dur {t/1e3:8.1f} us  -> {c / t:6.3f} cycles/ns")
PY
rm -f $out/*kernel_trace.csv
