# rocprofv3 kernel-trace summary of tools/op_bench.py <op> -> gpurun_out/<dir>   (usage: bash tools/run_stats_op.sh <op> <dir>)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null
out="gpurun_out/$2"; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python3 tools/op_bench.py "$1" > $out/stats.log 2>&1
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$out/stats_kernel_stats.csv")))[:12]:
    print(f"{int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
PY
