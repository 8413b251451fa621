# in-situ per-kernel A/B: rocprofv3 kernel statistics of bench.py under two environments on ONE box
#   bash tools/run_stats_ab.sh <dir> "<ENV_A>" "<ENV_B>"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 -c 'import __graft_entry__ as g; g.build(lab=False)' > /dev/null
for k in A B; do
  if [ $k = A ]; then E="$2"; else E="$3"; fi
  out="gpurun_out/$1/$k"; mkdir -p $out
  for kv in $E; do export "$kv"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/stats.log 2>&1
  for kv in $E; do unset "${kv%%=*}"; done
done
python3 - <<PY
import csv
def load(k):
    rows=list(csv.DictReader(open("gpurun_out/$1/%s/stats_kernel_stats.csv"%k)))
    return {r["Name"]:(int(r["Calls"]),float(r["AverageNs"])/1e3,float(r["TotalDurationNs"])/1e6) for r in rows}
a,b=load("A"),load("B")
ta,tb=sum(v[2] for v in a.values()),sum(v[2] for v in b.values())
print("total kernel ms: A %.2f  B %.2f"%(ta,tb))
names=sorted(set(a)|set(b),key=lambda n:-max(a.get(n,(0,0,0))[2],b.get(n,(0,0,0))[2]))
for n in names[:40]:
    x,y=a.get(n,(0,0,0)),b.get(n,(0,0,0))
    print("%-64s A %5d x %7.1f us = %7.2f ms | B %5d x %7.1f us = %7.2f ms"%(n.replace("(anonymous namespace)::","").replace("void ","")[:64],x[0],x[1],x[2],y[0],y[1],y[2]))
PY
