# which kernels pay for running under the gradient reducer?  Kernel statistics of the eager step without / with the forced
# reducer (world size 1) on one box:   bash tools/dp_stats_ab.sh <dir>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 -c 'import __graft_entry__ as g; g.build(lab=False)' > /dev/null
for k in none reducer; do
  out="gpurun_out/$1/$k"; mkdir -p $out
  export DP_ONLY=$k
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python3 tools/dp_rehearsal.py 10 > $out/stats.log 2>&1
done
python3 - <<PY
import csv
def load(k):
    rows=list(csv.DictReader(open("gpurun_out/$1/%s/stats_kernel_stats.csv"%k)))
    return {r["Name"]:(int(r["Calls"]),float(r["AverageNs"])/1e3,float(r["TotalDurationNs"])/1e6) for r in rows}
a,b=load("none"),load("reducer")
print("total kernel ms: none %.2f  reducer %.2f"%(sum(v[2] for v in a.values()),sum(v[2] for v in b.values())))
names=sorted(set(a)|set(b),key=lambda n:-abs(a.get(n,(0,0,0))[2]-b.get(n,(0,0,0))[2]))
for n in names[:25]:
    x,y=a.get(n,(0,0,0)),b.get(n,(0,0,0))
    print("%-60s none %5d x %7.1f us = %7.2f ms | reducer %5d x %7.1f us = %7.2f ms"%(n.replace("(anonymous namespace)::","").replace("void ","")[:60],x[0],x[1],x[2],y[0],y[1],y[2]))
PY
