out=gpurun_out/$1; mkdir -p $out
python3 -c 'import __graft_entry__ as g; g.build(lab=False)' > /dev/null
for lag in 1 2 3; do
  VPU_DIST_REPORT_LAG=$lag DP_BATCH=12 timeout -k 10 280 python tools/dp_rehearsal.py 20 $out/dp_lag$lag.json > $out/dp_lag$lag.log 2>&1
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/dp_lag*.json")):
    r=json.load(open(f))
    print(f.split('/')[-1], {k:r[k] for k in ("ms_no_reducer","ms_reducer_fp32_wire_reserve16","ms_reducer_fp32_wire_reserve0","ms_reducer_fp32_wire_reserve16_graph_chain","ms_reducer_fp32_wire_reserve0_graph_chain","collectives_per_step","graph_chain_backward_segments")})
PY
