# VERDICT r5 "Next" 7: which launch mode should an N > 1 run default to?  The forced reducer at world size 1, eager against the
# graph chain, at B = 12 and B = 4, with the process pinned to 32 host threads (what one of 8 ranks gets on a 256-thread host).
out=gpurun_out/$1; mkdir -p $out
python3 -c 'import __graft_entry__ as g; g.build(lab=False)' > /dev/null
for b in 12 4; do
  DP_BATCH=$b timeout -k 10 280 taskset -c 0-31 python tools/dp_rehearsal.py 20 $out/dp_b${b}_32thr.json > $out/dp_b${b}_32thr.log 2>&1
  DP_BATCH=$b timeout -k 10 280 python tools/dp_rehearsal.py 20 $out/dp_b${b}_all.json > $out/dp_b${b}_all.log 2>&1
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/dp_b*.json")):
    r=json.load(open(f))
    print(f.split('/')[-1], {k:r[k] for k in ("host_threads_allowed","ms_no_reducer","ms_reducer_fp32_wire_reserve16","eager_host_ms_per_step_reserve16","ms_reducer_fp32_wire_reserve16_graph_chain","graph_chain_host_ms_per_step","graph_chain_backward_segments")})
PY
