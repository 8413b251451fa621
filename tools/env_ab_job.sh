# runtime environment knobs against the default, interleaved on one box: bash tools/env_ab_job.sh <outdir> "<ENV1>" "<ENV2>" ...
out="gpurun_out/$1"; shift; mkdir -p $out
python3 -c 'import __graft_entry__ as g; g.build(lab=False)' > /dev/null
for r in 1 2; do
  i=0
  for e in "VPU_NONE=1" "$@"; do
    env $e timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/e${i}_$r.json 2> $out/e${i}_$r.err
    python3 - <<PY
import json
try:
    x=json.loads(open("$out/e${i}_$r.json").read().strip().splitlines()[-1]); print("round $r", "$e", x["value"], x["ms_per_step"])
except Exception as ex: print("round $r", "$e", "failed", ex)
PY
    i=$((i+1))
  done
done
