# same-box A/B of the whole step: bash tools/ab_bench.sh <outdir> "<ENV_A>" "<ENV_B>" [rounds] [extra bench args]
# (the pool's boxes differ by several per cent on the same code: only interleaved runs on ONE box compare)
out="gpurun_out/$1"; mkdir -p $out
A="$2"; B="$3"; R=${4:-2}; X="$5"
python3 -c 'import __graft_entry__ as g; g.build(lab=False)' > /dev/null
for r in $(seq 1 $R); do
  env $A timeout -k 10 280 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $X > $out/A$r.json 2> $out/A$r.err
  env $B timeout -k 10 280 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $X > $out/B$r.json 2> $out/B$r.err
done
python3 - <<PY
import json,glob
for k in "AB":
    v=[json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob("$out/%s*.json"%k)) if open(f).read().strip()]
    print(k, [round(x["value"],1) for x in v], [round(x["ms_per_step"],3) for x in v])
PY
