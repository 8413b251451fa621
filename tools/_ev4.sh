cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err
python3 -c "
import json
d=json.loads(open('gpurun_out/r04_bench_final.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['frac'], r.get('traffic'), d['cpu_baseline']['value'], d['parity']['bf16_rel'])
"
timeout -k 10 400 python3 bench.py --model vitl --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04_bench_vitl_b8.json 2>/dev/null
timeout -k 10 400 python3 bench.py --model vith --batch 12 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r04_bench_vith_b12.json 2>/dev/null
python3 -c "
import json
for n in ('vitl_b8','vith_b12'):
    d=json.load(open(f'gpurun_out/r04_bench_{n}.json')); r=d['roofline']
    print(n, d['value'], d['ms_per_step'], d['config']['mfma_roofline_frac_end_to_end'], r['kernel'], r['frac'])
"
