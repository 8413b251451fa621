cd $GRAFT_REPO_ROOT
for f in 0 16 4 20 0 16; do VPU_GEMM_K3_FORMS=$f timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('forms=$f', d['value'], d['ms_per_step'])"; done
