# SQ activity counters of bench.py's kernels (one rocprofv3 --pmc pass) -> gpurun_out/<dir>   usage: bash tools/run_pmc_sq_bench.sh <dir>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null
out="gpurun_out/$1"; mkdir -p $out
VPU_BENCH_GRAPH=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace --output-format csv -d $out -o sq -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/sq.log 2>&1
ls $out
