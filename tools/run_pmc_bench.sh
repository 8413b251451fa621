# HBM traffic of bench.py's kernels: two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass) -> gpurun_out/<dir>
# usage: bash tools/run_pmc_bench.sh <outdir>;  then  python tools/pmc_traffic.py gpurun_out/<outdir>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# (re)build the extension BEFORE the profiler is involved: its preload must not wrap 8 hipcc children
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null
out="gpurun_out/$1"; mkdir -p $out
VPU_WGRAD_STREAM=0 VPU_BENCH_GRAPH=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out -o fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/fetch.log 2>&1
VPU_WGRAD_STREAM=0 VPU_BENCH_GRAPH=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out -o write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/write.log 2>&1
ls $out
