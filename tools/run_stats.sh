# rocprofv3 kernel-trace summary of bench.py -> gpurun_out/<dir>/stats_kernel_stats.csv   (usage: bash tools/run_stats.sh <dir>)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# (re)build the extension BEFORE the profiler is involved: its preload must not wrap 8 hipcc children
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null
out="gpurun_out/$1"; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/stats.log 2>&1
ls $out
