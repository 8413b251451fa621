# rocprofv3 kernel-trace summary of bench.py at another batch size -> gpurun_out/<dir>/   (usage: bash tools/run_stats_b.sh <dir> <batch>)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 -c 'import __graft_entry__ as g; g.build(lab=False)' > /dev/null
out="gpurun_out/$1"; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python3 bench.py --batch $2 --steps 20 --warmup 3 --no-cpu-baseline > $out/stats.log 2>&1
f=$out/stats_kernel_trace.csv
[ -f $f ] && python3 tools/trace_seq.py $f > $out/seq.txt 2>&1 && rm -f $f
tail -2 $out/stats.log | cut -c1-200
