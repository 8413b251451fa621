# rocprofv3 kernel-trace summary of bench.py in another configuration -> gpurun_out/<dir>/   (usage: bash tools/run_stats_b.sh <dir> "<bench args>", e.g. "--batch 4" or "--model vitl --batch 8 --steps 6 --warmup 2")
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 -c 'import __graft_entry__ as g; g.build(lab=False)' > /dev/null
out="gpurun_out/$1"; mkdir -p $out
case "$2" in *--steps*) A="$2";; *) A="$2 --steps 20 --warmup 3";; esac
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python3 bench.py $A --no-cpu-baseline > $out/stats.log 2>&1
rm -f $out/stats_kernel_trace.csv
grep -c . $out/stats_kernel_stats.csv
