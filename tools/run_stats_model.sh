# rocprofv3 kernel-trace summary of bench.py for another backbone -> gpurun_out/<dir>   usage: bash tools/run_stats_model.sh <model> <batch> <dir>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null
out="gpurun_out/$3"; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python3 bench.py --model $1 --batch $2 --steps 2 --warmup 1 --no-cpu-baseline > $out/stats.log 2>&1
ls $out
