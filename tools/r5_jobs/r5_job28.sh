cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/j28; mkdir -p $out
for cfg in "vitl 8" "vith 12"; do set -- $cfg
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $1 -- python3 bench.py --model $1 --batch $2 --steps 6 --warmup 2 --no-cpu-baseline > $out/$1.log 2>&1
rm -f $out/$1_kernel_trace.csv
tail -1 $out/$1.log | cut -c1-160
done
