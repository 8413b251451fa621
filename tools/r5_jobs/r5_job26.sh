cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/j26; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o b4 -- python3 bench.py --batch 4 --steps 20 --warmup 3 --no-cpu-baseline > $out/b4.log 2>&1
python3 tools/trace_seq.py $out/b4_kernel_trace.csv > $out/seq_b4.txt 2>&1
rm -f $out/b4_kernel_trace.csv
tail -1 $out/b4.log | cut -c1-200
