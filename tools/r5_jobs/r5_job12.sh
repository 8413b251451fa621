# per-kernel times at batch 4 against batch 12 (which kernels do not scale with the batch)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/j12; mkdir -p $out
for b in 4 12; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o b$b -- python3 bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline > $out/b$b.log 2>&1
  python3 tools/trace_seq.py $out/b${b}_kernel_trace.csv > $out/seq_b$b.txt 2>&1
  rm -f $out/b${b}_kernel_trace.csv
  tail -1 $out/b$b.log | cut -c1-200
done
for b in 4 8 12; do python3 bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-175; done
ls $out
