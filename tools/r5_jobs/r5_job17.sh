cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j17
timeout -k 10 300 python3 -m pytest tests/test_ops_gpu.py -x -q -k "gate" > gpurun_out/j17/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/j17/pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/j17 -o st -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/j17/st.log 2>&1
grep "gate_bwd\|head_grad\|upsum\|bilinear" gpurun_out/j17/st_kernel_stats.csv | cut -c1-60,100-200
rm -f gpurun_out/j17/st_kernel_trace.csv
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c60-175
