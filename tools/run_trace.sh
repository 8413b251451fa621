# kernel timeline of bench.py (serialized: weight-gradient side stream off) -> gpurun_out/<dir>
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# (re)build the extension BEFORE the profiler is involved: its preload must not wrap 8 hipcc children
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null
out=gpurun_out/${1:-trace}
mkdir -p $out
VPU_WGRAD_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $out -o ser -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/ser.log 2>&1
ls -la $out
