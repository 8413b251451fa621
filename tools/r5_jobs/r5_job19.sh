cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j19
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/j19/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/j19/pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c60-175
