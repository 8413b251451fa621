set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests -q -m gpu > gpurun_out/full_test.log 2>&1; echo "test rc $?" >> gpurun_out/full_test.log
tail -15 gpurun_out/full_test.log
