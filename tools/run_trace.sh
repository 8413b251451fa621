set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/trace6
VPU_WGRAD_STREAM=0 VPU_GEMM_SHAPES=gpurun_out/gemm_shapes6.txt rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace6 -o ser -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/trace6_ser.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace6 -o par -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/trace6_par.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench8.log 2>&1
VPU_WGRAD_STREAM=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench8_ser.log 2>&1
ls -la gpurun_out/trace6
