cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for i in 1 2; do GEMM_BENCH_ONLY="proj" python3 tools/gemm_bench.py 50 2>&1 | grep -v amdgpu.ids; done
VPU_LIB_DIAG=1 GEMM_BENCH_ONLY="proj" python3 tools/gemm_bench.py 50 2>&1 | grep -v amdgpu.ids
python3 - <<'P'
import torch, sys
sys.path.insert(0, '.')
from pvpuformer_amd import ops
M,N,K=9408,768,768
A=(torch.rand(M,K,device='cuda')-0.5).to(torch.bfloat16); B=(torch.rand(K,N,device='cuda')-0.5).to(torch.bfloat16); C=torch.zeros(M,N,device='cuda',dtype=torch.bfloat16)
for rep in range(3):
    ops.gemm(A,B,C,M,N,K,K,N,N,0,transB=True,flags=0)
    print(ops.gemm_last_kernel())
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.gemm(A,B,C,M,N,K,K,N,N,0,transB=True,flags=0)
    e1.record(); torch.cuda.synchronize(); print(e0.elapsed_time(e1)/50*1e3,'us')
P
