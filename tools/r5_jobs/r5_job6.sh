cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j6
for m in 1 2 0; do echo "== VPU_ATTN_ONEPASS=$m"; VPU_ATTN_ONEPASS=$m timeout -k 10 120 python3 tools/attn_bwd_scale.py; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/j6/scale.txt
