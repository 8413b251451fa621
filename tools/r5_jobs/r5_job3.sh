# round-5 job 3: SQ counters of the window backward (modes 2 and 1), remaining tests
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j3
for m in 2 1; do
  export VPU_ATTN_ONEPASS=$m
  out=gpurun_out/j3/m$m; mkdir -p $out
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $out -o p1 -- python3 tools/op_bench.py attn_bwd_window > $out/p1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES --kernel-trace --output-format csv -d $out -o p2 -- python3 tools/op_bench.py attn_bwd_window > $out/p2.log 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --kernel-trace --output-format csv -d $out -o p3 -- python3 tools/op_bench.py attn_bwd_window > $out/p3.log 2>&1
  echo "== mode $m"; python3 tools/pmc_sq_summary.py $out attn_bwd
  rm -f $out/*kernel_trace.csv $out/*agent_info.csv
done 2>&1 | tee gpurun_out/j3/sq.txt
unset VPU_ATTN_ONEPASS
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py -x -q -k "one_pass or attention or flash or thick or pue or disk" > gpurun_out/j3/pytest_ops.log 2>&1; echo "pytest ops rc $?"
tail -3 gpurun_out/j3/pytest_ops.log
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py -x -q -k "lazy_zero or backbone_forward or scribble or tiny or public_coord" > gpurun_out/j3/pytest_model.log 2>&1; echo "pytest model rc $?"
tail -3 gpurun_out/j3/pytest_model.log
