# round-5 recording: everything profiles/r05_* is made from, one box (usage: bash tools/r5_record.sh; outputs -> gpurun_out/r05)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null
out=gpurun_out/r05; mkdir -p $out
git_rev=$(cat .git_rev 2>/dev/null); echo "commit: $git_rev" > $out/commit.txt
# 1. the default bench line (CPU baseline beside it), then the other batch sizes / backbones
python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc $?"; cut -c1-260 $out/bench.json
python3 bench.py --batch 4 --no-cpu-baseline > $out/bench_b4.json 2>/dev/null; cut -c1-200 $out/bench_b4.json
python3 bench.py --model vitl --batch 8 --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_vitl_b8.json 2>/dev/null; cut -c1-200 $out/bench_vitl_b8.json
python3 bench.py --model vith --batch 12 --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_vith_b12.json 2>/dev/null; cut -c1-200 $out/bench_vith_b12.json
# 2. kernel statistics of the same command (host-enqueued under the profiler; 20 timed steps so that model construction's copies amortise)
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/stats.log 2>&1
f=$(ls $out/stats_kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python3 tools/trace_seq.py $f > $out/seq.txt 2>&1; python3 tools/trace_by_grid.py $f > $out/by_grid.txt 2>&1; wc -l $out/seq.txt; rm -f $f; fi
# 3. HBM traffic per kernel (two PMC passes)
VPU_WGRAD_STREAM=0 VPU_BENCH_GRAPH=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out -o fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/fetch.log 2>&1
VPU_WGRAD_STREAM=0 VPU_BENCH_GRAPH=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out -o write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/write.log 2>&1
python3 tools/pmc_traffic.py $out $out/pmc_traffic.json > $out/pmc_traffic.txt 2>&1; head -8 $out/pmc_traffic.txt
rm -f $out/fetch_kernel_trace.csv $out/write_kernel_trace.csv $out/fetch_counter_collection.csv $out/write_counter_collection.csv
# 4. the reference-faithful 1-3-iteration training step
python3 tools/bench_trainstep.py 40 12 > $out/trainstep.txt 2>&1; BENCH_PROMPTS=0,1,2 python3 tools/bench_trainstep.py 40 12 >> $out/trainstep.txt 2>&1; grep -v amdgpu.ids $out/trainstep.txt
# 5. window attention backward: the three forms against the problem count
for m in 1 2 0; do echo "== VPU_ATTN_ONEPASS=$m"; VPU_ATTN_ONEPASS=$m python3 tools/attn_bwd_scale.py; done 2>&1 | grep -v amdgpu.ids > $out/attn_bwd_scale.txt
ls $out
# 6. SQ counters of the window attention backward, the default form (onepass 1, conflict-free dS image) and the pass form (2)
for m in 1 2; do
  export VPU_ATTN_ONEPASS=$m
  o=$out/sq_m$m; mkdir -p $o
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $o -o p1 -- python3 tools/op_bench.py attn_bwd_window > $o/p1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES --kernel-trace --output-format csv -d $o -o p2 -- python3 tools/op_bench.py attn_bwd_window > $o/p2.log 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --kernel-trace --output-format csv -d $o -o p3 -- python3 tools/op_bench.py attn_bwd_window > $o/p3.log 2>&1
  echo "== VPU_ATTN_ONEPASS=$m"; python3 tools/pmc_sq_summary.py $o attn_bwd
  rm -rf $o
done > $out/sq_counters.txt 2>&1
unset VPU_ATTN_ONEPASS
# 7. calibration against the libraries this image carries (not product paths): hipBLASLt through torch.matmul on the ViT-B GEMM
#    shapes, torch's scaled_dot_product_attention on the attention shapes; hipGraph branch / crossing costs
GEMM_BENCH_TORCH=1 python3 tools/gemm_bench.py 50 2>&1 | grep -v amdgpu.ids > $out/library_gemm.txt
python3 tools/sdpa_compare.py 2>&1 | grep -v amdgpu.ids > $out/library_attention.txt
python3 tools/op_bench.py 2>/dev/null | grep -i "attn" >> $out/library_attention.txt
python3 tools/graph_branch_probe.py 2>&1 | grep -v amdgpu.ids > $out/graph_branch_probe.txt
# 8. batch 8 and the two-lane neck (opt-in) beside the default
python3 bench.py --batch 8 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_b8.json 2>/dev/null; cut -c1-200 $out/bench_b8.json
for m in 0 1; do echo "VPU_NECK_LANES=$m"; VPU_NECK_LANES=$m python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-200; done > $out/neck_lanes_ab.txt
ls $out
    python3 tools/bench_nobrs.py 2>&1 | grep -v amdgpu.ids > $out/nobrs.txt
