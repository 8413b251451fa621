# Records everything profiles/rNN_* is made from, on ONE box, at the current commit.
#   on the GPU box:   bash tools/record_round.sh 06            (outputs -> gpurun_out/r06)
#   afterwards here:  bash tools/record_round.sh 06 copy       (gpurun_out/r06 -> profiles/r06_*)
R=${1:?round number, e.g. 06}
if [ "$2" = "copy" ]; then
  set -e
  cd "$(dirname "$0")/.."
  s=gpurun_out/r$R; d=profiles
  for f in bench bench_b4 bench_b8 bench_vitl_b8 bench_vith_b12 pmc_traffic; do [ -f $s/$f.json ] && cp $s/$f.json $d/r${R}_$f.json; done
  cp $s/stats_kernel_stats.csv $d/r${R}_bench_bs12_kernel_stats.csv
  for f in trainstep attn_bwd_scale sq_counters_k5 library_gemm library_attention nobrs k5_ab dp_mode dp_lag commit gemm_vitl gemm_vith; do
    [ -f $s/$f.txt ] && grep -v amdgpu.ids $s/$f.txt > $d/r${R}_$f.txt
  done
  [ -f $s/seq.txt ] && cp $s/seq.txt $d/r${R}_step_launch_sequence.txt
  python3 tools/roofline_table.py profiles/r$R > profiles/r${R}_roofline_table.txt 2>/dev/null || true
  ls -la $d | grep r$R
  exit 0
fi
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 -c 'import __graft_entry__ as g; g.build(lab=False)' > /dev/null
out=gpurun_out/r$R; mkdir -p $out
echo "commit: $(cat .git_rev 2>/dev/null)" > $out/commit.txt
# 1. the default bench line (CPU baseline beside it), then the other batch sizes / backbones
python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc $?"; cut -c1-260 $out/bench.json
python3 bench.py --batch 4 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_b4.json 2>/dev/null; cut -c1-160 $out/bench_b4.json
python3 bench.py --batch 8 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_b8.json 2>/dev/null; cut -c1-160 $out/bench_b8.json
python3 bench.py --model vitl --batch 8 --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_vitl_b8.json 2>/dev/null; cut -c1-160 $out/bench_vitl_b8.json
python3 bench.py --model vith --batch 12 --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_vith_b12.json 2>/dev/null; cut -c1-160 $out/bench_vith_b12.json
# 2. kernel statistics of the same command (host-enqueued under the profiler; 20 timed steps so that construction's copies amortise)
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/stats.log 2>&1
f=$(ls $out/stats_kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python3 tools/trace_seq.py $f > $out/seq.txt 2>&1; wc -l $out/seq.txt; rm -f $f; fi
# 3. HBM traffic per kernel (two PMC passes, nothing else traced)
VPU_WGRAD_STREAM=0 VPU_BENCH_GRAPH=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out -o fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/fetch.log 2>&1
VPU_WGRAD_STREAM=0 VPU_BENCH_GRAPH=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out -o write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/write.log 2>&1
python3 tools/pmc_traffic.py $out $out/pmc_traffic.json > $out/pmc_traffic.txt 2>&1; head -6 $out/pmc_traffic.txt
rm -f $out/fetch_kernel_trace.csv $out/write_kernel_trace.csv $out/fetch_counter_collection.csv $out/write_counter_collection.csv
# 4. the reference-faithful 1-3-iteration training step, the NoBRS loop
python3 tools/bench_trainstep.py 40 12 > $out/trainstep.txt 2>&1; BENCH_PROMPTS=0,1,2 python3 tools/bench_trainstep.py 40 12 >> $out/trainstep.txt 2>&1
python3 tools/bench_nobrs.py > $out/nobrs.txt 2>&1
# 5. the GEMM families side by side on the blocks' shapes (K2 default rule | K5 | K5 main loops only), ViT-B / -L / -H rows
GEMM_BENCH_K2=2,k5a,k5x python3 tools/gemm_bench.py 20 > $out/k5_ab.txt 2>&1
GEMM_BENCH_M=6272 GEMM_BENCH_D=1024 GEMM_BENCH_K2=2,k5a python3 tools/gemm_bench.py 20 > $out/gemm_vitl.txt 2>&1
GEMM_BENCH_M=12288 GEMM_BENCH_D=1280 GEMM_BENCH_K2=2,k5a python3 tools/gemm_bench.py 20 > $out/gemm_vith.txt 2>&1
# 6. calibration against the libraries of this image (not product paths)
GEMM_BENCH_TORCH=1 python3 tools/gemm_bench.py 50 > $out/library_gemm.txt 2>&1
python3 tools/sdpa_compare.py > $out/library_attention.txt 2>&1
# 7. SQ counters of the K5 forms and of fc1's K2 form (three passes of eight counters)
o=$out/sq; mkdir -p $o
for p in 1 2 3; do
  case $p in
    1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT";;
    2) C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES";;
    3) C="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL";;
  esac
  GEMM_BENCH_ONLY="qkv fwd,fc1 fwd,fc2 dgrad,fc2 fwd" rocprofv3 --pmc $C --kernel-trace --output-format csv -d $o -o p$p -- python3 tools/gemm_bench.py 5 > $o/p$p.log 2>&1
done
python3 tools/pmc_sq_summary.py $o gemm_bf16_k > $out/sq_counters_k5.txt 2>&1; rm -rf $o
# 8. the data-parallel launch mode and report lag on one GPU (forced reducer, world size 1, 32 host threads)
bash tools/dp_mode_job.sh r$R/dp > $out/dp_mode.txt 2>&1
bash tools/dp_lag_job.sh r$R/lag > $out/dp_lag.txt 2>&1
ls $out
