# MFMA / LDS activity counters of the GEMM micro-benchmark (two passes) -> gpurun_out/<dir>   usage: bash tools/run_pmc_gemm.sh <dir>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null
out="gpurun_out/$1"; mkdir -p $out
GEMM_BENCH_GROUP=1 GEMM_BENCH_ONLY="qkv fwd,fc1 fwd,fc2 fwd,fc1 dgrad,fc2 dgrad" rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out -o g1 -- python3 tools/gemm_bench.py 5 > $out/g1.log 2>&1
GEMM_BENCH_GROUP=1 GEMM_BENCH_ONLY="qkv fwd,fc1 fwd,fc2 fwd,fc1 dgrad,fc2 dgrad" rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_WAVES --kernel-trace --output-format csv -d $out -o g2 -- python3 tools/gemm_bench.py 5 > $out/g2.log 2>&1
ls $out
