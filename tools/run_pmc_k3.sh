# MFMA / LDS activity counters of the grouped weight-gradient kernels K2 / K3 / K4 / K4P (tools/k3_bench.py, two passes)
# -> gpurun_out/<dir>; then python tools/pmc_sq_summary.py gpurun_out/<dir>    usage: bash tools/run_pmc_k3.sh <dir>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null
out="gpurun_out/$1"; mkdir -p $out
K3_BENCH_CASES=pack rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out -o g1 -- python3 tools/k3_bench.py 3 > $out/g1.log 2>&1
K3_BENCH_CASES=pack rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $out -o g2 -- python3 tools/k3_bench.py 3 > $out/g2.log 2>&1
ls $out
