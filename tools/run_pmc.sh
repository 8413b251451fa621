# usage: bash tools/run_pmc.sh <shape-substring> <outdir>   (one rocprofv3 --pmc pass per counter group)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# (re)build the extension BEFORE the profiler is involved: its preload must not wrap 8 hipcc children
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null
shape="$1"; out="gpurun_out/$2"; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $out -o p1 -- python3 tools/gemm_one.py "$shape" 3 > $out/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES --kernel-trace --output-format csv -d $out -o p2 -- python3 tools/gemm_one.py "$shape" 3 > $out/p2.log 2>&1
ls $out
