# usage: bash tools/run_pmc_l2.sh <shape-substring> <outdir>   L2 hit/miss and fabric bytes of one GEMM shape
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# (re)build the extension BEFORE the profiler is involved: its preload must not wrap 8 hipcc children
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null
shape="$1"; out="gpurun_out/$2"; mkdir -p $out
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d $out -o l1 -- python3 tools/gemm_one.py "$shape" 3 > $out/l1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out -o l2 -- python3 tools/gemm_one.py "$shape" 3 > $out/l2.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $out -o l3 -- python3 tools/gemm_one.py "$shape" 3 > $out/l3.log 2>&1
