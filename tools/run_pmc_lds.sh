# LDS bank-conflict share per kernel of bench.py (usage: bash tools/run_pmc_lds.sh <dir>)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# (re)build the extension BEFORE the profiler is involved: its preload must not wrap 8 hipcc children
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null
out="gpurun_out/$1"; mkdir -p $out
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out -o lds -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/lds.log 2>&1
ls $out
