# Which kernels does the vendor library (hipBLASLt / rocBLAS through torch.matmul) run on the blocks' shapes and on 4096^3?
# Their names encode macro-tile, MFMA shape, wave layout, direct-to-LDS / direct-to-VGPR, prefetch depth (VERDICT r5 "Next" 2).
#   bash tools/lib_gemm_names.sh <dir>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 -c 'import __graft_entry__ as g; g.build(lab=False)' > /dev/null
out=gpurun_out/$1; mkdir -p $out
export GEMM_BENCH_TORCH=1 GEMM_BENCH_ONLY="square,fc2 fwd,fc1 dgrad,fc1 fwd,qkv fwd,fc1 wgrad"
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o lib -- python3 tools/gemm_bench.py 10 > $out/lib.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/lib_kernel_stats.csv")))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"])):
    n=r["Name"]
    if "Cijk" in n or "gemm" in n.lower():
        print("%6d x %8.1f us  %s"%(int(r["Calls"]), float(r["AverageNs"])/1e3, n[:400]))
PY
grep -v amdgpu $out/lib.log | tail -8
