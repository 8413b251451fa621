# copies what tools/r5_record.sh left under gpurun_out/r05 into profiles/r05_* (run here, after the gpurun call)
set -e
cd "$(dirname "$0")/.."
s=gpurun_out/r05; d=profiles
cp $s/bench.json $d/r05_bench.json
cp $s/bench_b4.json $d/r05_bench_b4.json
cp $s/bench_b8.json $d/r05_bench_b8.json
cp $s/bench_vitl_b8.json $d/r05_bench_vitl_b8.json
cp $s/bench_vith_b12.json $d/r05_bench_vith_b12.json
cp $s/stats_kernel_stats.csv $d/r05_bench_bs12_kernel_stats.csv
cp $s/pmc_traffic.json $d/r05_pmc_traffic.json
grep -v amdgpu.ids $s/trainstep.txt > $d/r05_trainstep.txt
cp $s/attn_bwd_scale.txt $d/r05_attn_bwd_scale.txt
cp $s/seq.txt $d/r05_step_launch_sequence.txt
cp $s/sq_counters.txt $d/r05_sq_counters.txt
cp $s/library_gemm.txt $d/r05_library_gemm.txt
cp $s/library_attention.txt $d/r05_library_attention.txt
cp $s/graph_branch_probe.txt $d/r05_graph_branch_probe.txt
cp $s/neck_lanes_ab.txt $d/r05_neck_lanes_ab.txt
cp $s/commit.txt $d/r05_commit.txt
ls -la $d | grep r05
python3 tools/roofline_table.py profiles/r05 > profiles/r05_roofline_table.txt 2>/dev/null || true
cp $s/nobrs.txt $d/r05_nobrs.txt 2>/dev/null || true
