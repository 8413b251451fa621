"""bench.py's multi-rank plumbing without a GPU: ``python bench.py --gpus 2`` started with plain python (no WORLD_SIZE in
the environment) must start the two ranks itself -- fresh child processes, gloo on the CPU under ``--rehearse-launch`` --
and print ONE JSON line with n_gpus = 2 and the data-parallel diagnostics; a launcher / --gpus mismatch must fail loudly
instead of silently measuring one rank (the reference's launch: train.py:18,74, isegm/utils/distributed.py:25-67)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_bench_starts_its_own_ranks_and_reports_dp_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--rehearse-launch"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] is None and d["config"]["parallelism"] == "dp2" and d["config"]["reduced_ok"]
    dp = d["dp"]
    assert dp["world_size"] == 2 and dp["ranks_seen_by_all_reduce"] == 2 and dp["backend"] == "gloo"
    assert dp["collectives_per_step"] >= 2 and dp["payload_bytes_per_step"] == 4 << 20
    assert dp["wire_bytes_per_gpu_per_step"] == 4 << 20          # ring all-reduce: 2 (N - 1) / N x payload
    assert dp["exposed_comm_ms"] is not None and dp["exposed_comm_ms"] >= 0
    assert len(dp["buckets"]) == dp["collectives_per_step"]
    for b in dp["buckets"]:
        assert b["done_ms_vs_bwd_end"] is not None and b["launched_ms_vs_bwd_end"] <= 0.5 and b["host_wait_ms"] >= 0
    for k in ("launch_mode", "NCCL_MAX_NCHANNELS", "VPU_DIST_RESERVE_CUS", "split_adam", "reserve_cus", "bucket_mb"):
        assert k in dp


def test_bench_refuses_a_launcher_mismatch():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-launch"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_finish_keep_last_checks_the_order():
    """GradReducer.finish(keep_last): [reduced_from, total) is only called final when the kept collectives lie wholly below
    every waited one; out-of-order ranges make it wait for everything."""
    import torch
    from pvpuformer_amd.parallel import GradReducer

    class W:
        def __init__(self): self.waited = False
        def wait(self): self.waited = True

    g = torch.zeros(100)
    red = GradReducer(g)
    red.enabled, red.world = True, 2
    # tail-first: waits for the first, keeps the last two
    red._works, red.launched = [W(), W(), W()], [(60, 100), (30, 60), (0, 30)]
    red.finish(keep_last=2)
    assert red.reduced_from == 60 and len(red._works) == 2
    # out of order: the kept range (70, 100) lies above the waited one -> everything is waited for
    ws = [W(), W(), W()]
    red._works, red.launched = list(ws), [(0, 30), (30, 70), (70, 100)]
    red.finish(keep_last=2)
    assert red.reduced_from == 0 and not red._works and all(w.waited for w in ws)
