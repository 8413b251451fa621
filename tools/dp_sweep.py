#!/usr/bin/env python
"""One-shot sweep for the FIRST multi-GPU node this build sees: the two knobs of the data-parallel step that could only be
guessed on one GPU -- the CUs the persistent GEMM grids leave to RCCL (VPU_DIST_RESERVE_CUS) and RCCL's channel cap
(NCCL_MAX_NCHANNELS) -- plus the wire format, the split optimizer step, the eager loop against the graph chain and the
report lag of the gradient ranges.  Every cell is one
``python bench.py --gpus N`` (bench.py starts its own ranks); the table shows images/s, exposed communication and the
single-GPU-relative efficiency.

    python tools/dp_sweep.py --gpus 8 [--steps 20] [--quick]

Never run by the tests (it needs N GPUs)."""
import argparse
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n, steps, env_over):
    env = dict(os.environ)
    env.update({k: str(v) for k, v in env_over.items() if v is not None})
    for k, v in env_over.items():
        if v is None:
            env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(steps), "--warmup", "5",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=1800)
    line = next((l for l in r.stdout.splitlines() if l.startswith("{")), None)
    if r.returncode != 0 or line is None:
        return {"error": (r.stderr or r.stdout)[-300:]}
    return json.loads(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, required=True)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--quick", action="store_true", help="reserve x channels only (skip wire / split-Adam / chain)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "dp_sweep.json"))
    a = ap.parse_args()
    base = run(1, a.steps, {})
    one = base.get("value")
    print(f"1 GPU: {one} images/s")
    rows = []
    grid = list(itertools.product((0, 8, 16, 32), (None, 8, 16, 32)))       # (reserved CUs, channel cap; None = RCCL's default)
    # (round 6: the graph chain is bench.py's default at N > 1 and a finished gradient range may wait two blocks for its weight-
    # gradient launches to fill, VPU_DIST_REPORT_LAG = 2: both decided on one GPU -- the eager loop and the other lags are cells here)
    extra = [] if a.quick else [({"VPU_DIST_WIRE": "bf16"}, "bf16 wire"), ({"VPU_DIST_SPLIT_ADAM": 2}, "split Adam"),
                                ({"VPU_BENCH_DP_GRAPH": 0}, "eager loop"), ({"VPU_DIST_REPORT_LAG": 1}, "report lag 1"),
                                ({"VPU_DIST_REPORT_LAG": 3}, "report lag 3"),
                                ({"VPU_DIST_REPORT_LAG": 3, "VPU_DIST_WIRE": "bf16"}, "report lag 3 + bf16 wire")]
    for res, ch in grid:
        # (no cap given: bench.py's configure_rccl_env() caps the channels at the reserve; reserve 0 leaves RCCL's default)
        env = {"VPU_DIST_RESERVE_CUS": res, "NCCL_MAX_NCHANNELS": ch}
        d = run(a.gpus, a.steps, env)
        rows.append((f"reserve {res:2d} channels {ch if ch is not None else ('=reserve' if res else 'default')}", d))
        dp = d.get("dp", {})
        print(f"{rows[-1][0]:36s} {d.get('value')} images/s  eff {d.get('value', 0) / (a.gpus * one) if one and d.get('value') else None}"
              f"  exposed {dp.get('exposed_comm_ms')} ms  {d.get('error', '')}", flush=True)
    for env, name in extra:
        d = run(a.gpus, a.steps, env)
        rows.append((name, d))
        print(f"{name:36s} {d.get('value')} images/s  exposed {d.get('dp', {}).get('exposed_comm_ms')} ms  {d.get('error', '')}", flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump({"one_gpu": base, "rows": rows}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
