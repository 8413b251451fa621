#!/usr/bin/env python
"""K3 (two 256-thread workgroups per CU, free-running) against K2 (one 512-thread workgroup, ping-pong K halves) on grouped
weight-gradient launches of equal work, interleaved in one process (median of 5 rounds).
usage: python tools/k3_bench.py [reps]"""
import os
import sys

os.environ.setdefault("VPU_LIB_DIAG", "1")       # the K3 / ring families live in the laboratory library (bash pvpuformer_amd/csrc/build.sh diag)

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops  # noqa: E402

R = 9408


def problems(shapes, red=R):
    probs, fl = [], 0.0
    for m, n in shapes:
        A = (torch.rand(red, m, device="cuda") - 0.5).to(torch.bfloat16)
        B = (torch.rand(red, n, device="cuda") - 0.5).to(torch.bfloat16)
        C = torch.zeros(m, n, device="cuda")
        cs = torch.zeros(m, device="cuda")
        probs.append(((A, B, C, m, n, red, m, n, n, 0), dict(transA=True, transB=True, flags=ops.EPI_OUT_F32 | ops.EPI_ACCUM, colsum=cs)))
        fl += 2.0 * m * n * red
    return probs, fl


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    blk = [(3072, 768), (768, 3072), (2304, 768), (768, 768)]
    cases = [("1 block (216 tiles)", blk), ("2 blocks (432 tiles)", blk + blk), ("4096x4096 (512 tiles)", [(4096, 4096)]),
             ("243-tile pack", [(3072, 768), (768, 3072), (2304, 768), (768, 768), (768, 384), (768, 384), (512, 384)]),
             ("486-tile pack", blk + blk + [(768, 384), (768, 384), (512, 384)] * 2)]
    if os.environ.get("K3_BENCH_CASES") == "pack":      # (counter passes: one case, every kernel family)
        cases = [c for c in cases if c[0].startswith("486")]
    nsets = int(os.environ.get("K3_BENCH_SETS", "1"))      # > 1: cycle over that many operand sets (cold operands, as in the step)
    for name, shapes in cases:
        sets = [problems(shapes) for _ in range(nsets)]
        fl = sets[0][1]
        res = {}
        for rnd in range(5):
            for k3 in (0, 1, 8, 24):
                ops.gemm_set_option("k3", k3)
                ops.gemm_grouped(sets[0][0])
                kern = ops.gemm_last_kernel()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for r in range(reps):
                    ops.gemm_grouped(sets[r % nsets][0])
                e1.record()
                torch.cuda.synchronize()
                res.setdefault((k3, kern), []).append(e0.elapsed_time(e1) * 1e-3 / reps)
        ops.gemm_set_option("k3", -1)
        nm = {0: "K2", 1: "K3", 8: "K4", 24: "K4P"}
        txt = "   ".join(f"{nm[k[0]]}({k[1][10:13]}): {sorted(v)[2] * 1e6:7.1f} us {fl / sorted(v)[2] / 1e12:6.0f} TF" for k, v in res.items())
        print(f"{name:24s} {txt}", flush=True)


if __name__ == "__main__":
    main()
