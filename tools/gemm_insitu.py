#!/usr/bin/env python
"""Why is a block GEMM slower inside the training step than in tools/gemm_bench.py?  Two experiments per shape:
  sustained: the same launch repeated for ~1.5 s, time per launch reported per 100-launch window (clock / power drift);
  cold:      the launch cycles over enough distinct operand sets (> 600 MB) that nothing is left in L2 / MALL.
usage: python tools/gemm_insitu.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops  # noqa: E402

M = 9408
SHAPES = [
    ("qkv fwd", 0, 0, M, 2304, 768, ops.EPI_BIAS),
    ("fc1 fwd gelu", 0, 0, M, 3072, 768, ops.EPI_BIAS | ops.EPI_GELU | ops.EPI_SAVE_DGELU),
    ("fc2 fwd+res", 0, 0, M, 768, 3072, ops.EPI_BIAS | ops.EPI_RESID),
    ("fc2 dgrad*aux", 0, 1, M, 3072, 768, ops.EPI_MULAUX),
    ("fc1 dgrad", 0, 1, M, 768, 3072, 0),
]


def operands(tA, tB, m, n, k, flags, dev="cuda"):
    A = (torch.rand((k, m) if tA else (m, k), device=dev) - 0.5).to(torch.bfloat16)
    Bm = (torch.rand((k, n) if tB else (n, k), device=dev) - 0.5).to(torch.bfloat16)
    C = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
    kw = dict(transA=bool(tA), transB=bool(tB), flags=flags, bias=torch.rand(n, device=dev),
              resid=torch.zeros(m, n, device=dev, dtype=torch.bfloat16), ldr=n,
              aux=torch.ones(m, n, device=dev, dtype=torch.bfloat16), ldaux=n,
              preact=torch.zeros(m, n, device=dev, dtype=torch.bfloat16))
    return (A, Bm, C, m, n, k, (m if tA else k), (n if tB else k), n, 0), kw


def timed(calls, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        a, kw = calls[i % len(calls)]
        ops.gemm(*a, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    only = [t for t in os.environ.get("GEMM_BENCH_ONLY", "").split(",") if t]
    if os.environ.get("INSITU_AB"):        # warm / cold per kernel family (k2 option values), no sustained run
        for name, tA, tB, m, n, k, flags in SHAPES:
            one = [operands(tA, tB, m, n, k, flags)]
            many = [operands(tA, tB, m, n, k, flags) for _ in range(10)]
            txt = []
            for opt in os.environ["INSITU_AB"].split(","):
                ops.gemm_set_option("k2", int(opt))
                timed(one, 20); warm = sorted(timed(one, 100) for _ in range(5))[2]
                timed(many, 20); cold = sorted(timed(many, 100) for _ in range(5))[2]
                txt.append(f"k2={opt}: warm {warm:6.1f} cold {cold:6.1f}")
            print(f"{name:16s} " + "   ".join(txt), flush=True)
        return
    for name, tA, tB, m, n, k, flags in SHAPES:
        if only and not any(t in name for t in only):
            continue
        one = [operands(tA, tB, m, n, k, flags)]
        timed(one, 10)
        first = timed(one, 50)
        win = [timed(one, 100) for _ in range(int(os.environ.get("INSITU_WINDOWS", "200")))]
        nset = 10
        many = [operands(tA, tB, m, n, k, flags) for _ in range(nset)]
        timed(many, 20)
        cold = sorted(timed(many, 100) for _ in range(5))[2]
        warm_after = timed(one, 50)
        print(f"{name:16s} first {first:6.1f}  sustained: w0 {win[0]:6.1f} w{len(win)//2} {win[len(win)//2]:6.1f} last {win[-1]:6.1f}"
              f"  cold({nset} sets) {cold:6.1f}  warm again {warm_after:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
