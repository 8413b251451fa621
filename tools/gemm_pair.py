#!/usr/bin/env python
"""Does a GEMM run slower right behind the GEMM that wrote its operand (fc1 -> fc2, as in the step) than repeated alone
(tools/gemm_bench.py)?  Times fc1 alone, fc2 alone and the alternating pair on the real data flow (fc1's output is fc2's A operand);
the pair minus the sum is what the step pays per block for the hand-over.  GEMM_PAIR_SPREAD=n: the pair cycles over n sets of
activations (n x 190 MB), as consecutive blocks of the model do.
usage: python tools/gemm_pair.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops  # noqa: E402

M, D, H = 9408, 768, 3072


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    nset = int(os.environ.get("GEMM_PAIR_SPREAD", "1"))
    dev = "cuda"
    r = lambda *s: (torch.rand(*s, device=dev) - 0.5).to(torch.bfloat16)
    W1, W2 = r(H, D) * 0.05, r(D, H) * 0.05
    b1, b2 = torch.rand(H, device=dev), torch.rand(D, device=dev)
    sets = [dict(x=r(M, D), h=torch.zeros(M, H, device=dev, dtype=torch.bfloat16), gp=torch.zeros(M, H, device=dev, dtype=torch.bfloat16),
                 y=torch.zeros(M, D, device=dev, dtype=torch.bfloat16)) for _ in range(nset)]

    def fc1(s):
        ops.gemm(s["x"], W1, s["h"], M, H, D, D, D, H, 0, flags=ops.EPI_BIAS | ops.EPI_GELU | ops.EPI_SAVE_DGELU, bias=b1, preact=s["gp"])

    def fc2(s):
        ops.gemm(s["h"], W2, s["y"], M, D, H, H, H, D, 0, flags=ops.EPI_BIAS | ops.EPI_RESID, bias=b2, resid=s["x"], ldr=D)

    def timed(fn):
        for i in range(10):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    for rnd in range(3):
        t1 = timed(lambda i: fc1(sets[i % nset]))
        t2 = timed(lambda i: fc2(sets[i % nset]))
        tp = timed(lambda i: (fc1(sets[i % nset]), fc2(sets[i % nset])))
        print(f"round {rnd}: fc1 alone {t1:6.1f} us   fc2 alone {t2:6.1f} us   pair {tp:6.1f} us   pair - sum {tp - t1 - t2:+6.1f} us   ({nset} set(s))", flush=True)


if __name__ == "__main__":
    main()
