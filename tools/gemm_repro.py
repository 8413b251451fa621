#!/usr/bin/env python
"""Launch-to-launch reproducibility of the GEMM kernels on the step's shapes: every shape N times on the same operands (a
busy second stream beside it for half of the launches), outputs compared bit for bit with the first -- a hand-counted
s_waitcnt or a raw s_barrier that is off by one shows up here as a difference.  usage: python tools/gemm_repro.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops                       # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import SHAPES, NECK, M             # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = "cuda"
    side = torch.cuda.Stream()
    noise_a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    bad = 0
    for name, tA, tB, m, n, k, flags in SHAPES + NECK:
        g = torch.Generator(device=dev).manual_seed(hash(name) & 0xFFFF)
        A = (torch.rand((k, m) if tA else (m, k), device=dev, generator=g) - 0.5).to(torch.bfloat16)
        Bm = (torch.rand((k, n) if tB else (n, k), device=dev, generator=g) - 0.5).to(torch.bfloat16)
        out_f32 = bool(flags & ops.EPI_OUT_F32)
        bias = torch.rand(n, device=dev, generator=g)
        R = torch.rand(m, n, device=dev, generator=g).to(torch.bfloat16)
        aux = torch.rand(m, n, device=dev, generator=g).to(torch.bfloat16)
        ref = None
        for r in range(reps):
            C = torch.full((m, n), 0.25, device=dev, dtype=torch.float32 if out_f32 else torch.bfloat16)
            pre = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
            if r % 2:
                with torch.cuda.stream(side):
                    torch.mm(noise_a, noise_a)        # something else on the chip
            ops.gemm(A, Bm, C, m, n, k, (m if tA else k), (n if tB else k), n, 0, transA=bool(tA), transB=bool(tB), flags=flags,
                     bias=bias, resid=R, ldr=n, aux=aux, ldaux=n, preact=pre)
            torch.cuda.synchronize()
            cur = (C, pre)
            if ref is None:
                ref = cur
            elif not (torch.equal(ref[0], C) and torch.equal(ref[1], pre)):
                bad += 1
                print(f"{name}: launch {r} differs from launch 0 ({ops.gemm_last_kernel()}): max |d| = "
                      f"{(ref[0].float() - C.float()).abs().max().item():.3e}")
                break
        else:
            print(f"{name:18s} {ops.gemm_last_kernel():60s} {reps} launches identical")
    # the grouped weight-gradient launch of a ViT block
    shapes = [(3072, 768), (768, 3072), (2304, 768), (768, 768)]
    probs = []
    for mm, nn in shapes:
        A = (torch.rand(M, mm, device=dev) - 0.5).to(torch.bfloat16)
        Bm = (torch.rand(M, nn, device=dev) - 0.5).to(torch.bfloat16)
        C = torch.zeros(mm, nn, device=dev)
        cs = torch.zeros(mm, device=dev)
        probs.append(((A, Bm, C, mm, nn, M, mm, nn, nn, 0), dict(transA=True, transB=True, flags=ops.EPI_OUT_F32 | ops.EPI_ACCUM, colsum=cs)))
    ref = None
    for r in range(reps):
        for a, kw in probs:
            a[2].zero_(); kw["colsum"].zero_()
        ops.gemm_grouped(probs)
        torch.cuda.synchronize()
        cur = [a[2].clone() for a, _ in probs] + [kw["colsum"].clone() for _, kw in probs]
        if ref is None:
            ref = cur
        elif not all(torch.equal(x, y) for x, y in zip(ref, cur)):
            bad += 1
            print("grouped weight gradients: launch", r, "differs", ops.gemm_last_kernel())
            break
    else:
        print(f"{'wgrad group':18s} {ops.gemm_last_kernel():60s} {reps} launches identical")
    print("NOT REPRODUCIBLE:" if bad else "all reproducible:", bad, "shape(s) differ")


if __name__ == "__main__":
    main()
