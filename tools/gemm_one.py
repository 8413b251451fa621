#!/usr/bin/env python
"""Runs ONE shape of tools/gemm_bench.py a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py <name substring> [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops  # noqa: E402
import gemm_bench as gb  # noqa: E402

name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for nm, tA, tB, m, n, k, flags in gb.SHAPES:
    if name in nm:
        A = (torch.rand((k, m) if tA else (m, k), device="cuda") - 0.5).to(torch.bfloat16)
        Bm = (torch.rand((k, n) if tB else (n, k), device="cuda") - 0.5).to(torch.bfloat16)
        C = torch.zeros(m, n, device="cuda", dtype=torch.float32 if flags & ops.EPI_OUT_F32 else torch.bfloat16)
        bias = torch.rand(n, device="cuda")
        R = torch.zeros(m, n, device="cuda", dtype=torch.bfloat16)
        kw = dict(transA=bool(tA), transB=bool(tB), flags=flags, bias=bias, resid=R, ldr=n, aux=R, ldaux=n, preact=R.clone())
        for _ in range(reps):
            ops.gemm(A, Bm, C, m, n, k, (m if tA else k), (n if tB else k), n, 0, **kw)
        torch.cuda.synchronize()
        print(nm, "done")
        break
