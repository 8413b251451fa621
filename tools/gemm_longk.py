#!/usr/bin/env python
"""Main-loop rate of the K2 forms as a function of K (dgrad orientation, M = 9408, N = 3072; random bf16 data):
k2 = 1 the 256 x 128 ping-pong form, 3 the 256 x 256 two-stage form.  usage: python tools/gemm_longk.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops  # noqa: E402

M, N = 9408, 3072
for K in (768, 3072, 9216):
    A = (torch.rand(M, K, device="cuda") - 0.5).to(torch.bfloat16)
    Bm = (torch.rand(K, N, device="cuda") - 0.5).to(torch.bfloat16)
    C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    for opt in (1, 3):
        ops.gemm_set_option("k2", opt)
        f = lambda: ops.gemm(A, Bm, C, M, N, K, K, N, N, 0, transB=True, flags=0)
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"K={K:5d} k2={opt} {ops.gemm_last_kernel():42s} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
ops.gemm_set_option("k2", -1)
