#!/usr/bin/env python
"""What a collective's channel workgroups do to the persistent one-workgroup-per-CU GEMM launches, measured on ONE GPU:
N single-workgroup spin kernels (torch.cuda._sleep, one stream each) stand in for N RCCL channels; 20 GEMMs of the fc1-dgrad
shape run on the main stream meanwhile, with and without the reserve.  usage: python tools/reserve_cus_experiment.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops

M, N, K = 9408, 768, 3072
A = (torch.rand(M, K, device="cuda") - 0.5).to(torch.bfloat16)
W = (torch.rand(K, N, device="cuda") - 0.5).to(torch.bfloat16)       # dgrad orientation: B is [K, N]
C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
streams = [torch.cuda.Stream() for _ in range(64)]
side = torch.cuda.Stream()
sink = torch.zeros(16, device="cuda")

def gemms(n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.gemm(A, W, C, M, N, K, K, N, N, 0, transB=True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

for _ in range(3): gemms(5)
base = gemms()
print(f"no other work:                                  {base:6.1f} us per GEMM  ({ops.gemm_last_kernel()})")
for nsleep in (16, 32):
    for reserve in (0, 16, 32):
        ops.gemm_set_option("reserve_cus", reserve)
        torch.cuda.synchronize()
        for s in streams[:nsleep]:
            with torch.cuda.stream(s):
                torch.cuda._sleep(12_000_000)          # ~5 ms: outlasts the 20 GEMMs
        t = gemms()
        torch.cuda.synchronize()
        print(f"{nsleep:2d} spinning workgroups, reserve_cus = {reserve:2d}:        {t:6.1f} us per GEMM  ({ops.gemm_last_kernel()})")
print("channel-sized workgroups (512 threads, ~96 registers, 16 KiB LDS: vpu_debug_spin), ONE launch on a second stream:")
for nsleep in (16, 32, 64):
    for reserve in (0, 16, 32, 64):
        ops.gemm_set_option("reserve_cus", reserve)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            ops.debug_spin(sink, nsleep, 12_000_000)      # ~5 ms
        t = gemms()
        torch.cuda.synchronize()
        print(f"{nsleep:2d} channel workgroups, reserve_cus = {reserve:2d}:         {t:6.1f} us per GEMM  ({ops.gemm_last_kernel()})")
ops.gemm_set_option("reserve_cus", 0)
print("the 128 x 128 kernel (two workgroups per CU, 444 tiles) beside the same channel workgroups, no reserve:")
ops.gemm_set_option("k2", 0)
base0 = gemms()
print(f"alone:                                           {base0:6.1f} us per GEMM  ({ops.gemm_last_kernel()})")
for nsleep in (16, 32, 64):
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        ops.debug_spin(sink, nsleep, 12_000_000)
    t = gemms()
    torch.cuda.synchronize()
    print(f"{nsleep:2d} channel workgroups:                           {t:6.1f} us per GEMM  ({ops.gemm_last_kernel()})")
ops.gemm_set_option("k2", -1)
