"""Window attention backward (n = 196, 12 heads) against the number of windows: how the launch time grows with the problems per
CU (usage: python tools/attn_bwd_scale.py; VPU_ATTN_ONEPASS selects the form)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops
dev = "cuda"; D, Hh, n = 768, 12, 196
def timeit(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for nb in (5, 10, 21, 32, 42, 48, 64, 96):
    qkv = torch.randn(nb * n, 3 * D, device=dev).to(torch.bfloat16)
    o = torch.randn(nb * n, D, device=dev).to(torch.bfloat16); do = torch.randn(nb * n, D, device=dev).to(torch.bfloat16)
    lse = torch.randn(nb * Hh, n, device=dev); delta = torch.empty(nb * Hh, n, device=dev)
    dqkv = torch.empty_like(qkv)
    us = timeit(lambda: ops.attn_bwd((qkv, 0), (qkv, D), (qkv, 2 * D), o, do, lse, delta, (dqkv, 0), (dqkv, D), (dqkv, 2 * D), nb, Hh, n, 64, 3 * D, D, 3 * D, 0.125))
    print(f"windows {nb:3d} problems {nb * Hh:5d} ({nb * Hh / 256:5.2f} per CU) {us:8.1f} us  {ops.attn_last_kernel()}")
