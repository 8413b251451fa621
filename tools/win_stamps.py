#!/usr/bin/env python
"""Reads the s_memtime stamps an experiment build of attn_bwd_win_kernel (-DVPU_WIN_STAMPS) leaves in the `delta` buffer: per wave,
per 32-query block: cycles from the barrier to the end of the block's work, and from there to the next barrier.
usage: VPU_LIB_FILE=libvpu_hip_x.so python tools/win_stamps.py [windows]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 48
n, Hh, D = 196, 12, 768
qkv = torch.randn(nb * n, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.randn(nb * n, D, device="cuda").to(torch.bfloat16); do = torch.randn(nb * n, D, device="cuda").to(torch.bfloat16)
lse = torch.randn(nb * Hh, n, device="cuda"); delta = torch.zeros(nb * Hh, n, device="cuda")
dqkv = torch.empty_like(qkv)
for _ in range(3):
    ops.attn_bwd((qkv, 0), (qkv, D), (qkv, 2 * D), o, do, lse, delta, (dqkv, 0), (dqkv, D), (dqkv, 2 * D), nb, Hh, n, 64, 3 * D, D, 3 * D, 0.125)
torch.cuda.synchronize()
st = delta.view(torch.int32).cpu().view(nb * Hh, n)[:, :128].view(nb * Hh, 8, 16).long() & 0xffffffff
for prob in (0, 100, nb * Hh - 1):
    s = st[prob]
    t0 = s[:, 15].min()
    print(f"problem {prob}: per wave: start, scores(0) begins, [barrier->end, end->next barrier] x 6, kernel end (cycles since the first wave's start)")
    for w in range(8):
        r = s[w]
        row = [int((r[15] - t0) & 0xffffffff), int((r[14] - t0) & 0xffffffff)]
        it = []
        for i in range(6):
            a, b = r[2 * i], r[2 * i + 1]
            nxt = r[2 * i + 2] if i < 5 else r[13]
            it.append(f"{int((b - a) & 0xffffffff):5d}/{int((nxt - b) & 0xffffffff):5d}")
        print(f"  wave {w}: {row[0]:6d} {row[1]:6d} | " + " ".join(it) + f" | first barrier at {int((r[0] - t0) & 0xffffffff):6d}, end {int((r[13] - t0) & 0xffffffff):6d}")
