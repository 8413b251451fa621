#!/usr/bin/env python
"""s_memtime stamps of the persistent window backward (experiment build -DVPU_WIN_STAMPS): the SECOND problem of each workgroup.
usage: VPU_LIB_FILE=libvpu_hip_x.so VPU_ATTN_ONEPASS=3 python tools/winx_stamps.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, Hh, D = 196, 12, 768
qkv = torch.randn(nb * n, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.randn(nb * n, D, device="cuda").to(torch.bfloat16); do = torch.randn(nb * n, D, device="cuda").to(torch.bfloat16)
lse = torch.randn(nb * Hh, n, device="cuda"); delta = torch.zeros(nb * Hh, n, device="cuda")
dqkv = torch.empty_like(qkv)
for _ in range(3):
    ops.attn_bwd((qkv, 0), (qkv, D), (qkv, 2 * D), o, do, lse, delta, (dqkv, 0), (dqkv, D), (dqkv, 2 * D), nb, Hh, n, 64, 3 * D, D, 3 * D, 0.125)
torch.cuda.synchronize()
print(ops.attn_last_kernel())
st = delta.view(torch.int32).cpu().view(nb * Hh, n)[:, :128].view(nb * Hh, 8, 16).long() & 0xffffffff
names = ["switch done", "operands read", "scores(0) done"] + [f"barrier {i}" for i in range(7)] + ["-", "final grads", "stores issued"]
for wg in (0, 100, 255):
    s = st[wg]
    t0 = int(s[:, 0].min())
    print(f"workgroup {wg}, second problem: cycles since its switch, per wave")
    for w in range(8):
        print(f"  wave {w}: " + " ".join(f"{(int(s[w, k]) - t0) & 0xffffffff:6d}" for k in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12)))
print("columns: switch done, operands read, scores(0) done, after barrier 0..6, final grads done, stores issued")
