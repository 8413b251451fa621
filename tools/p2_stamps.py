#!/usr/bin/env python
"""s_memtime stamps of the persistent p2cl_up's phases, taken by ONE thread (experiment build:
VPU_X_loss="-DVPU_P2_STAMPS -DP2_STAMP_THREAD=0" bash csrc/build.sh x; VPU_LIB_FILE=libvpu_hip_x.so).
usage: VPU_LIB_FILE=libvpu_hip_x.so python tools/p2_stamps.py"""
import ctypes as C
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops, _lib
B, S, h, H = 12, 48, 112, 448
low = torch.sigmoid(torch.randn(B, S, h, h, device="cuda"))
gt = (torch.rand(B, 1, H, H, device="cuda") > 0.5).float()
part, dlow = torch.empty(B, S, device="cuda"), torch.empty_like(low)
for _ in range(3):
    ops.p2cl_up_fwd_bwd(low, gt, None, None, part, dlow, 1e-6, B, S, h, h, H, H)
torch.cuda.synchronize()
buf = (C.c_uint32 * (8192 * 8))()
lib = _lib.load()
lib.vpu_dbg_p2.argtypes = [C.c_void_p]
rc = lib.vpu_dbg_p2(C.cast(buf, C.c_void_p))
t = torch.tensor(list(buf), dtype=torch.int64).view(8192, 8)
nwg = B * S * lib.vpu_p2cl_up_nband(h, h)
t = t[:min(nwg, 8192)]
d = lambda a, b: ((t[:, b] - t[:, a]) & 0xffffffff).double()
seq = [(0, 1, "item start -> operands in LDS (barrier)"), (1, 2, "horizontal interpolation (barrier)"), (2, 7, "pixel pass (this thread)"),
       (7, 3, "request of the next item (this thread)"), (3, 4, "loss reduction (two barriers)"), (4, 5, "cell pass + four folds (barriers)"),
       (5, 6, "gradient store + end barrier"), (0, 6, "whole item")]
print(f"rc {rc}; {nwg} items; cycles per phase (median / 90th percentile)")
for a, b, nm in seq:
    x = d(a, b)
    print(f"  {nm:44s} {x.median():8.0f} {x.quantile(0.9):8.0f}")
