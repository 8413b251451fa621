#!/usr/bin/env python
"""head_grad_fused of the loaded library on fixed inputs: launch time, and the outputs into a file; `cmp a b` compares two files bit
for bit.  usage: [VPU_LIB_FILE=libvpu_hip_x.so] python tools/headgrad_ab.py out.pt | python tools/headgrad_ab.py cmp a.pt b.pt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        print(f"{k}: {'bit-identical' if torch.equal(a[k], b[k]) else 'DIFFERENT'}")
    sys.exit(0)
from pvpuformer_amd import ops, _lib
B, HW, Cd = 12, 112 * 112, 256
rows = B * HW
g = torch.Generator(device="cuda").manual_seed(5)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
dfn, y, x = r(rows, Cd).bfloat16(), r(rows, Cd).bfloat16(), r(rows, Cd).bfloat16()
inv, dout, w = r(rows).abs() + 0.5, r(rows), r(Cd)
mask = (torch.rand(B, Cd, device="cuda", generator=g) > 0.1).float()
nblk = _lib.load().vpu_convseg_bwd_nblk(rows)
out = {}
for tag, mk in (("mask", mask), ("nomask", None)):
    dx = torch.full_like(x, float("nan")); part = torch.zeros(nblk, Cd, device="cuda"); part_b = torch.zeros(nblk, device="cuda")
    fn = lambda: ops.head_grad_fused(dfn, y, inv, dout, x, w, mk, dx, part, part_b, rows, HW, Cd)
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"head_grad_fused ({tag}): {e0.elapsed_time(e1) * 1e3 / 30:.1f} us")
    out[f"dx_{tag}"] = dx.cpu(); out[f"part_{tag}"] = part.cpu(); out[f"part_b_{tag}"] = part_b.cpu()
torch.save(out, sys.argv[1])
