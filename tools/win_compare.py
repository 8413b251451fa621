#!/usr/bin/env python
"""Window attention backward of the loaded library on fixed inputs -> a file; `cmp a b` compares two such files bit for bit
(an experiment library against the product library: VPU_LIB_FILE=libvpu_hip_x.so python tools/win_compare.py out_x.pt).
usage: python tools/win_compare.py out.pt | python tools/win_compare.py cmp a.pt b.pt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        same = torch.equal(a[k], b[k])
        d = (a[k].float() - b[k].float()).abs().max().item()
        print(f"{k}: {'bit-identical' if same else 'DIFFERENT'}  max |a - b| {d:.3e}  max |a| {a[k].float().abs().max().item():.3e}")
    sys.exit(0)
from pvpuformer_amd import ops
out = {}
for n, nb, Hh in ((196, 48, 12), (100, 3, 2), (256, 2, 2), (197, 2, 3), (16, 3, 1)):
    D = Hh * 64
    g = torch.Generator(device="cuda").manual_seed(n)
    qkv = (torch.randn(nb * n, 3 * D, device="cuda", generator=g) * 1.5).to(torch.bfloat16)
    O = torch.zeros(nb * n, D, device="cuda", dtype=torch.bfloat16); lse = torch.zeros(nb * Hh, n, device="cuda")
    ops.attn_fwd(qkv, (qkv, D), (qkv, 2 * D), O, lse, nb, Hh, n, 64, 3 * D, D, 0.125)
    dO = torch.randn(nb * n, D, device="cuda", generator=g).to(torch.bfloat16)
    dqkv = torch.full_like(qkv, float("nan")); delta = torch.zeros(nb * Hh, n, device="cuda")
    ops.attn_bwd(qkv, (qkv, D), (qkv, 2 * D), O, dO, lse, delta, dqkv, (dqkv, D), (dqkv, 2 * D), nb, Hh, n, 64, 3 * D, D, 3 * D, 0.125)
    torch.cuda.synchronize()
    print(n, ops.attn_last_kernel(), "finite:", bool(torch.isfinite(dqkv.float()).all()))
    out[f"n{n}"] = dqkv.cpu()
torch.save(out, sys.argv[1])
