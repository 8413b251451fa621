#!/usr/bin/env python
"""Attention forward of the loaded library on fixed inputs -> a file; `cmp a b` compares two such files bit for bit (experiment
library against the product library).  usage: python tools/attn_fwd_compare.py out.pt | python tools/attn_fwd_compare.py cmp a b"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        same = torch.equal(a[k], b[k])
        print(f"{k}: {'bit-identical' if same else 'DIFFERENT'}  max |a - b| {(a[k].float() - b[k].float()).abs().max().item():.3e}")
    sys.exit(0)
from pvpuformer_amd import ops
out = {}
for n, nb, Hh, hd in ((196, 48, 12, 64), (784, 12, 12, 64), (784, 2, 16, 80), (100, 3, 2, 64), (50, 3, 4, 32), (300, 2, 3, 96), (784, 2, 2, 128)):
    D = Hh * hd
    g = torch.Generator(device="cuda").manual_seed(n + hd)
    qkv = (torch.randn(nb * n, 3 * D, device="cuda", generator=g) * 1.5).to(torch.bfloat16)
    O = torch.full((nb * n, D), float("nan"), device="cuda", dtype=torch.bfloat16); lse = torch.zeros(nb * Hh, n, device="cuda")
    ops.attn_fwd(qkv, (qkv, D), (qkv, 2 * D), O, lse, nb, Hh, n, hd, 3 * D, D, hd ** -0.5)
    torch.cuda.synchronize()
    print(n, hd, ops.attn_last_kernel(), "finite:", bool(torch.isfinite(O.float()).all()))
    out[f"o_n{n}_hd{hd}"] = O.cpu(); out[f"lse_n{n}_hd{hd}"] = lse.cpu()
torch.save(out, sys.argv[1])
