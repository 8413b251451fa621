#!/usr/bin/env python
"""p2cl_up of the loaded library on fixed inputs -> a file; `cmp a b` compares two such files bit for bit (experiment library
against the product library).  usage: python tools/p2_compare.py out.pt | python tools/p2_compare.py cmp a.pt b.pt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        print(f"{k}: {'bit-identical' if torch.equal(a[k], b[k]) else 'DIFFERENT'}  max |a - b| {(a[k].double() - b[k].double()).abs().max().item():.3e}")
    sys.exit(0)
from pvpuformer_amd import ops
out = {}
for (B, S, h, H, soft) in ((12, 48, 112, 448, False), (2, 4, 28, 112, False), (3, 6, 56, 224, True), (1, 2, 112, 448, False), (5, 48, 112, 448, False)):
    g = torch.Generator(device="cuda").manual_seed(B * 100 + h)
    low = torch.sigmoid(torch.randn(B, S, h, h, device="cuda", generator=g))
    gt = (torch.rand(B, 1, H, H, device="cuda", generator=g) > 0.5).float()
    if soft:
        gt[:, :, ::7, ::5] = 0.3
        gt[:, :, 3::11, 2::9] = -1.0
    dlow = torch.full_like(low, float("nan"))
    bands = ops.p2cl_up_fwd_bwd(low, gt, None, None, None, dlow, 1e-6, B, S, h, h, H, H)
    lo = ops.p2cl_up_fwd_bwd(low, gt, None, None, None, None, 1e-6, B, S, h, h, H, H)
    torch.cuda.synchronize()
    print(B, S, h, H, "finite:", bool(torch.isfinite(dlow).all()), float(bands.double().sum()))
    out[f"bands_{B}_{S}_{h}"] = bands.cpu(); out[f"dlow_{B}_{S}_{h}"] = dlow.cpu(); out[f"lossonly_{B}_{S}_{h}"] = lo.cpu()
torch.save(out, sys.argv[1])
