#!/usr/bin/env python
"""Micro-benchmark of the non-GEMM kernels at the ViT-B bs=12 shapes (HIP-event timing, random data).
usage: python tools/op_bench.py [name-substring ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops  # noqa: E402

dev = "cuda"


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    only = sys.argv[1:]
    B, S, h, H = 12, 48, 112, 448
    cases = {}
    low = torch.sigmoid(torch.randn(B, S, h, h, device=dev))
    gt = (torch.rand(B, 1, H, H, device=dev) > 0.5).float()
    part, dlow = torch.empty(B, S, device=dev), torch.empty_like(low)
    cases["p2cl_up"] = (lambda: ops.p2cl_up_fwd_bwd(low, gt, None, None, part, dlow, 1e-6, B, S, h, h, H, H),
                        low.numel() * 8 + gt.numel() * 4)
    cases["p2cl_up_loss_only"] = (lambda: ops.p2cl_up_fwd_bwd(low, gt, None, None, part, None, 1e-6, B, S, h, h, H, H),
                                  low.numel() * 4 + gt.numel() * 4)      # (no gradient output: the pixel pass alone)
    d_inst, dseg = torch.randn(B, 1, H, H, device=dev), torch.empty(B, h * h, device=dev)
    cases["upsample_ac_bwd"] = (lambda: ops.upsample_ac_bwd(d_inst, dseg, B, h, h, H, H), d_inst.numel() * 4 + dseg.numel() * 4)
    logits = torch.randn(B, 1, H, H, device=dev)
    out, dl = torch.empty(B, 2, device=dev), torch.empty(B, H * H, device=dev)
    cases["nfl_dice"] = (lambda: ops.nfl_dice_fwd_bwd(logits, gt, None, out, dl, 1.0, 1.0, B, H * H), logits.numel() * 20)
    for r, hh in ((8, 14), (4, 28), (2, 56)):
        C = 256
        dcat = torch.randn(B * 112 * 112, 4 * C, device=dev).to(torch.bfloat16)
        din = torch.empty(B * hh * hh, C, device=dev, dtype=torch.bfloat16)
        cases[f"bilinear_bwd_r{r}"] = (lambda dcat=dcat, din=din, hh=hh: ops.bilinear_cl_bwd((dcat, C), 4 * C, din, C, B, hh, hh, 112, 112, C, 0),
                                       B * 112 * 112 * C * 2)
        cases[f"bilinear_fwd_r{r}"] = (lambda dcat=dcat, din=din, hh=hh: ops.bilinear_cl_fwd(din, C, (dcat, C), 4 * C, B, hh, hh, 112, 112, C, 0),
                                       B * 112 * 112 * C * 2)
    D, Hh = 768, 12
    for tag, nb, n in (("window", 48, 196), ("global", 12, 784)):
        qkv = torch.randn(nb * n, 3 * D, device=dev).to(torch.bfloat16)
        o, do = torch.empty(nb * n, D, device=dev, dtype=torch.bfloat16), torch.randn(nb * n, D, device=dev).to(torch.bfloat16)
        lse, delta = torch.empty(nb * Hh, n, device=dev), torch.empty(nb * Hh, n, device=dev)
        dqkv = torch.empty_like(qkv)
        fl = 4.0 * n * n * 64 * nb * Hh
        cases[f"attn_fwd_{tag}"] = (lambda qkv=qkv, o=o, lse=lse, nb=nb, n=n: ops.attn_fwd((qkv, 0), (qkv, D), (qkv, 2 * D), o, lse, nb, Hh, n, 64, 3 * D, D, 0.125), fl)
        cases[f"attn_bwd_{tag}"] = (lambda qkv=qkv, o=o, do=do, lse=lse, delta=delta, dqkv=dqkv, nb=nb, n=n: ops.attn_bwd(
            (qkv, 0), (qkv, D), (qkv, 2 * D), o, do, lse, delta, (dqkv, 0), (dqkv, D), (dqkv, 2 * D), nb, Hh, n, 64, 3 * D, D, 3 * D, 0.125), 2.5 * fl)
    rows, Cd = 9408, 768
    xl = torch.randn(rows, Cd, device=dev).to(torch.bfloat16)
    dyl, drl = torch.randn_like(xl), torch.randn_like(xl)
    yl, dxl = torch.empty_like(xl), torch.empty_like(xl)
    wl, bl = torch.rand(Cd, device=dev), torch.rand(Cd, device=dev)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    partl = torch.empty(ops.layernorm_bwd_nblk(rows), 2, Cd, device=dev)
    cases["layernorm_fwd"] = (lambda: ops.layernorm_fwd(xl, wl, bl, yl, mean, rstd, rows, Cd, 1e-6), rows * Cd * 4)
    cases["layernorm_bwd"] = (lambda: ops.layernorm_bwd(dyl, xl, wl, mean, rstd, drl, dxl, partl, rows, Cd), rows * Cd * 8)
    Bg, HWg, Cg = 12, 12544, 128
    xg = torch.randn(Bg * HWg, Cg, device=dev).to(torch.bfloat16)
    dyg, yg, dxg = torch.randn_like(xg), torch.empty_like(xg), torch.empty_like(xg)
    wg_, bg_ = torch.rand(Cg, device=dev), torch.rand(Cg, device=dev)
    mg, rg = torch.empty(Bg, device=dev), torch.empty(Bg, device=dev)
    nch = ops.groupnorm_nchunk()
    stg = torch.empty(Bg, nch, 2, device=dev, dtype=torch.float64)
    pg = torch.empty(Bg * nch, 2, Cg, device=dev)
    cases["groupnorm_fwd"] = (lambda: ops.groupnorm_fwd(xg, wg_, bg_, yg, mg, rg, stg, Bg, HWg, Cg, 1e-5, True), xg.numel() * 6)
    cases["groupnorm_bwd"] = (lambda: ops.groupnorm_bwd(dyg, xg, wg_, bg_, mg, rg, dxg, pg, stg, Bg, HWg, Cg, True), xg.numel() * 10)
    for name, (fn, nbytes) in cases.items():
        if only and not any(t in name for t in only):
            continue
        us = timeit(fn)
        unit = "TFLOP/s" if name.startswith("attn") else "TB/s (algorithmic bytes)"
        print(f"{name:22s} {us:9.1f} us   {nbytes / us / 1e6:8.2f} {unit}")


if __name__ == "__main__":
    main()
