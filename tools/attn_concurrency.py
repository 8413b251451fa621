#!/usr/bin/env python
"""Do two latency-bound attention kernels overlap when they run on two streams?  Two independent window-attention
backward problems (ViT-B bs 12 shapes), back to back on one stream vs side by side on two."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops

def mk(nb, n, Hh=12, hd=64):
    D = Hh * hd
    qkv = (torch.randn(nb * n, 3 * D, device="cuda") * 0.5).to(torch.bfloat16)
    o = torch.zeros(nb * n, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(nb * Hh, n, device="cuda")
    ops.attn_fwd((qkv, 0), (qkv, D), (qkv, 2 * D), o, lse, nb, Hh, n, hd, 3 * D, D, 0.125)
    do = torch.randn_like(o)
    return dict(qkv=qkv, o=o, lse=lse, do=do, delta=torch.zeros_like(lse), dqkv=torch.zeros_like(qkv), nb=nb, n=n, D=D, Hh=Hh)

def bwd(s):
    D = s["D"]
    ops.attn_bwd((s["qkv"], 0), (s["qkv"], D), (s["qkv"], 2 * D), s["o"], s["do"], s["lse"], s["delta"], (s["dqkv"], 0), (s["dqkv"], D),
                 (s["dqkv"], 2 * D), s["nb"], s["Hh"], s["n"], 64, 3 * D, D, 3 * D, 0.125)

for tag, nb, n in (("window", 48, 196), ("global", 12, 784)):
    a, b = mk(nb, n), mk(nb, n)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3): bwd(a); bwd(b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): bwd(a); bwd(b)
    e1.record(); torch.cuda.synchronize()
    serial = e0.elapsed_time(e1) / 20
    main = torch.cuda.current_stream()
    e0.record()
    for _ in range(20):
        s1.wait_stream(main); s2.wait_stream(main)
        with torch.cuda.stream(s1): bwd(a)
        with torch.cuda.stream(s2): bwd(b)
        main.wait_stream(s1); main.wait_stream(s2)
    e1.record(); torch.cuda.synchronize()
    par = e0.elapsed_time(e1) / 20
    print(f"{tag}: two backward passes back to back {serial * 1e3:7.1f} us, on two streams {par * 1e3:7.1f} us")
