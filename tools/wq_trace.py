#!/usr/bin/env python
"""Every GEMM launch of one backward pass of the bench step (after one warm-up pass, so that the packed queue knows its
totals): kernel, problems (N, K, reduction, batch entries), tiles, HIP-event time.  usage: python tools/wq_trace.py [vitb|vitl|vith] [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops  # noqa: E402
from pvpuformer_amd.isegm.engine.trainer import vpu_step_losses  # noqa: E402
from pvpuformer_amd.isegm.model.is_vpu_model import VitMultiGaussianVector_ed_Model  # noqa: E402
from pvpuformer_amd.synth import synth_batch, vitb_model_kwargs  # noqa: E402
import pvpuformer_amd.engine as E  # noqa: E402

MODELS = {"vitb": dict(embed_dim=768, depth=12, num_heads=12, patch=16), "vitl": dict(embed_dim=1024, depth=24, num_heads=16, patch=16),
          "vith": dict(embed_dim=1280, depth=32, num_heads=16, patch=14)}
name = sys.argv[1] if len(sys.argv) > 1 else "vitb"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0)
model = VitMultiGaussianVector_ed_Model(**vitb_model_kwargs(**MODELS[name])).to(dev)
model.set_compute_dtype("bf16"); model.train()
eng = model._ensure_engine(); eng.refresh_weights()
b = synth_batch(B, 448, seed=0, device=dev)
x = torch.cat([b["images"], torch.zeros(B, 1, 448, 448, device=dev)], 1).contiguous()
log = []
og, ogg = E.ops.gemm, E.ops.gemm_grouped


def step(record):
    eng.zero_grad()
    inst, _ = eng.forward(x, b["points"].float(), None, 0, None, training=True, materialize_aux=False)
    losses, d_inst, d_sim = vpu_step_losses(inst, None, b["instances"].float(), None, None, iter_weight=1.0, sim_low=eng.sim_low)
    if record:
        def gemm(A, Bm, C, M, N, K, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); og(A, Bm, C, M, N, K, *a, **kw); e1.record()
            log.append((ops.gemm_last_kernel(), [(M, N, K, kw.get("batch", 1))], e0, e1, bool(kw.get("transA"))))
        def grouped(p):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ogg(p); e1.record()
            log.append((ops.gemm_last_kernel(), [(a[3], a[4], a[5], k.get("batch", 1)) for a, k in p], e0, e1, bool(p[0][1].get("transA"))))
        E.ops.gemm, E.ops.gemm_grouped = gemm, grouped
    eng.backward(d_inst, None, d_sim_low=d_sim)
    E.ops.gemm, E.ops.gemm_grouped = og, ogg
    torch.cuda.synchronize()


step(False); step(False); step(True)
tn = eng.wgrad_tn
tot = 0.0
for kern, probs, e0, e1, tA in log:
    if not tA:
        continue
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    tiles = sum(((m + 255) // 256) * ((n + tn - 1) // tn) * bt for m, n, k, bt in probs)
    fl = sum(2.0 * m * n * k * bt for m, n, k, bt in probs)
    print(f"{us:8.1f} us {fl / us / 1e6:7.0f} TF  tiles({tn}) {tiles:4d}  {kern[10:40]:30s} {probs}")
print(f"weight-gradient launches: {tot:.0f} us")
