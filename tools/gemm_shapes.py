#!/usr/bin/env python
"""Every GEMM launch of one training step (bench.py's configuration) with its shape, flags and kernel, grouped by shape, and the time
of each distinct launch repeated alone (GEMM_SHAPES_MAX_ROWS: only problems with at most that many rows; default 4096 -- the neck's
prompt-token chain and the small maps).  usage: python tools/gemm_shapes.py [batch]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops  # noqa: E402
from pvpuformer_amd.isegm.model.is_vpu_model import VitMultiGaussianVector_ed_Model  # noqa: E402
from pvpuformer_amd.isegm.engine.trainer import vpu_step_losses  # noqa: E402
from pvpuformer_amd.synth import synth_batch, vitb_model_kwargs  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    cap = int(os.environ.get("GEMM_SHAPES_MAX_ROWS", "4096"))
    dev = "cuda"
    torch.manual_seed(0)
    model = VitMultiGaussianVector_ed_Model(**vitb_model_kwargs()).to(dev)
    model.set_compute_dtype("bf16")
    model.train()
    eng = model._ensure_engine()
    eng.refresh_weights()
    batch = synth_batch(B, 448, seed=100, device=dev)
    image4 = torch.cat([batch["images"], torch.zeros(B, 1, 448, 448, device=dev)], 1).contiguous()
    log = []
    real_gemm, real_grouped = ops.gemm, ops.gemm_grouped

    def gemm(*a, **kw):
        real_gemm(*a, **kw)
        log.append(("gemm", a, kw, ops.gemm_last_kernel()))

    def grouped(problems):
        real_grouped(problems)
        log.append(("grouped", problems, None, ops.gemm_last_kernel()))

    def step():
        eng.zero_grad(lazy=True)
        mask = ops.dropout_mask(B, model.head.channels, 1.0 - model.head.dropout_ratio, dev)
        inst, _ = eng.forward(image4, batch["points"], None, 0, mask, training=True, materialize_aux=False)
        _, d_inst, d_sim = vpu_step_losses(inst, None, batch["instances"], None, None, iter_weight=1.0, sim_low=eng.sim_low)
        eng.backward(d_inst, None, d_sim_low=d_sim)

    step()
    ops.gemm, ops.gemm_grouped = gemm, grouped
    step()
    ops.gemm, ops.gemm_grouped = real_gemm, real_grouped
    torch.cuda.synchronize()

    def key(a, kw):
        return (a[3], a[4], a[5], int(bool(kw.get("transA"))), int(bool(kw.get("transB"))), kw.get("flags", 0), kw.get("batch", 1))

    groups = collections.OrderedDict()
    for kind, a, kw, name in log:
        if kind == "gemm":
            k = ("gemm",) + key(a, kw)
            rows = a[3] * kw.get("batch", 1)
        else:
            k = ("grouped",) + tuple(key(pa, pk) for pa, pk in a)
            rows = max(pa[3] for pa, pk in a)
        if rows > cap:
            continue
        g = groups.setdefault(k, dict(n=0, call=(kind, a, kw), name=name))
        g["n"] += 1
    total = 0.0
    for k, g in groups.items():
        kind, a, kw = g["call"]
        fn = (lambda: real_gemm(*a, **kw)) if kind == "gemm" else (lambda: real_grouped(a))
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 20
        total += us * g["n"]
        print(f"{g['n']:3d} x {us:6.1f} us  {k}  {g['name']}")
    print(f"total {total / 1e3:.3f} ms per step in {sum(g['n'] for g in groups.values())} launches (each repeated alone, warm)")


if __name__ == "__main__":
    main()
