#!/usr/bin/env python
"""When the engine reports its finished gradient ranges during a data-parallel backward (a stand-in reducer, one GPU): per
range the number of library launches of that backward so far.  usage: python tools/dp_report_timing.py [vitb|vitl|vith] [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pvpuformer_amd import _lib                                                         # noqa: E402
from pvpuformer_amd.isegm.engine.trainer import vpu_step_losses                        # noqa: E402
from pvpuformer_amd.isegm.model.is_vpu_model import VitMultiGaussianVector_ed_Model   # noqa: E402
from pvpuformer_amd.synth import synth_batch, vitb_model_kwargs                       # noqa: E402

MODELS = {"vitb": dict(embed_dim=768, depth=12, num_heads=12, patch=16), "vitl": dict(embed_dim=1024, depth=24, num_heads=16, patch=16),
          "vith": dict(embed_dim=1280, depth=32, num_heads=16, patch=14)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "vitb"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = VitMultiGaussianVector_ed_Model(**vitb_model_kwargs(**MODELS[name])).to(dev)
    model.set_compute_dtype("bf16")
    model.train()
    eng = model._ensure_engine()
    eng.refresh_weights()
    b = synth_batch(B, 448, seed=100, device=dev)
    x = torch.cat([b["images"], torch.zeros(B, 1, 448, 448, device=dev)], 1).contiguous()

    class Red:
        reserve_cus = 16

        def __init__(self):
            self.seen = []

        def ready(self, lo, hi):
            self.seen.append((lo, hi, ncall[0]))
    red, ncall, orig = Red(), [0], _lib.call

    def counting(name_, *a):
        ncall[0] += 1
        return orig(name_, *a)
    for _ in range(3):
        eng.grad_ready_hook = red.ready
        eng.zero_grad()
        inst, _ = eng.forward(x, b["points"], None, 0, None, training=True, materialize_aux=False)
        _, d_inst, d_sim = vpu_step_losses(inst, None, b["instances"], None, None, iter_weight=1.0, sim_low=eng.sim_low)
        red.seen, ncall[0] = [], 0
        _lib.call = counting
        eng.backward(d_inst, None, d_sim_low=d_sim)
        _lib.call = orig
    torch.cuda.synchronize()
    total = ncall[0]
    late = sum(hi - lo for lo, hi, c in red.seen if c >= total - 1)
    print(f"{name} B={B}: {len(red.seen)} ranges over {total} launches; reported at launch", [c for _, _, c in red.seen])
    print(f"  MB per range: {[round((hi - lo) * 4 / 1e6, 1) for lo, hi, _ in red.seen]}")
    print(f"  reported only at the end: {late * 4 / 1e6:.1f} MB of {eng.total * 4 / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
