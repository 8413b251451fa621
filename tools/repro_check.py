#!/usr/bin/env python
"""Run-to-run reproducibility of one training step (same inputs, same process): logits and every gradient tensor compared
bit for bit between repetitions.  usage: python tools/repro_check.py [vitb|vitl|vith] [batch] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pvpuformer_amd.isegm.engine.trainer import vpu_step_losses                        # noqa: E402
from pvpuformer_amd.isegm.model.is_vpu_model import VitMultiGaussianVector_ed_Model   # noqa: E402
from pvpuformer_amd.synth import synth_batch, vitb_model_kwargs                       # noqa: E402

MODELS = {"vitb": dict(embed_dim=768, depth=12, num_heads=12, patch=16), "vitl": dict(embed_dim=1024, depth=24, num_heads=16, patch=16),
          "vith": dict(embed_dim=1280, depth=32, num_heads=16, patch=14)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "vitb"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = VitMultiGaussianVector_ed_Model(**vitb_model_kwargs(**MODELS[name])).to(dev)
    model.set_compute_dtype("bf16")
    model.train()
    eng = model._ensure_engine()
    eng.refresh_weights()
    b = synth_batch(B, 448, seed=100, device=dev)
    x = torch.cat([b["images"], torch.zeros(B, 1, 448, 448, device=dev)], 1).contiguous()
    ref = None
    poison = os.environ.get("REPRO_POISON", "0") == "1"
    for r in range(reps):
        if poison and r > 0:
            # hand the allocator memory full of NaNs: whatever a kernel reads without having written it shows up as NaN
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            big = [torch.full((1 << 28,), float("nan"), device=dev) for _ in range(24)]          # 24 x 1 GiB
            small = [torch.full((n,), float("nan"), device=dev) for n in (256, 4096, 65536, 200000) for _ in range(400)]
            del big, small
        eng.zero_grad()
        inst, _ = eng.forward(x, b["points"], None, 0, None, training=True, materialize_aux=False)
        sim = eng.sim_low.clone()
        _, d_inst, d_sim = vpu_step_losses(inst, None, b["instances"], None, None, iter_weight=1.0, sim_low=eng.sim_low)
        eng.backward(d_inst, None, d_sim_low=d_sim)
        torch.cuda.synchronize()
        cur = (inst.clone(), sim, d_inst.clone(), d_sim.clone(), eng.gflat.clone())
        if ref is None:
            ref = cur
            continue
        same = [torch.equal(a, c) for a, c in zip(ref, cur)]
        if poison:
            print("   non-finite values: logits", int((~torch.isfinite(cur[0])).sum()), "sim_low", int((~torch.isfinite(cur[1])).sum()),
                  "gradients", int((~torch.isfinite(cur[4])).sum()))
        bad = []
        if not same[4]:
            for n, (off, shape, numel) in eng.names.items():
                if not torch.equal(ref[4][off:off + numel], cur[4][off:off + numel]):
                    d = (ref[4][off:off + numel] - cur[4][off:off + numel]).abs().max().item()
                    bad.append((n, d, ref[4][off:off + numel].abs().max().item()))
        print(f"{name} B={B} rep {r}: logits {same[0]} sim_low {same[1]} d_inst {same[2]} d_sim {same[3]} gradients {same[4]}"
              f"; {len(bad)} gradient tensors differ", bad[:6], "... last:", bad[-3:] if len(bad) > 6 else "")


if __name__ == "__main__":
    main()
