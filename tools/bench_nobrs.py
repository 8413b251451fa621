#!/usr/bin/env python
"""Timing of the NoBRS evaluation loop (BASELINE.json configs[2]: ViT-B 448, 20-click budget, flip TTA => batch 2, no
grad) on one synthetic 448x448 image: seconds per click split into the model forward and the host side (oracle click
from the Clicker's distance transform, prompt packing, transforms).  usage: python tools/bench_nobrs.py [clicks]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pvpuformer_amd.isegm.inference.clicker import Clicker            # noqa: E402
from pvpuformer_amd.isegm.inference.predictors import get_predictor   # noqa: E402
from pvpuformer_amd.isegm.model.is_vpu_model import VitMultiGaussianVector_ed_Model   # noqa: E402
from pvpuformer_amd.synth import synth_batch, vitb_model_kwargs       # noqa: E402


def main():
    clicks = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    torch.manual_seed(0)
    model = VitMultiGaussianVector_ed_Model(**vitb_model_kwargs()).cuda()
    model.set_compute_dtype("bf16")
    model.eval()
    model.weights_frozen = True
    b = synth_batch(1, 448, seed=7, device="cpu")
    image = (b["images"][0].permute(1, 2, 0).numpy() * 255).astype(np.uint8)
    gt = b["instances"][0, 0].numpy().astype(np.int32)
    pred = get_predictor(model, "NoBRS", "cuda", with_flip=True, zoom_in_params=dict(skip_clicks=-1, target_size=(448, 448)))
    fwd_ms = []
    orig = model.forward

    def timed(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = orig(*a, **k)
        torch.cuda.synchronize(); fwd_ms.append((time.perf_counter() - t0) * 1e3)
        return out
    model.forward = timed
    for rep in range(2):     # first pass warms up
        pred.set_input_image(image)
        clicker = Clicker(gt_mask=gt)
        mask = np.zeros_like(gt, dtype=bool)
        fwd_ms.clear()
        t0 = time.perf_counter()
        for i in range(clicks):
            clicker.make_next_click(mask)
            probs, _ = pred.get_vqu_prediction(clicker, gt_mask=gt, as_prompt_type=0, click_indx=i)
            mask = probs > 0.49
        total = time.perf_counter() - t0
    print(f"NoBRS ViT-B/448 bf16, flip TTA (batch 2), {clicks} clicks: {total / clicks * 1e3:.1f} ms per click, of which "
          f"model forward {np.mean(fwd_ms):.2f} ms (min {np.min(fwd_ms):.2f}); host side {total / clicks * 1e3 - np.mean(fwd_ms):.1f} ms")


if __name__ == "__main__":
    main()
