#!/usr/bin/env python
"""The reference-faithful training step ("also report the mixed 1-3-iteration step"):
``VPUTrainStep.batch_forward`` = ISTrainer.batch_forward (isegm/engine/trainer.py:310-491) with num_iters ~ randint(1, 3),
click / box prompt type per iteration, the next click and the error-mask label simulated between the iterations
(host bookkeeping, distance transforms on the GPU), each iteration back-propagated, one fused Adam step.
Prints optimizer steps/s and images/s (images = batch size per step, as the reference counts them) next to the
single-iteration rate of bench.py.   usage: python tools/bench_trainstep.py [steps] [batch]
BENCH_PROMPTS=0,1,2 samples click / box / scribble prompts per iteration (BASELINE.json config 4's mix)."""
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pvpuformer_amd.isegm.engine.trainer import VPUTrainStep                      # noqa: E402
from pvpuformer_amd.isegm.model.is_vpu_model import VitMultiGaussianVector_ed_Model   # noqa: E402
from pvpuformer_amd.optim import FusedAdam                                        # noqa: E402
from pvpuformer_amd.synth import synth_batch, vitb_model_kwargs                   # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    torch.manual_seed(0)
    model = VitMultiGaussianVector_ed_Model(**vitb_model_kwargs()).cuda()
    model.set_compute_dtype("bf16")
    model.train()
    eng = model._ensure_engine()
    eng.refresh_weights()
    step = VPUTrainStep(model, optimizer=FusedAdam(model, lr=5e-5))
    batch_host = {k: v.pin_memory() for k, v in synth_batch(B, 448, seed=3, device="cpu").items()}   # a loader's output
    rng, np_rng = random.Random(0), np.random.RandomState(0)
    mixed = os.environ.get("BENCH_PROMPTS", "")          # e.g. "0,1,2": config 4's click / box / scribble mix
    if mixed:
        step.ptypes = tuple(int(t) for t in mixed.split(","))
        print("prompt types sampled per iteration:", step.ptypes)
    # warm-up: every (prompt type, iteration number) pass is host-enqueued the first time and captured the second
    warm = int(os.environ.get("BENCH_WARMUP", "24"))
    for fixed in (1, None):
        iters = 0
        for i in range(steps + warm):
            if i == warm:
                torch.cuda.synchronize(); t0 = time.perf_counter(); iters = 0
            logged, _ = step.batch_forward(step.upload(batch_host, "cuda"), num_iters=fixed, rng=rng, np_rng=np_rng)
            iters += logged["num_iters"]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{'num_iters = 1' if fixed else 'num_iters ~ randint(1,3)'}: {steps / dt:6.2f} steps/s = {steps * B / dt:7.1f} images/s "
              f"({dt / steps * 1e3:6.1f} ms/step, {iters / steps:.2f} forward+backward passes per step)")


if __name__ == "__main__":
    main()
