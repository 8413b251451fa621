#!/usr/bin/env python
"""Where the host spends a reference-faithful training step (tools/bench_trainstep.py's loop): wall time of batch_forward
without a synchronisation (the host's share: enqueue / replay + the simulators' read-backs) beside the step time, and the
time inside the prompt simulator.  usage: python tools/trainstep_host_time.py [steps] [num_iters or 0 for randint(1,3)]"""
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pvpuformer_amd.isegm.engine import prompt_sim                                  # noqa: E402
from pvpuformer_amd.isegm.engine.trainer import VPUTrainStep                        # noqa: E402
from pvpuformer_amd.isegm.model.is_vpu_model import VitMultiGaussianVector_ed_Model   # noqa: E402
from pvpuformer_amd.optim import FusedAdam                                          # noqa: E402
from pvpuformer_amd.synth import synth_batch, vitb_model_kwargs                     # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    fixed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    torch.manual_seed(0)
    model = VitMultiGaussianVector_ed_Model(**vitb_model_kwargs()).cuda()
    model.set_compute_dtype("bf16")
    model.train()
    model._ensure_engine().refresh_weights()
    step = VPUTrainStep(model, optimizer=FusedAdam(model, lr=5e-5))
    batch_host = {k: v.pin_memory() for k, v in synth_batch(12, 448, seed=3, device="cpu").items()}
    rng, np_rng = random.Random(0), np.random.RandomState(0)
    sim_t = [0.0]
    inner = prompt_sim._get_next_promts_gpu

    def timed(*a, **k):
        t = time.perf_counter()
        r = inner(*a, **k)
        sim_t[0] += time.perf_counter() - t
        return r
    prompt_sim._get_next_promts_gpu = timed
    host = up = 0.0
    for i in range(steps + 24):
        if i == 24:
            torch.cuda.synchronize(); t0 = time.perf_counter(); host = up = 0.0; sim_t[0] = 0.0
        t = time.perf_counter()
        b = step.upload(batch_host, "cuda")
        t1 = time.perf_counter()
        step.batch_forward(b, num_iters=fixed or None, rng=rng, np_rng=np_rng)
        host += time.perf_counter() - t1
        up += t1 - t
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"step {dt / steps * 1e3:.2f} ms; host in batch_forward {host / steps * 1e3:.2f} ms (simulator {sim_t[0] / steps * 1e3:.2f} ms), "
          f"upload {up / steps * 1e3:.2f} ms; graph {'on' if step.use_graph else 'off'}")


if __name__ == "__main__":
    main()
