"""Fused Adam over the engine's flat parameter buffer (torch.optim.Adam semantics as configured by the reference at
models/iSegNet/vpu_base448_cocolvis.py:149-154: lr 5e-5, betas (0.9, 0.999), eps 1e-8; isegm/engine/optimizer.py:6-27
builds one param group PER TENSOR -> 349 tiny kernels per step; here it is ONE launch that also writes the bf16 shadow
used by the MFMA GEMMs)."""
import torch

from . import ops


class FusedAdam:
    def __init__(self, model, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.m = self.v = None

    def _state(self, eng):
        if self.m is None or self.m.numel() != eng.total or self.m.device != eng.flat.device:
            self.m = torch.zeros_like(eng.flat)
            self.v = torch.zeros_like(eng.flat)
        return self.m, self.v

    def step(self, grad_scale=1.0):
        """``grad_scale`` multiplies the gradient first (1/world_size after a SUM all-reduce)."""
        eng = self.model._ensure_engine()
        m, v = self._state(eng)
        self.step_count += 1
        ops.adam_step(eng.flat, eng.gflat, m, v, eng.shadow, eng.total, self.lr, self.betas[0], self.betas[1], self.eps,
                      self.weight_decay, self.step_count, grad_scale)
        eng.refresh_weights(shadow_is_fresh=True)

    def zero_grad(self):
        self.model._ensure_engine().zero_grad()

    def state_dict(self):
        return {"step": self.step_count, "m": self.m, "v": self.v, "lr": self.lr}

    def load_state_dict(self, sd):
        self.step_count, self.m, self.v, self.lr = sd["step"], sd["m"], sd["v"], sd["lr"]
