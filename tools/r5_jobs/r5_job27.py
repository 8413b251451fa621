"""tinyh gradient-norm errors per parameter under the laboratory knob VPU_GEMM_K2_NARROW_MIN (set by the caller)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import vpu_oracle as vo
import test_model_gpu as T
fx, cfg, sd, model, batch, img4 = T._setup(os.path.join(ROOT, "tests", "golden"), "tinyh.npz", "bf16")
model.zero_grad()
out = T._run(model, img4, batch, 0)
gt = batch["instances"].cuda()
total, _ = vo.step_loss(out, gt, vo.ed_mask_label(gt))
total.backward()
params = dict(model.named_parameters())
names = [str(n) for n in fx["click_grad_names"]]
norms = dict(zip(names, fx["click_grad_norms"]))
errs = sorted(((abs(float(params[n].grad.norm()) - norms[n]) / norms[n], n, norms[n]) for n in names if norms[n] > 1e-4), reverse=True)
print(os.environ.get("VPU_GEMM_K2_NARROW_MIN"), cfg)
for e, n, v in errs[:8]:
    print(f"  {e:.4f} {n} (norm {v:.3e}, shape {tuple(params[n].shape)})")
