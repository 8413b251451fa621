import os, sys, torch
sys.path.insert(0, os.getcwd())
from pvpuformer_amd.isegm.model.is_vpu_model import VitMultiGaussianVector_ed_Model
from pvpuformer_amd.synth import synth_batch, vitb_model_kwargs
from pvpuformer_amd.isegm.engine.trainer import vpu_step_losses
import pvpuformer_amd.engine as E
dev = torch.device("cuda", 0)
model = VitMultiGaussianVector_ed_Model(**vitb_model_kwargs(embed_dim=1024, depth=24, num_heads=16, patch=16)).to(dev)
model.set_compute_dtype("bf16"); model.train()
eng = model._ensure_engine(); eng.refresh_weights()
B = 8
b = synth_batch(B, 448, seed=0, device=dev)
x = torch.cat([b["images"], torch.zeros(B, 1, 448, 448, device=dev)], 1).contiguous()
orig = E.ops.gemm_grouped
log = []
def grouped(p):
    import traceback
    st = [f.name for f in traceback.extract_stack()[-6:-1]]
    log.append(([(a[3], a[4], a[5]) for a, _ in p], st))
    orig(p)
E.ops.gemm_grouped = grouped
eng.zero_grad()
inst, _ = eng.forward(x, b["points"].float(), None, 0, None, training=True, materialize_aux=False)
losses, d_inst, d_sim = vpu_step_losses(inst, None, b["instances"].float(), None, None, iter_weight=1.0, sim_low=eng.sim_low)
eng.backward(d_inst, None, d_sim_low=d_sim)
torch.cuda.synchronize()
for probs, st in log:
    if any(k >= 2048 for _, _, k in probs):
        print(len(probs), probs[:8], st)
