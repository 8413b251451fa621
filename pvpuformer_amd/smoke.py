"""smoke(): one small invocation of the hot path on cuda:0 (tiny VPUFormer config, forward + losses + backward + one
fused-Adam step through the HIP kernels), checked against the CPU oracle on the same inputs."""
import os
import sys

import torch


def run_smoke():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    import vpu_oracle as vo  # checker only
    from pvpuformer_amd.isegm.engine.trainer import vpu_step_losses
    from pvpuformer_amd.isegm.model.is_vpu_model import VitMultiGaussianVector_ed_Model
    from pvpuformer_amd.optim import FusedAdam
    from pvpuformer_amd.synth import vitb_model_kwargs
    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    cfg = vo.make_cfg(embed_dim=128, depth=8, num_heads=4, out_dims=(16, 32, 64, 128), head_channels=32)
    model = VitMultiGaussianVector_ed_Model(**vitb_model_kwargs(embed_dim=128, depth=8, num_heads=4,
                                                                out_dims=(16, 32, 64, 128), channels=32)).cuda()
    sd = vo.synth_state_dict(vo.param_shapes(cfg), seed=0)
    model.load_state_dict(sd, strict=True)
    model.eval()
    b = vo.synth_batch(2, 448, seed=7)
    img4 = torch.cat([b["images"], torch.zeros(2, 1, 448, 448)], 1)
    with torch.no_grad():
        ref = vo.vpu_forward(sd, cfg, img4, b["points"])
    for dtype, tol in (("f32", 1e-3), ("bf16", 2e-2)):      # bf16: measured 1.08e-2 (bound = 1.85 x that)
        model.set_compute_dtype(dtype)
        eng = model._ensure_engine()
        eng.refresh_weights()
        eng.zero_grad()
        inst, aux = eng.forward(img4.cuda(), b["points"].cuda(), None, 0, None, training=True)
        err = (inst.cpu() - ref["instances"]).abs().max().item() / ref["instances"].abs().max().item()
        err2 = (aux.cpu() - ref["instances_aux"]).abs().max().item()
        assert err < tol and err2 < tol, (dtype, err, err2)
        losses, d_inst, d_aux = vpu_step_losses(inst, aux, b["instances"].cuda())
        eng.backward(d_inst, d_aux)
        assert torch.isfinite(eng.gflat).all() and eng.gflat.abs().max() > 0
        torch.cuda.synchronize()
        print(f"smoke[{dtype}]: logits rel err {err:.2e}, aux abs err {err2:.2e}, loss {losses['total'].item():.5f}")
    # one fused-Adam step (after the parity checks: it changes the weights the oracle output was computed with)
    before = eng.flat.clone()
    FusedAdam(model).step()
    torch.cuda.synchronize()
    delta = (eng.flat - before).abs().max().item()
    assert 0 < delta < 1e-3 and torch.equal(eng.shadow, eng.flat.to(torch.bfloat16)), delta
    print("smoke ok")
