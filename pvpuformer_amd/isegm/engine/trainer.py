"""Training-step semantics of the reference's ISTrainer.batch_forward / add_loss (isegm/engine/trainer.py:310-491,
:533-554) on the HIP path: losses and their gradients come from the fused loss kernels, the model backward from the
engine's tape.  (The full trainer mirror -- click simulation loop, optimizer, DDP -- builds on these pieces.)"""
import torch

from pvpuformer_amd import ops


def vpu_step_losses(inst, aux, gt, slot_idx=None, override=None, iter_weight=1.0, w_nfl=1.0, w_dice=1.0, w_pcl=2.0,
                    want_grads=True, sim_low=None):
    """One click-iteration's loss = (NFL*1 + Dice*1 + P2CL*2) * iter_weight (vpu_base448_cocolvis.py:72-80,
    trainer.py:399-419) and its gradient w.r.t. the model outputs.

    inst fp32 [B,1,H,W] logits; aux fp32 [B,S,H,W] in [0,1]; gt fp32 [B,1,H,W]; slot_idx int32 [B,S] (-1 or an index
    into ``override`` [n,H,W], the per-slot error masks written by get_next_promts, trainer.py:756,764).
    ``sim_low`` fp32 [B,S,h,w] (with aux=None): the P2CL term is taken by the fused upsample+loss kernel on the
    low-resolution similarities and the third return value is the gradient w.r.t. ``sim_low`` instead of ``aux``.
    Returns ({'total','nfl','dice','p2cl'} device scalars, d_inst, d_aux_or_d_sim_low)."""
    H, W = inst.shape[-2:]
    B, S = (aux.shape[:2] if aux is not None else sim_low.shape[:2])
    dev = inst.device
    gt = gt.contiguous().float()
    sums = torch.zeros(B, 8, device=dev, dtype=torch.float64)
    out = torch.empty(B, 2, device=dev)
    d_inst = torch.empty_like(inst) if want_grads else None
    ops.nfl_dice_fwd_bwd(inst, gt, sums, out, d_inst, w_nfl * iter_weight / B, w_dice * iter_weight / B, B, H * W)
    part = torch.empty(B, S, device=dev)
    gs = w_pcl * iter_weight / (B * S * H * W)
    if aux is not None:
        d_aux = torch.empty_like(aux) if want_grads else None
        ops.p2cl_fwd_bwd(aux, gt, slot_idx, override, part, d_aux, gs, B, S, H, W)
    else:
        d_aux = torch.empty_like(sim_low) if want_grads else None
        ops.p2cl_up_fwd_bwd(sim_low, gt, slot_idx, override, part, d_aux, gs, B, S, sim_low.shape[2], sim_low.shape[3],
                            H, W)
    nfl, dice = out[:, 0].mean(), out[:, 1].mean()
    pcl = part.sum() / (B * S * H * W)
    total = (w_nfl * nfl + w_dice * dice + w_pcl * pcl) * iter_weight
    return {"total": total, "nfl": nfl, "dice": dice, "p2cl": pcl}, d_inst, d_aux
