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
    out = torch.empty(B, 2, device=dev)
    d_inst = torch.empty_like(inst) if want_grads else None
    ops.nfl_dice_fwd_bwd(inst, gt, None, out, d_inst, w_nfl * iter_weight / B, w_dice * iter_weight / B, B, H * W)
    part = torch.empty(B, S, device=dev)
    gs = w_pcl * iter_weight / (B * S * H * W)
    if aux is not None:
        d_aux = torch.empty_like(aux) if want_grads else None
        ops.p2cl_fwd_bwd(aux, gt, slot_idx, override, part, d_aux, gs, B, S, H, W)
    else:
        d_aux = torch.empty_like(sim_low) if want_grads else None
        ops.p2cl_up_fwd_bwd(sim_low, gt, slot_idx, override, part, d_aux, gs, B, S, sim_low.shape[2], sim_low.shape[3],
                            H, W)
    res = torch.empty(4, device=dev)
    ops.loss_finalize(out, part, B, part.numel(), 1.0 / (B * S * H * W), w_nfl, w_dice, w_pcl, iter_weight, res)
    return {"total": res[0], "nfl": res[1], "dice": res[2], "p2cl": res[3]}, d_inst, d_aux


class VPUTrainStep:
    """``ISTrainer.batch_forward`` + ``add_loss`` + backward + optimizer step for the VPU configuration
    (isegm/engine/trainer.py:310-491, 523-554, 186-202; constructor values of models/iSegNet/vpu_base448_cocolvis.py:163-179:
    max_num_next_clicks=3, iterloss_weights=[1,2,3], as_multi_prompts_ed_loss, as_allmask=False).

    Per click iteration: prompt type ~ randint(0,1) (click / box), forward on cat(image, prev_mask), NFL + Dice + 2*P2CL
    times the iteration weight, prev_mask = sigmoid(logits), then the next click / box / error-mask label from
    ``prompt_sim.get_next_promts``.  The iterations' graphs are independent (inputs detached), so each iteration is
    back-propagated immediately and its activations are dropped -- same gradient as the reference's single backward of
    the summed loss.  Gradient all-reduce buckets are launched during the LAST iteration's backward only.
    """

    def __init__(self, model, optimizer=None, reducer=None, max_num_next_clicks=3, iterloss_weights=(1, 2, 3),
                 prompt_types=(0, 1), as_allmask=False, loss_weights=(1.0, 1.0, 2.0)):
        self.model, self.opt, self.red = model, optimizer, reducer
        self.max_clicks, self.iter_w, self.ptypes = max_num_next_clicks, tuple(iterloss_weights), tuple(prompt_types)
        self.as_allmask, self.lw = as_allmask, loss_weights

    def batch_forward(self, batch, num_iters=None, rng=None, np_rng=None, record=None):
        import random

        import numpy as np

        from .prompt_sim import PromptState, get_next_promts
        rng = rng or random
        np_rng = np_rng or np.random
        eng = self.model._ensure_engine()
        image, gt, points = batch['images'], batch['instances'].float().contiguous(), batch['points'].float()
        B, _, H, W = image.shape
        dev = image.device
        S = 2 * self.model.num_max_points
        net_input = torch.cat([image, torch.zeros(B, 1, H, W, device=dev)], 1).contiguous()   # prev_output = 0 (:324)
        prev = net_input[:, 3:4]
        state = PromptState(B, S, H, W, dev, max_rounds=self.max_clicks)
        num_iters = num_iters or rng.randint(1, self.max_clicks)
        logged, boxes = {}, None
        from pvpuformer_amd.optim import FusedAdam
        if not isinstance(self.opt, FusedAdam):
            # an external optimizer stepped the fp32 master parameters: the bf16 shadow and the derived operands are stale
            # (the fused optimizer writes the shadow itself and refreshes the rest)
            eng.shadow_valid = False
        eng.zero_grad()
        if self.red is not None:
            self.red.begin()
        for it in range(num_iters):
            ptype = self.ptypes[rng.randint(0, len(self.ptypes) - 1)]
            if it == 0:   # boxes from the (empty) previous output; the returned click is discarded (:372-378)
                _, boxes = get_next_promts(prev, gt, points, None, as_allmask=self.as_allmask, np_rng=np_rng, rng=rng)
            mask = None
            if self.model.training and self.model.head.dropout_ratio > 0:
                keep = 1.0 - self.model.head.dropout_ratio
                mask = torch.bernoulli(torch.full((B, self.model.head.channels), keep, device=dev)) / keep
            if record is not None:
                record.append(dict(points=points.clone(), boxes=boxes.clone(), ptype=ptype, net_input=net_input.clone(),
                                   slot_idx=state.slot_idx.clone(), override=state.override.clone()))
            last = it == num_iters - 1
            eng.grad_ready_hook = self.red.ready if (self.red is not None and last) else None
            inst, _ = eng.forward(net_input, points, boxes, ptype, mask, training=True, materialize_aux=False)
            losses, d_inst, d_sim = vpu_step_losses(inst, None, gt, state.slot_idx, state.override,
                                                    iter_weight=float(self.iter_w[it]), w_nfl=self.lw[0],
                                                    w_dice=self.lw[1], w_pcl=self.lw[2], sim_low=eng.sim_low)
            eng.backward(d_inst, None, d_sim_low=d_sim)
            for k, v in losses.items():
                logged[f"{k}_{it}_{self.iter_w[it]}"] = v
            if not last:
                ops.sigmoid_to_channel(inst, net_input, B, H * W, 4, 3)          # prev_output = sigmoid(instances) (:428)
                points, boxes = get_next_promts(prev, gt, points, state, as_allmask=self.as_allmask, np_rng=np_rng,
                                                rng=rng)
        scale = self.red.finish() if self.red is not None else 1.0
        if self.opt is not None:
            self.opt.step(grad_scale=scale)
        logged["num_iters"] = num_iters
        return logged, points
