"""Training-step semantics of the reference's ISTrainer.batch_forward / add_loss (isegm/engine/trainer.py:310-491,
:533-554) on the HIP path: losses and their gradients come from the fused loss kernels, the model backward from the
engine's tape.  (The full trainer mirror -- click simulation loop, optimizer, DDP -- builds on these pieces.)"""
import torch

from pvpuformer_amd import ops
from .prompt_sim import cal_box, cal_scribble, get_iou, max_connected_regions   # noqa: F401  (module-level names of trainer.py:1045-1243)


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
        part = ops.p2cl_up_fwd_bwd(sim_low, gt, slot_idx, override, None, d_aux, gs, B, S, sim_low.shape[2],
                                   sim_low.shape[3], H, W)
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
        # The prompt simulators read results back (component sizes, distance maxima, the chosen pixel): on the training
        # stream every such read waits for everything queued before it -- the previous step's backward, or this
        # iteration's.  They run on a stream of their own instead (``use_sim_stream`` = False: on the training stream): the
        # iteration-0 box simulation depends on the batch only, the later ones on the forward output only, so the host
        # waits for the few small simulator kernels and keeps the training stream fed meanwhile.
        import os
        self.sim_stream = None
        self.use_sim_stream = True      # (attribute: the simulators on their own stream; settled A/B, no environment knob)
        # One click iteration = ~530 kernel launches the host needs 7-9 ms to enqueue -- beside the simulators' host work
        # that is more than the 13-14 ms the GPU needs for them.  The iteration is therefore captured once per (prompt type,
        # iteration number, batch geometry) as hipGraphs over static input buffers -- forward + losses, and backward, so that
        # the next prompts are still simulated beside the backward -- and replayed (VPU_TRAIN_GRAPH=0: always
        # host-enqueued).  Under a gradient reducer the backward that reports the finished gradient ranges -- the last
        # iteration's -- is a chain of graphs cut at the reports, the reducer's collectives launched by the host between
        # two segments (pvpuformer_amd/graphs.py).  First sight of a key runs host-enqueued (lazily created workspaces
        # and kernel attributes must exist before a capture), the second is captured.
        self.use_graph = os.environ.get("VPU_TRAIN_GRAPH", "1") != "0"
        self._static, self._passes, self._pool, self._reserving = {}, {}, None, False

    def upload(self, batch_cpu, device):
        """Host batch -> device on the simulator stream (not behind the previous step's kernels); the returned dict carries
        the event (``'_ready'``) the training stream has to pass before it touches the tensors."""
        if not self.use_sim_stream:
            return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch_cpu.items()}
        side = self._side(device)
        if any(torch.is_tensor(v) and v.is_cuda for v in batch_cpu.values()):
            side.wait_stream(torch.cuda.current_stream(device))     # device tensors: their producer is the current stream
        with torch.cuda.stream(side):
            out = {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch_cpu.items()}
            ev = torch.cuda.Event()
            ev.record(side)
        main = torch.cuda.current_stream(device)
        for v in out.values():
            if torch.is_tensor(v):
                v.record_stream(main)
        out['_ready'] = ev
        out['_host'] = batch_cpu          # (the scribble simulator works on the host copy of the ground truth)
        return out

    def _side(self, device):
        if self.sim_stream is None:
            self.sim_stream = torch.cuda.Stream(device=device)
        return self.sim_stream

    def _static_buffers(self, B, S, H, W, dev):
        """The inputs a captured pass reads, at fixed addresses: written before every replay."""
        from types import SimpleNamespace
        key = (B, S, H, W, str(dev))
        st = self._static.get(key)
        if st is None:
            st = SimpleNamespace(net_input=torch.zeros(B, 4, H, W, device=dev), points=torch.zeros(B, S, 3, device=dev),
                                 boxes=torch.zeros(B, 5, dtype=torch.int32, device=dev),
                                 gt=torch.zeros(B, 1, H, W, device=dev),
                                 slot_idx=torch.full((B, S), -1, dtype=torch.int32, device=dev),
                                 override=torch.zeros(self.max_clicks * B, H, W, device=dev), curve=None, prof=None)
            self._static[key] = st
        return st

    def _pass_body(self, eng, st, ptype, it):
        """forward + losses of one click iteration on the static buffers; returns what the backward needs."""
        B = st.net_input.shape[0]
        mask = None
        if self.model.training and self.model.head.dropout_ratio > 0:
            mask = ops.dropout_mask(B, self.model.head.channels, 1.0 - self.model.head.dropout_ratio, st.net_input.device)
        inst, _ = eng.forward(st.net_input, st.points, st.boxes, ptype, mask, training=True, materialize_aux=False,
                              scribble=(st.curve, st.prof) if ptype == 2 else None)
        losses, d_inst, d_sim = vpu_step_losses(inst, None, st.gt, st.slot_idx, st.override,
                                                iter_weight=float(self.iter_w[it]), w_nfl=self.lw[0], w_dice=self.lw[1],
                                                w_pcl=self.lw[2], sim_low=eng.sim_low)
        return inst, losses, d_inst, d_sim

    def _graph_pass(self, eng, st, ptype, it, after_forward, reducer=None):
        """One click iteration on the static buffers: host-enqueued the first time its key is seen, captured the second
        time, replayed from then on.  ``after_forward()`` runs between the forward + loss part and the backward (the
        event the simulator stream waits for).  Returns (logits, loss dict) -- the logits live in the capture's pool and
        are overwritten by the next replay of the same key.  ``reducer``: this backward reports its finished gradient ranges
        to it (``reducer.begin()`` has been called: the GEMM grids leave its CUs free)."""
        from types import SimpleNamespace
        from pvpuformer_amd.graphs import SegmentedBackward, capture
        hook = reducer.ready if reducer is not None else None
        key = (ptype, it, tuple(st.net_input.shape), tuple(st.points.shape), None if st.curve is None else tuple(st.curve.shape),
               bool(self.model.training), bool(eng.shadow_valid), id(eng),   # (a stale bf16 shadow is re-cast inside forward)
               reducer is not None, int(getattr(self.red, "reserve_cus", 0) or 0) if self._reserving else 0,
               bool(eng._lazy))   # (the first backward after zero_grad(lazy=True) WRITES the single-writer gradients: baked in)
        ent = self._passes.get(key)
        if ent is None or ent is False:
            if ent is None:
                self._passes[key] = "seen"
            eng.grad_ready_hook = hook
            inst, losses, d_inst, d_sim = self._pass_body(eng, st, ptype, it)
            after_forward()
            eng.backward(d_inst, None, d_sim_low=d_sim)
            eng.grad_ready_hook = None
            return inst, losses
        if ent == "seen":
            try:
                eng.grad_ready_hook = None
                fwd = torch.cuda.CUDAGraph()
                with capture(fwd, pool=self._pool, device=st.net_input.device):
                    inst, losses, d_inst, d_sim = self._pass_body(eng, st, ptype, it)
                if self._pool is None:
                    self._pool = fwd.pool()      # one pool for every captured pass: they never run beside each other, and
                # what a pass hands out (logits, losses) stays allocated
                bwd = SegmentedBackward.capture(eng, lambda: eng.backward(d_inst, None, d_sim_low=d_sim), hook_owner=reducer,
                                                pool=self._pool)      # (no reducer: nothing is reported, one segment)
                ent = SimpleNamespace(fwd=fwd, bwd=bwd, inst=inst, res=losses["total"]._base)
                self._passes[key] = ent
            except Exception as e:               # a capture enqueues nothing: this iteration is host-enqueued instead
                import warnings
                warnings.warn(f"VPUTrainStep: hipGraph capture failed ({type(e).__name__}: {str(e)[:120]}); this pass stays host-enqueued")
                torch.cuda.synchronize()
                eng.abort_pass()                 # (queue entries of the aborted capture must not reach the retry's launches)
                self._passes[key] = False
                return self._graph_pass(eng, st, ptype, it, after_forward, reducer)
        ent.fwd.replay()
        after_forward()
        ent.bwd.replay(hook)
        eng._lazy = set()                        # (what a lazily zeroed step left unwritten, this replay has written)
        res = ent.res.clone()                    # (the caller may read the losses after later replays)
        return ent.inst, {"total": res[0], "nfl": res[1], "dice": res[2], "p2cl": res[3]}

    def batch_forward(self, batch, num_iters=None, rng=None, np_rng=None, record=None, zero_grad=True, step=True,
                      grad_scale=1.0):
        """``zero_grad`` / ``step`` = False: gradient accumulation (trainer.py:188-202) -- the flat gradient buffer keeps
        what earlier batches left in it, the exchange and the optimizer step happen on the batch that closes the group;
        ``grad_scale`` multiplies into the optimizer's gradient scale (1 / accumulate_grad under ``cfg.amp``)."""
        import random

        import numpy as np

        from .prompt_sim import PromptState, get_next_promts
        rng = rng or random
        np_rng = np_rng or np.random
        eng = self.model._ensure_engine()
        ready = batch.get('_ready') if isinstance(batch, dict) else None
        image = batch['images']
        dev = image.device
        main = torch.cuda.current_stream(dev) if image.is_cuda else None
        side = self._side(dev) if (self.use_sim_stream and image.is_cuda) else None
        if ready is not None:
            main.wait_event(ready)          # the batch was uploaded on the simulator stream (upload())
        # dtype / layout conversions (uint8 or float64 masks, integer points) are kernels on the training stream: they come
        # BEFORE the hand-over to the simulator stream, whose iteration-0 box simulation reads gt and points
        gt, points = batch['instances'].float().contiguous(), batch['points'].float()
        converted = gt is not batch['instances'] or points is not batch['points']
        if side is not None and (ready is None or converted):
            side.wait_stream(main)          # producer unknown, or converted just now: everything queued so far comes first
            gt.record_stream(side); points.record_stream(side)
        B, _, H, W = image.shape
        S = 2 * self.model.num_max_points
        from pvpuformer_amd.optim import FusedAdam
        # (an optimizer other than the fused one steps the fp32 masters behind the engine's back: host-enqueued)
        graphed = (self.use_graph and image.is_cuda and record is None
                   and (self.opt is None or isinstance(self.opt, FusedAdam)))
        if graphed:     # the captured passes read their inputs from fixed buffers
            st = self._static_buffers(B, S, H, W, dev)
            st.net_input[:, :3].copy_(image)
            st.net_input[:, 3:].zero_()                                                       # prev_output = 0 (:324)
            st.gt.copy_(gt)
            st.slot_idx.fill_(-1)
            net_input = st.net_input
            state = PromptState(B, S, H, W, dev, max_rounds=self.max_clicks, buffers=(st.slot_idx, st.override))
        else:
            net_input = torch.cat([image, torch.zeros(B, 1, H, W, device=dev)], 1).contiguous()   # prev_output = 0 (:324)
            state = PromptState(B, S, H, W, dev, max_rounds=self.max_clicks)
        prev = net_input[:, 3:4]
        num_iters = num_iters or rng.randint(1, self.max_clicks)
        logged, boxes = {}, None
        if not isinstance(self.opt, FusedAdam):
            # an external optimizer stepped the fp32 master parameters: the bf16 shadow and the derived operands are stale
            # (the fused optimizer writes the shadow itself and refreshes the rest)
            eng.shadow_valid = False
        if zero_grad:
            eng.zero_grad(lazy=True)      # (this call's one backward follows; nothing reads the gradients in between)
        self._reserving = self.red is not None and step          # (begin() keeps the reducer's CUs out of the GEMM grids)
        if self.red is not None and step:
            self.red.begin()
        for it in range(num_iters):
            ptype = self.ptypes[rng.randint(0, len(self.ptypes) - 1)]
            if it == 0:   # boxes from the (empty) previous output; the returned click is discarded (:372-378)
                if side is not None:
                    with torch.cuda.stream(side):
                        _, boxes = get_next_promts(torch.zeros(B, 1, H, W, device=dev), gt, points, None,
                                                   as_allmask=self.as_allmask, np_rng=np_rng, rng=rng)
                    boxes.record_stream(main)
                    main.wait_stream(side)
                else:
                    _, boxes = get_next_promts(prev, gt, points, None, as_allmask=self.as_allmask, np_rng=np_rng, rng=rng)
            last = it == num_iters - 1
            scribble = None
            if ptype == 2:   # stroke over the ground-truth region, vectors drawn from `rng` (the reference: global random)
                from ..model.scribble import scribble_curves, scribble_profiles
                from .prompt_sim import cal_scribble
                host = batch.get('_host') if isinstance(batch, dict) else None
                if host is not None and not host['instances'].is_cuda:
                    gt_np = host['instances'].detach().float().numpy()[:, 0]
                elif side is not None:      # read back on the simulator stream: not behind the queued training kernels
                    with torch.cuda.stream(side):
                        gt_np = gt[:, 0].detach().cpu().numpy()
                else:
                    gt_np = gt[:, 0].detach().cpu().numpy()
                scr, rects = cal_scribble(gt_np > 0.5, rng=rng, np_rng=np_rng)
                scribble = (torch.from_numpy(scribble_curves(scr)), torch.from_numpy(scribble_profiles(scr, rects, H, rng)))
            fwd_done = [None]

            def after_forward():
                if side is not None and not last:
                    fwd_done[0] = torch.cuda.Event()    # after the loss kernels: they are the last readers of the slot table
                    fwd_done[0].record(main)            # and the override masks the simulator is about to update
            if graphed:
                st.points.copy_(points)
                st.boxes.copy_(boxes)
                if scribble is not None:
                    if st.curve is None or st.curve.shape != scribble[0].shape:
                        st.curve = torch.zeros(scribble[0].shape, dtype=torch.int32, device=dev)
                        st.prof = torch.zeros(scribble[1].shape, dtype=torch.float64, device=dev)
                    st.curve.copy_(scribble[0].to(torch.int32).pin_memory(), non_blocking=True)
                    st.prof.copy_(scribble[1].to(torch.float64).pin_memory(), non_blocking=True)
                inst, losses = self._graph_pass(eng, st, ptype, it, after_forward,
                                                self.red if (self.red is not None and last and step) else None)
            else:
                mask = None
                if self.model.training and self.model.head.dropout_ratio > 0:
                    keep = 1.0 - self.model.head.dropout_ratio
                    mask = ops.dropout_mask(B, self.model.head.channels, keep, dev)
                if record is not None:
                    record.append(dict(points=points.clone(), boxes=boxes.clone(), ptype=ptype, net_input=net_input.clone(),
                                       slot_idx=state.slot_idx.clone(), override=state.override.clone()))
                eng.grad_ready_hook = self.red.ready if (self.red is not None and last and step) else None
                inst, _ = eng.forward(net_input, points, boxes, ptype, mask, training=True, materialize_aux=False,
                                      scribble=scribble)
                losses, d_inst, d_sim = vpu_step_losses(inst, None, gt, state.slot_idx, state.override,
                                                        iter_weight=float(self.iter_w[it]), w_nfl=self.lw[0],
                                                        w_dice=self.lw[1], w_pcl=self.lw[2], sim_low=eng.sim_low)
                after_forward()
                eng.backward(d_inst, None, d_sim_low=d_sim)
            fwd_done = fwd_done[0]
            for k, v in losses.items():
                logged[f"{k}_{it}_{self.iter_w[it]}"] = v
            if not last and side is not None:
                # the next prompts need the forward output only: simulated on the side stream while this iteration's
                # backward (queued above) runs; nothing in backward reads the logits, the previous-mask channel or the
                # slot table
                with torch.cuda.stream(side):
                    side.wait_event(fwd_done)
                    if not graphed:
                        inst.record_stream(side)
                    ops.sigmoid_to_channel(inst, net_input, B, H * W, 4, 3)      # prev_output = sigmoid(instances) (:428)
                    points, boxes = get_next_promts(prev, gt, points, state, as_allmask=self.as_allmask, np_rng=np_rng,
                                                    rng=rng)
                points.record_stream(main); boxes.record_stream(main)
                main.wait_stream(side)
            elif not last:
                ops.sigmoid_to_channel(inst, net_input, B, H * W, 4, 3)          # prev_output = sigmoid(instances) (:428)
                points, boxes = get_next_promts(prev, gt, points, state, as_allmask=self.as_allmask, np_rng=np_rng,
                                                rng=rng)
        # the last iteration's logits (what the reference feeds its train metrics); a captured pass's live in its pool
        self.last_instances = inst.clone() if graphed else inst
        if step:
            # (under a reducer the update of everything but the front of the buffer starts while the step's last
            # collectives are still on the wire: parallel.finish_and_step)
            from pvpuformer_amd.parallel import finish_and_step
            finish_and_step(self.red, self.opt, grad_scale)
        logged["num_iters"] = num_iters
        return logged, points


def get_next_points(pred, gt, points, pred_thresh=0.49, np_rng=None):
    """The click-only simulator of the reference (isegm/engine/trainer.py:615-654): per sample one new click at a random
    pixel of the inner half (distance > max/2) of the larger of the false-negative / false-positive regions, written into
    the first free positive (negative) slot with order = largest order so far + 1.  pred [B,1,H,W] probabilities, gt
    [B,1,H,W], points [B,2n,3]; returns the updated points (a copy) on points' device.  The distance maps come from the
    exact transform (``vpu_edt`` on a CUDA ``pred``, scipy otherwise; the reference's cv2 chamfer transform is not
    available here: prompt_sim's parity note)."""
    import numpy as np

    from .prompt_sim import next_click
    pred_np = pred.detach().float().cpu().numpy()[:, 0]
    gt_np = gt.detach().cpu().numpy()[:, 0] > 0.5
    new_pts, _, _, _ = next_click(pred_np, gt_np, points.detach().float().cpu().numpy(), pred_thresh,
                                  np_rng or np.random, device=pred.device if pred.is_cuda else None)
    return torch.from_numpy(new_pts).to(points.device)


def _with_dense_label(pred, gt, points, ed_mask_label, pred_thresh, np_rng):
    """One simulated click per sample plus the reference's in-place label update: the slot that received the click now
    predicts this round's false-negative (positive click) / false-positive mask (trainer.py:756,764)."""
    import numpy as np

    from .prompt_sim import next_click
    pred_np = pred.detach().float().cpu().numpy()[:, 0]
    gt_np = (gt.detach().cpu().numpy() if torch.is_tensor(gt) else np.asarray(gt))[:, 0] > 0.5
    new_pts, picks, fn, fp = next_click(pred_np, gt_np, points.detach().float().cpu().numpy(), pred_thresh,
                                        np_rng or np.random, device=pred.device if pred.is_cuda else None)
    if ed_mask_label is not None:
        for b, pk in enumerate(picks):
            if pk is not None:
                mask = torch.from_numpy((fn if pk[1] else fp)[b].astype(np.int32))
                ed_mask_label[b, pk[0]] = mask.type_as(ed_mask_label).to(ed_mask_label.device)
    return torch.from_numpy(new_pts).to(points.device), gt_np, fn, fp


def get_next_points_and_mask(pred, gt, points, ed_mask_label, pred_thresh=0.49, np_rng=None):
    """trainer.py:656-700: ``get_next_points`` that also rewrites the clicked slot of the dense P2CL label."""
    new_pts, _, _, _ = _with_dense_label(pred, gt, points, ed_mask_label, pred_thresh, np_rng)
    return new_pts, ed_mask_label


def get_next_promts(pred, gt, points, ed_mask_label=None, pred_thresh=0.49, as_allmask=False, jitter_box=True):
    """The reference's call shape (trainer.py:703-768) over ``prompt_sim``: box from the state BEFORE the new click,
    scribble stroke of the ground truth, then the click; returns ``(points, boxes int32 [B,5], [scribbles, rectangles]``
    ``[, ed_mask_label])``.  Draw order of the generators as in the reference: ``random`` for the box jitter and the
    stroke, ``np.random`` for the stroke's coin and the click.  (The hot path calls ``prompt_sim.get_next_promts``, which
    keeps the label factored and everything on the device; this adapter serves drivers written against the reference.)"""
    import random

    import numpy as np

    from .prompt_sim import cal_box, cal_scribble
    pred_np = pred.detach().float().cpu().numpy()[:, 0]
    gt_np = (gt.detach().cpu().numpy() if torch.is_tensor(gt) else np.asarray(gt))
    gt_np = (gt_np[None] if len(gt_np) != len(pred_np) else gt_np[:, 0]) > 0.5        # trainer.py:708-712
    fn0 = np.logical_and(gt_np, pred_np < pred_thresh)
    fp0 = np.logical_and(np.logical_not(gt_np), pred_np > pred_thresh)
    boxes = cal_box(gt_np, fn0, fp0, points.detach().float().cpu().numpy(), as_allmask=as_allmask, jitter_box=jitter_box,
                    rng=random)
    scribbles = cal_scribble(gt_np, min_p=3, max_p=10, num_samples=1000, rng=random, np_rng=np.random)
    gt_t = torch.from_numpy(gt_np[:, None].astype(np.float32))
    new_pts, _, _, _ = _with_dense_label(pred, gt_t, points, ed_mask_label, pred_thresh, np.random)
    boxes = torch.from_numpy(boxes).to(points.device)
    if ed_mask_label is not None:
        return new_pts, boxes, scribbles, ed_mask_label
    return new_pts, boxes, scribbles


def load_weights(model, path_to_weights):
    """Partial load (``--weights``, trainer.py:1054-1058): keys present in the file replace the model's, the rest stay."""
    state = model.state_dict()
    state.update(torch.load(path_to_weights, map_location='cpu', weights_only=False)['state_dict'])
    model.load_state_dict(state)


class ISTrainer:
    """Constructor- and method-compatible mirror of the reference's trainer for the VPU configuration
    (isegm/engine/trainer.py:25-308; built exactly as models/iSegNet/vpu_base448_cocolvis.py:163-179 does), over
    ``VPUTrainStep``: ``ISTrainer(model, cfg, model_cfg, loss_cfg, trainset, valset, optimizer='adam', ...).run(n)``.

    What it does per batch is ``VPUTrainStep.batch_forward`` (1-3 click iterations, iteration-weighted NFL + Dice + P2CL,
    backward, gradient exchange, fused optimizer step); per epoch the learning-rate schedule and the checkpoint rule
    (``checkpoint_interval`` int or [(from_epoch, every)]).  Experiment logging, TensorBoard, image dumps, tqdm and the
    AMP grad scaler of the reference are outside the hot path and are not mirrored (bf16 needs no scaler);
    ``cfg.accumulate_grad`` groups batches the way trainer.py:188-202 does (optimizer step and gradient reset on every
    n-th batch and on the epoch's last one; the loss is divided by n only under ``cfg.amp``, like there); train metrics are
    reset per epoch and, on the master rank, updated after every batch with the last click iteration's logits and the
    ground truth (``m.update(instances, gt)``, trainer.py:483-487).  Only the configuration the
    reference can actually run is accepted: ``ed_loss`` with ``as_multi_prompts_ed_loss`` (its other two branches call
    ``_forward`` with too few arguments, trainer.py:395,397)."""

    def __init__(self, model, cfg, model_cfg, loss_cfg, trainset, valset, optimizer='adam', optimizer_params=None,
                 layerwise_decay=False, image_dump_interval=200, checkpoint_interval=10, tb_dump_period=25,
                 max_interactive_points=0, lr_scheduler=None, metrics=None, additional_val_metrics=None,
                 net_inputs=('images', 'points'), max_num_next_clicks=0, click_models=None, prev_mask_drop_prob=0.0,
                 use_iterloss=False, iterloss_weights=None, use_random_clicks=True, iter_train='epochiter',
                 penalty_loss=False, ed_loss=False, pclout=False, as_multi_prompts_ed_loss=False, as_allmask=True):
        from torch.utils.data import DataLoader

        from .optimizer import get_optimizer, get_optimizer_with_layerwise_decay
        if not (ed_loss and as_multi_prompts_ed_loss) or pclout or click_models is not None or penalty_loss:
            raise NotImplementedError("ISTrainer mirror: the VPU configuration only (ed_loss=True, "
                                      "as_multi_prompts_ed_loss=True, no click_models / pclout / penalty_loss)")
        self.cfg, self.model_cfg, self.loss_cfg = cfg, model_cfg, loss_cfg
        self.max_interactive_points, self.net_inputs = max_interactive_points, net_inputs
        self.max_num_next_clicks = max_num_next_clicks
        self.use_iterloss, self.iterloss_weights = use_iterloss, iterloss_weights
        self.as_allmask, self.ed_loss, self.as_multi_prompts_ed_loss = as_allmask, ed_loss, as_multi_prompts_ed_loss
        self.checkpoint_interval, self.image_dump_interval, self.tb_dump_period = checkpoint_interval, image_dump_interval, tb_dump_period
        self.train_metrics = list(metrics or [])
        self.val_metrics = list(metrics or []) + list(additional_val_metrics or [])
        self.task_prefix, self.current_epoch = '', 0
        get = (lambda k, d=None: cfg.get(k, d)) if hasattr(cfg, "get") else (lambda k, d=None: getattr(cfg, k, d))
        self._get = get
        distributed = bool(get("distributed", False))
        if distributed:
            cfg.batch_size //= cfg.ngpus
            cfg.val_batch_size //= cfg.ngpus
        self.trainset, self.valset = trainset, valset

        def loader(ds, bs, shuffle):
            if ds is None:
                return None
            sampler = None
            if distributed:
                sampler = torch.utils.data.distributed.DistributedSampler(ds, shuffle=shuffle)
            return DataLoader(ds, bs, sampler=sampler, shuffle=(shuffle and sampler is None), drop_last=True,
                              pin_memory=True, num_workers=get("workers", 0))
        self.train_data = loader(trainset, cfg.batch_size, True)
        self.val_data = loader(valset, get("val_batch_size", cfg.batch_size), False)
        self.device = get("device", "cuda")
        self.net = model.to(self.device)
        self.optim = (get_optimizer_with_layerwise_decay if layerwise_decay else get_optimizer)(
            self.net, optimizer, dict(optimizer_params or {}))
        self.lr = (optimizer_params or {}).get('lr')
        self.distributed, self.reducer, self._ready = distributed, None, False
        self._zero_next, self._step_now, self._grad_scale = True, True, 1.0
        if distributed:
            # the persistent GEMM grids leave `reserve_cus` CUs to RCCL's channel kernels while buckets are in flight; the
            # channel cap has to be in the environment before the communicator exists (the reference initialises it in
            # init_experiment, exp.py:29-32, i.e. before this constructor in a launched job -- then this is too late and
            # only warns)
            from pvpuformer_amd.parallel import configure_rccl_env
            configure_rccl_env()
        weights = (1, 2, 3) if not use_iterloss or not iterloss_weights else tuple(iterloss_weights)
        lw = (float(loss_cfg.get('instance_loss_weight', 1.0)), float(loss_cfg.get('instance_aux_loss_weight', 1.0)),
              float(loss_cfg.get('instance_aux3_loss_weight', 2.0)))
        self.step_fn = VPUTrainStep(self.net, self.optim, None, max_num_next_clicks=max(1, max_num_next_clicks),
                                    iterloss_weights=weights if use_iterloss else (1,) * max(1, max_num_next_clicks),
                                    as_allmask=as_allmask, loss_weights=lw)
        if lr_scheduler is not None:
            self.lr_scheduler = self._make_scheduler(lr_scheduler, self.optim)
            for _ in range(int(get("start_epoch", 0))):
                self.lr_scheduler.step()

    @staticmethod
    def _make_scheduler(factory, opt):
        """The model scripts pass ``partial(torch.optim.lr_scheduler.MultiStepLR, milestones=..., gamma=...)``
        (vpu_base448_cocolvis.py:153-154); torch's class only accepts a torch.optim.Optimizer, the fused optimizer gets
        this package's scheduler of the same rule."""
        if getattr(factory, "func", factory) is torch.optim.lr_scheduler.MultiStepLR:
            from pvpuformer_amd.optim import MultiStepLR
            return MultiStepLR(opt, *getattr(factory, "args", ()), **(getattr(factory, "keywords", None) or {}))
        return factory(optimizer=opt)

    def _setup_device_side(self):
        """First batch: bind the engine (needs the GPU), and under ``cfg.distributed`` do what DDP's constructor does --
        broadcast rank 0's parameters and buffers -- and attach the bucketed gradient reducer."""
        if self._ready:
            return
        from pvpuformer_amd.parallel import GradReducer, broadcast_parameters
        eng = self.net._ensure_engine()
        if self.distributed:
            broadcast_parameters(eng.flat)           # DDP's constructor broadcast (trainer.py:118-120)
            for b in self.net.buffers():             # e.g. the random pe_layer.positional_encoding_gaussian_matrix
                broadcast_parameters(b)
            eng.shadow_valid = False
            self.reducer = GradReducer(eng.gflat)
            self.step_fn.red = self.reducer
        self._ready = True

    @property
    def is_master(self):
        return int(self._get("local_rank", 0)) == 0

    def run(self, num_epochs, start_epoch=None, validation=True):
        start_epoch = int(self._get("start_epoch", 0)) if start_epoch is None else start_epoch
        for epoch in range(start_epoch, num_epochs):
            self.current_epoch = epoch
            self.training(epoch)
            if validation and self.val_data is not None:
                self.validation(epoch)

    def batch_forward(self, batch_data, validation=False):
        """One batch through the step (training) or a no-grad loss evaluation (validation).  Returns
        (loss, losses_logging, batch, outputs) like the reference (outputs: the logged scalars only -- the 38.5-MB/image
        auxiliary tensor is never materialised in training)."""
        self._setup_device_side()
        batch = self.step_fn.upload(batch_data, self.device) if not validation else \
            {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in batch_data.items()}
        if validation:
            with torch.no_grad():
                image = batch['images']
                B, _, H, W = image.shape
                x = torch.cat([image, torch.zeros(B, 1, H, W, device=image.device)], 1).contiguous()
                eng = self.net._ensure_engine()
                inst, _ = eng.forward(x, batch['points'].float(), None, 0, None, training=False, materialize_aux=False)
                losses, _, _ = vpu_step_losses(inst, None, batch['instances'].float(), None, None, want_grads=False,
                                               sim_low=eng.sim_low, w_nfl=self.step_fn.lw[0], w_dice=self.step_fn.lw[1],
                                               w_pcl=self.step_fn.lw[2])
            return losses["total"], dict(losses), batch, {"instances": inst}
        logged, _ = self.step_fn.batch_forward(batch, zero_grad=self._zero_next, step=self._step_now,
                                               grad_scale=self._grad_scale)
        if self.is_master and self.train_metrics:
            with torch.no_grad():
                inst, gt = self.step_fn.last_instances.float(), batch['instances'].float()
                for m in self.train_metrics:
                    if hasattr(m, "update"):
                        m.update(inst, gt)
        n = logged.pop("num_iters")
        loss = sum(v for k, v in logged.items() if k.startswith("total_"))
        logged["num_iters"] = n
        return loss, logged, batch, {}

    def training(self, epoch):
        from pvpuformer_amd.parallel import reduce_loss_dict
        if hasattr(getattr(self.train_data, "sampler", None), "set_epoch"):
            self.train_data.sampler.set_epoch(epoch)
        for m in self.train_metrics:
            if hasattr(m, "reset_epoch_stats"):
                m.reset_epoch_stats()
        self.net.train()
        self.last_train_loss = None
        acc = max(1, int(self._get("accumulate_grad", 1) or 1))
        self._grad_scale = 1.0 / acc if self._get("amp", False) else 1.0          # trainer.py:191-192 vs :199
        self._zero_next = True
        for i, batch_data in enumerate(self.train_data):
            self._step_now = (i + 1) % acc == 0 or i + 1 == len(self.train_data)    # trainer.py:188-189
            loss, logged, _, _ = self.batch_forward(batch_data)
            self._zero_next = self._step_now
            scal = {k: v for k, v in logged.items() if torch.is_tensor(v)}
            scal['overall'] = loss
            self.last_train_loss = reduce_loss_dict(scal)['overall']
        if self.is_master:
            ci = self.checkpoint_interval
            if isinstance(ci, (list, tuple)):
                ci = [x for x in ci if x[0] <= epoch][-1][1]
            if ci and epoch % ci == 0 and self._get("CHECKPOINTS_PATH") is not None:
                from ..utils.misc import save_checkpoint
                save_checkpoint(self.net, self._get("CHECKPOINTS_PATH"), prefix=self.task_prefix, epoch=epoch)
        if hasattr(self, 'lr_scheduler'):
            self.lr_scheduler.step()

    def validation(self, epoch):
        self.net.eval()
        tot, n = 0.0, 0
        for batch_data in self.val_data:
            loss, _, _, _ = self.batch_forward(batch_data, validation=True)
            tot += float(loss)
            n += 1
        self.last_val_loss = tot / max(n, 1)
        return self.last_val_loss
