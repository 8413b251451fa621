"""Fused Adam / AdamW over the engine's flat parameter buffer (torch.optim.Adam semantics as configured by the
reference at models/iSegNet/vpu_base448_cocolvis.py:149-154: lr 5e-5, betas (0.9, 0.999), eps 1e-8).
isegm/engine/optimizer.py:6-27 builds one param group PER TENSOR -> 349 tiny kernels per step; here it is ONE launch that
also writes the bf16 shadow used by the MFMA GEMMs.  Per-tensor learning rates / weight decay (``param.lr_mult``,
layer-wise lr decay of isegm/utils/lr_decay.py) are a small segment table read by the same kernel."""
import torch

from . import ops


# Tensors of the VPU model that never receive a gradient (unused at upsample='x1' / never called in forward):
# the reference leaves their .grad at None, so torch.optim.Adam skips them -- no weight decay, no moment update.
NEVER_USED_PREFIXES = ("backbone.cls_token", "backbone.fc_norm.", "backbone.head.", "head.logit_scale", "head.up_conv1.",
                       "head.up_conv2.", "point_embeddings.", "not_a_point_embed.", "head_aux.")


def is_never_used(name):
    return any(name == p or name.startswith(p) for p in NEVER_USED_PREFIXES)


class FusedAdam:
    def __init__(self, model, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled_weight_decay=False,
                 per_param=None, capturable=False):
        """``per_param``: optional {parameter name: (lr_scale, weight_decay)}; names missing from it use (1, weight_decay).
        The effective learning rate of a tensor is ``self.lr * lr_scale`` (so a scheduler only touches ``self.lr``).
        ``capturable``: the step-dependent scalars (lr, bias corrections, grad_scale) live in device memory: call
        ``prepare_step(grad_scale)`` on the host before every ``step()`` -- ``step()`` itself can then be captured in a
        hipGraph and replayed (it no longer advances the step count)."""
        self.model = model
        self.capturable = capturable
        self._hyper = None
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.decoupled = decoupled_weight_decay
        self.per_param = dict(per_param) if per_param else None
        if self.per_param is None and weight_decay != 0.0:
            self.per_param = {}      # the segment table is what exempts never-used / frozen tensors from weight decay
        self.step_count = 0
        self.m = self.v = None
        self._seg = None

    # torch.optim-like view used by schedulers / loggers
    @property
    def param_groups(self):
        return [{"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}]

    def _state(self, eng):
        if self.m is None or self.m.numel() != eng.total or self.m.device != eng.flat.device:
            self.m = torch.zeros_like(eng.flat)
            self.v = torch.zeros_like(eng.flat)
        return self.m, self.v

    def _segments(self, eng):
        """(seg_end int64, lr_scale fp32, wd fp32) on the device, one segment per tensor of the flat buffer."""
        if self._seg is None:
            names = list(eng.names.items())
            frozen = {n for n, p in self.model.named_parameters() if not p.requires_grad}
            ends, scales, wds = [], [], []
            for i, (n, (off, _, numel)) in enumerate(names):
                end = names[i + 1][1][0] if i + 1 < len(names) else eng.total   # padding belongs to the tensor before it
                sc, wd = self.per_param.get(n, (1.0, self.weight_decay))
                if is_never_used(n) or n in frozen:   # torch.optim.Adam skips tensors whose .grad is None / frozen ones
                    sc, wd = 0.0, 0.0
                ends.append(end); scales.append(float(sc)); wds.append(float(wd))
            dev = eng.flat.device
            self._seg = (torch.tensor(ends, dtype=torch.int64, device=dev), torch.tensor(scales, device=dev),
                         torch.tensor(wds, device=dev), len(ends))
        return self._seg

    def prepare_step(self, grad_scale=1.0):
        """capturable mode: advance the step count and upload {lr, 1-b1^t, sqrt(1-b2^t), grad_scale} (pinned, async)."""
        eng = self.model._ensure_engine()
        if self._hyper is None:
            self._hyper = (torch.empty(4, dtype=torch.float32).pin_memory(), torch.empty(4, device=eng.flat.device))
        self.step_count += 1
        host, dev = self._hyper
        host[0], host[1] = self.lr, 1.0 - self.betas[0] ** self.step_count
        host[2], host[3] = (1.0 - self.betas[1] ** self.step_count) ** 0.5, grad_scale
        dev.copy_(host, non_blocking=True)

    def step(self, grad_scale=1.0):
        """``grad_scale`` multiplies the gradient first (1/world_size after a SUM all-reduce)."""
        eng = self.model._ensure_engine()
        m, v = self._state(eng)
        if self.capturable:
            assert self._hyper is not None, "capturable FusedAdam: call prepare_step() before step()"
            if self.per_param is None and not self.decoupled:
                ops.adam_step_hyper(eng.flat, eng.gflat, m, v, eng.shadow, eng.total, self._hyper[1], None, None, None, 0,
                                    self.betas[0], self.betas[1], self.eps, self.weight_decay, False)
            else:
                if self.per_param is None:
                    self.per_param = {}
                ends, scales, wds, nseg = self._segments(eng)
                ops.adam_step_hyper(eng.flat, eng.gflat, m, v, eng.shadow, eng.total, self._hyper[1], ends, scales, wds,
                                    nseg, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.decoupled)
            eng.refresh_weights(shadow_is_fresh=True)
            return
        self.step_count += 1
        if self.per_param is None and not self.decoupled:
            ops.adam_step(eng.flat, eng.gflat, m, v, eng.shadow, eng.total, self.lr, self.betas[0], self.betas[1],
                          self.eps, self.weight_decay, self.step_count, grad_scale)
        else:
            if self.per_param is None:
                self.per_param = {}
            ends, scales, wds, nseg = self._segments(eng)
            ops.adam_step_groups(eng.flat, eng.gflat, m, v, eng.shadow, eng.total, ends, scales * self.lr, wds, nseg,
                                 self.betas[0], self.betas[1], self.eps, self.decoupled, self.step_count, grad_scale)
        eng.refresh_weights(shadow_is_fresh=True)

    def zero_grad(self, set_to_none=False):
        self.model._ensure_engine().zero_grad()

    def state_dict(self):
        """Adam state plus the Dropout2d mask stream of the model's device ({"seed", "calls"}: the reference's dropout follows
        torch's RNG state, which a checkpoint of a run can carry; here the stream is (seed, call number) in device memory)."""
        from . import ops
        sd = {"step": self.step_count, "m": self.m, "v": self.v, "lr": self.lr}
        if self.m.is_cuda:
            sd["dropout"] = ops.dropout_state(self.m.device)
        return sd

    def load_state_dict(self, sd):
        self.step_count, self.m, self.v, self.lr = sd["step"], sd["m"], sd["v"], sd["lr"]
        if sd.get("dropout") is not None and self.m.is_cuda:
            from . import ops
            ops.set_dropout_state(self.m.device, sd["dropout"])


class OverlappedAdam:
    """The optimizer step of ``FusedAdam`` run range by range on a second HIP stream WHILE backward is still running.

    The engine reports every finished contiguous range of the flat gradient buffer through ``Engine.grad_ready_hook``
    (tail first: head + neck, then ViT block 11 ... 0, then the patch embeddings); the parameters of a finished range
    are not read again by the rest of backward, so its Adam update -- an HBM-bound stream of 30 bytes per parameter --
    can run beside the MFMA-bound GEMMs of the earlier blocks instead of after them (0.65 ms of the 17 ms ViT-B step
    when serialised).  With a ``GradReducer`` the update of a bucket waits for that bucket's all-reduce only.  The same
    stream zeroes the range afterwards (``zero_grads``), which replaces the whole-buffer memset of the next step.
    Uniform hyper-parameters only (no per-tensor table); identical arithmetic to ``FusedAdam.step``.

        ov = OverlappedAdam(opt, eng, reducer);  ...forward, losses...;  ov.begin();  eng.backward(...);  ov.finish()
    """

    def __init__(self, opt, eng, reducer=None, zero_grads=True):
        assert opt.per_param is None and not opt.decoupled and not opt.capturable, "uniform, non-capturable Adam only"
        self.opt, self.eng, self.red, self.zero_grads = opt, eng, reducer, zero_grads
        self.side = torch.cuda.Stream(device=eng.flat.device)
        self.use_red = reducer is not None and reducer.enabled
        if self.use_red:
            reducer.on_bucket = self._range
        eng.grad_ready_hook = self.ready
        self.done = []            # ranges updated in the current step (tests)

    def begin(self):
        self.opt._state(self.eng)
        self.opt.step_count += 1
        self.done = []
        if self.use_red:
            self.red.begin()

    def ready(self, lo, hi):
        if self.use_red:
            self.red.ready(lo, hi)       # -> _range(bucket) once the bucket's collective is launched
        else:
            self._range(lo, hi, None)

    def _range(self, lo, hi, work):
        o, e = self.opt, self.eng
        main = torch.cuda.current_stream(e.flat.device)
        self.side.wait_stream(main)      # the range is final in main-stream order
        scale = 1.0 / self.red.world if self.use_red else 1.0
        with torch.cuda.stream(self.side):
            if work is not None:
                work.wait()              # this stream waits for the bucket's all-reduce
            sh = None if e.shadow is None else (e.shadow, lo)
            ops.adam_step((e.flat, lo), (e.gflat, lo), (o.m, lo), (o.v, lo), sh, hi - lo, o.lr, o.betas[0], o.betas[1],
                          o.eps, o.weight_decay, o.step_count, scale)
            if self.zero_grads:
                e.gflat[lo:hi].zero_()
        self.done.append((lo, hi))

    def finish(self):
        """After ``eng.backward``: the last bucket goes out, the main stream joins the optimizer stream and the four small
        derived operands are rebuilt from the updated master weights."""
        e = self.eng
        if self.use_red:
            self.red.finish()
        torch.cuda.current_stream(e.flat.device).wait_stream(self.side)
        e.refresh_weights(shadow_is_fresh=True)


class MultiStepLR:
    """torch.optim.lr_scheduler.MultiStepLR as the reference configures it (models/iSegNet/vpu_base448_cocolvis.py:
    ``partial(MultiStepLR, milestones=[50, 55], gamma=0.1)``, stepped once per epoch, trainer.py:204-206)."""

    def __init__(self, optimizer, milestones, gamma=0.1, last_epoch=-1):
        self.optimizer, self.milestones, self.gamma = optimizer, sorted(milestones), gamma
        self.base_lr = optimizer.lr
        self.last_epoch = last_epoch
        self.step()

    def get_last_lr(self):
        return [self.optimizer.lr]

    def get_lr(self):
        return self.get_last_lr()

    def step(self):
        self.last_epoch += 1
        passed = sum(1 for m in self.milestones if m <= self.last_epoch)
        self.optimizer.lr = self.base_lr * (self.gamma ** passed)
