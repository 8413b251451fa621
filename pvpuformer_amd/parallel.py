"""Data-parallel gradient exchange for the VPU training step: one process per GPU, RCCL (torch.distributed backend
"nccl" on ROCm) over xGMI.  Replaces the reference's DistributedDataParallel wrapper + loss reduce
(isegm/utils/distributed.py:25-67, isegm/engine/trainer.py:118-120).

The engine keeps all gradients in ONE flat fp32 buffer laid out in parameter order; backward finishes them from the
tail (head, neck) to the front (block 11 ... block 0, patch embeddings) and reports each finished contiguous range
through ``Engine.grad_ready_hook``.  The reducer coalesces ranges into buckets of >= ``bucket_bytes`` and launches an
asynchronous SUM all-reduce per bucket as soon as it is complete, so the exchange overlaps the rest of backward; the
29 never-used tensors are just zeros inside the flat buffer (no find_unused_parameters machinery).  The mean is folded
into the optimizer (grad_scale = 1/world_size).  xGMI is point-to-point (7 links x ~153 GB/s): few large buckets keep
every link busy.  Bucket size: the engine reports one range per ViT block (28 MB of fp32 gradients at ViT-B) after the
152-MB head + neck range; with 25-MB buckets every block's range goes out as soon as it is final, so the only exchange left
after backward ends is block 0 (28 MB) + the patch embeddings -- with 64-MB buckets it was the last three blocks (85 MB).
"""
import torch
import torch.distributed as dist


class GradReducer:
    """``wire``: "fp32" (the reference's DDP exchanges fp32 gradients) or "bf16": a bucket is rounded to bf16 into a
    staging buffer, summed in bf16 by the collective and widened back into the fp32 gradient -- half the bytes per xGMI
    link (a single ring is bound by one ~153 GB/s link; 489 MB of fp32 gradients cost 5.6 ms there, more than
    the compute of a 50 %-roofline step), at bf16 summation error over <= 8 addends.  ``force``: run the collectives even
    at world size 1 (tests: exercises RCCL launch / stream-wait plumbing on one GPU).
    ``reserve_cus``: while a step's buckets are in flight the persistent GEMM launches leave that many CUs unclaimed
    (vpu_gemm_set_option("reserve_cus")): RCCL's channel kernels are ordinary workgroups that need a CU to land on, and a
    persistent GEMM grid of one workgroup per CU (144 KiB of LDS, 2 x 256 registers per SIMD) leaves none -- without the
    reserve either the collective waits for a launch boundary, or, once its workgroups hold some CUs, every later
    256-workgroup persistent launch runs its last workgroups in a second round.  UNMEASURED on hardware (one GPU per box
    here); 0 disables."""

    def __init__(self, flat_grad, group=None, bucket_bytes=25 << 20, wire="fp32", force=False, reserve_cus=None):
        assert wire in ("fp32", "bf16")
        self.g = flat_grad
        self.group = group
        self.wire = wire
        self.bucket_elems = max(1, bucket_bytes // flat_grad.element_size())
        ready = dist.is_available() and dist.is_initialized()
        self.enabled = ready and (dist.get_world_size(group) > 1 or force)
        self.world = dist.get_world_size(group) if self.enabled else 1
        if reserve_cus is None:
            import os
            reserve_cus = int(os.environ.get("VPU_DIST_RESERVE_CUS", "16"))
        self.reserve_cus = reserve_cus if (self.enabled and flat_grad.is_cuda) else 0
        self._staged = []         # (lo, hi, bf16 staging tensor) of the bf16 buckets in flight
        self._pending_hi = None   # current open bucket is [lo, hi) growing downwards
        self._pending_lo = None
        self._works = []
        self.launched = []        # (lo, hi) of every collective of the current step (for tests / logging)
        self.on_bucket = None     # optional callable(lo, hi, work): called right after a bucket's collective is launched
        self.reduced_from = 0     # see finish(keep_last)
        self.last_split = 0       # where finish_and_step cut the optimizer step in two (0: it did not)
        self.trace = None         # a list: every launch / completed wait / mark() of the step is stamped (see summary())

    # ---- diagnostics of ONE step (bench.py at N > 1, tools/dp_sweep.py): where the exchange sits relative to backward
    def _stamp(self):
        """(host seconds, device event or None): the event is recorded on the CURRENT stream, so for a 'done' stamp --
        taken right after the stream was made to wait for a collective -- it fires when that collective has completed and
        everything queued before it on the compute stream has run."""
        import time
        ev = None
        if self.g.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
        return (time.perf_counter(), ev)

    def mark(self, name):
        if self.trace is not None:
            self.trace.append((name, 0, 0, self._stamp(), 0.0))

    def summary(self):
        """After the step has been synchronised: per-bucket launch / completion times relative to the 'bwd_end' mark, the
        exposed communication (end of backward -> last bucket complete, on the compute stream), bytes on the wire."""
        tr = self.trace or []
        def at(stamp, ref):
            (h, e), (h0, e0) = stamp, ref
            return e0.elapsed_time(e) if (e is not None and e0 is not None) else (h - h0) * 1e3
        ref = next((t[3] for t in tr if t[0] == "bwd_end"), None)
        esz = 2 if self.wire == "bf16" else self.g.element_size()
        buckets, last_done, host_wait = [], 0.0, 0.0
        launch = {(lo, hi): st for n, lo, hi, st, _ in tr if n == "launch"}
        for n, lo, hi, st, hw in tr:
            if n != "done":
                continue
            d = at(st, ref) if ref is not None else None
            l = at(launch[(lo, hi)], ref) if (ref is not None and (lo, hi) in launch) else None
            buckets.append({"mb": round((hi - lo) * esz / 1e6, 1), "launched_ms_vs_bwd_end": None if l is None else round(l, 3),
                            "done_ms_vs_bwd_end": None if d is None else round(d, 3), "host_wait_ms": round(hw * 1e3, 3)})
            host_wait += hw
            if d is not None:
                last_done = max(last_done, d)
        payload = sum((hi - lo) * esz for n, lo, hi, _, _ in tr if n == "launch")
        step_end = next((t[3] for t in tr if t[0] == "step_end"), None)
        return {"world_size": self.world, "collectives_per_step": len(launch), "wire_dtype": self.wire,
                "payload_bytes_per_step": int(payload),
                "wire_bytes_per_gpu_per_step": int(payload * 2 * (self.world - 1) / max(self.world, 1)),
                "exposed_comm_ms": round(last_done, 3) if ref is not None else None,
                "bwd_end_to_step_end_ms": round(at(step_end, ref), 3) if (ref is not None and step_end is not None) else None,
                "host_wait_ms_total": round(host_wait * 1e3, 3), "reserve_cus": self.reserve_cus,
                "bucket_mb": round(self.bucket_elems * self.g.element_size() / 1e6, 1), "buckets": buckets}

    def begin(self):
        self._pending_hi = self._pending_lo = None
        self._works = []
        self._staged = []
        self.launched = []
        if self.reserve_cus:
            from . import ops
            ops.gemm_set_option("reserve_cus", self.reserve_cus)

    def ready(self, lo, hi):
        """gflat[lo:hi] is final.  Ranges arrive tail-first and contiguous; anything else is flushed separately."""
        if not self.enabled or hi <= lo:
            return
        if self._pending_lo is not None and hi == self._pending_lo:
            self._pending_lo = lo
        else:
            self._flush()
            self._pending_lo, self._pending_hi = lo, hi
        if self._pending_hi - self._pending_lo >= self.bucket_elems:
            self._flush()

    def _flush(self):
        if self._pending_lo is None:
            return
        lo, hi = self._pending_lo, self._pending_hi
        self._pending_lo = self._pending_hi = None
        if self.wire == "bf16":
            stage = self.g[lo:hi].to(torch.bfloat16)
            work = dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._staged.append((lo, hi, stage))
        else:
            work = dist.all_reduce(self.g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append(work)
        self.launched.append((lo, hi))
        if self.trace is not None:
            self.trace.append(("launch", lo, hi, self._stamp(), 0.0))
        if self.on_bucket is not None:
            self.on_bucket(lo, hi, work)

    def finish(self, keep_last=0):
        """Launches the last partial bucket and makes the current stream wait for every collective -- all but the last
        ``keep_last`` of them (those cover the FRONT of the buffer: ranges arrive tail-first), which a second call
        ``finish()`` waits for.  Returns the factor the optimizer must apply to the summed gradient; after a call with
        ``keep_last`` > 0, ``self.reduced_from`` is the element offset from which the buffer is final."""
        if self.enabled:
            self._flush()
            n_wait = max(0, len(self._works) - keep_last) if keep_last else len(self._works)
            first = len(self.launched) - len(self._works)          # index in `launched` of the oldest collective in flight
            if keep_last and n_wait < len(self._works):
                # [reduced_from, total) may only be called final if the buckets really went out tail-first: every range that
                # stays in flight must lie wholly below every range that has been waited for.  ready() accepts ranges in any
                # order (it flushes them separately), so check instead of assuming -- otherwise wait for everything.
                waited, kept = self.launched[first:first + n_wait], self.launched[first + n_wait:]
                cut = min((lo for lo, _ in waited), default=None)
                if cut is None or any(hi > cut for _, hi in kept):
                    n_wait = len(self._works)
            done_lo = min(lo for lo, _ in self.launched[first:first + n_wait]) if n_wait else None
            import time
            for i, w in enumerate(self._works[:n_wait]):
                t0 = time.perf_counter()
                w.wait()
                if self.trace is not None:
                    lo, hi = self.launched[first + i]
                    self.trace.append(("done", lo, hi, self._stamp(), time.perf_counter() - t0))
            self._works = self._works[n_wait:]
            rest = []
            for lo, hi, stage in self._staged:
                if done_lo is None or lo < done_lo:
                    rest.append((lo, hi, stage))            # its collective is still in flight
                else:
                    self.g[lo:hi].copy_(stage)
            self._staged = rest
            self.reduced_from = 0 if not self._works else (done_lo if done_lo is not None else self.g.numel())
            if self.reserve_cus and not self._works:
                from . import ops
                ops.gemm_set_option("reserve_cus", 0)
        else:
            self.reduced_from = 0
        return 1.0 / self.world


def finish_and_step(red, opt, grad_scale=1.0, keep_last=None):
    """``opt.step(grad_scale * red.finish())`` with the optimizer's work started before the exchange has ended: the last
    ``keep_last`` collectives of a step -- block 0, the patch embeddings: ranges the backward can only hand over when it
    ends -- are on the wire when nothing of the backward is left to hide them; the Adam update of everything ELSE (an
    HBM-bound stream over ~90 % of the parameters, 0.55 ms at ViT-B) does not need them.  So: wait for all but the last
    collectives, update [reduced_from, total), wait for the rest, update [0, reduced_from).  Same arithmetic as one launch
    (the update is element-wise).  Falls back to the plain sequence for a per-tensor hyper-parameter table, a capturable
    optimizer or a disabled reducer.  ``keep_last`` None: ``VPU_DIST_SPLIT_ADAM`` (default 0 = the plain sequence: the split
    has only run at world size 1 on RCCL and over gloo; opt in once it has been measured on a multi-GPU node)."""
    from . import ops
    from .optim import FusedAdam
    if keep_last is None:
        import os
        keep_last = int(os.environ.get("VPU_DIST_SPLIT_ADAM", "0"))
    plain = (red is None or not red.enabled or not isinstance(opt, FusedAdam) or opt.capturable or opt.per_param is not None
             or opt.decoupled or keep_last <= 0)
    if plain:
        scale = red.finish() if red is not None else 1.0
        if opt is not None:
            opt.step(grad_scale=scale * grad_scale)
        return scale
    scale = red.finish(keep_last=keep_last)
    cut = red.last_split = red.reduced_from        # (kept for tests / logging)
    eng = opt.model._ensure_engine()
    if cut <= 0 or cut >= eng.total:          # nothing was kept back (few buckets) -- or everything
        red.finish()
        opt.step(grad_scale=scale * grad_scale)
        return scale
    m, v = opt._state(eng)
    opt.step_count += 1
    sh = lambda o: None if eng.shadow is None else (eng.shadow, o)
    args = (opt.lr, opt.betas[0], opt.betas[1], opt.eps, opt.weight_decay, opt.step_count, scale * grad_scale)
    ops.adam_step((eng.flat, cut), (eng.gflat, cut), (m, cut), (v, cut), sh(cut), eng.total - cut, *args)
    red.finish()
    ops.adam_step(eng.flat, eng.gflat, m, v, sh(0), cut, *args)
    eng.refresh_weights(shadow_is_fresh=True)
    return scale


def configure_rccl_env(channels=None):
    """Call BEFORE the RCCL communicator is created (``init_process_group``).  An RCCL collective runs as one workgroup per
    channel, each on a CU of its own for the whole collective; the persistent GEMM launches of the backward pass own one
    workgroup per CU with static tile lists, so a collective that takes more CUs than ``GradReducer.reserve_cus`` leaves
    free pushes the last GEMM workgroups into a second round.  The channel count is therefore capped at the reserve
    (``NCCL_MAX_NCHANNELS``, honoured by RCCL; an explicit setting in the environment wins).  16 channels move the 856 MB a
    ViT-B step puts on the wire per GPU well inside the backward pass (needed: ~110 GB/s of the 7 x 153 GB/s links).
    UNMEASURED on hardware, like the reserve itself."""
    import os
    import warnings
    if channels is None:
        channels = int(os.environ.get("VPU_DIST_RESERVE_CUS", "16"))
    if channels > 0 and "NCCL_MAX_NCHANNELS" not in os.environ:
        if dist.is_available() and dist.is_initialized():
            warnings.warn("configure_rccl_env() after init_process_group: a communicator that already exists keeps its "
                          "channel count; call pvpuformer_amd.install() (or this function) before the process group is created")
        os.environ["NCCL_MAX_NCHANNELS"] = str(channels)
        if os.environ.get("RANK", "0") == "0":
            import sys
            sys.stderr.write(f"pvpuformer_amd: NCCL_MAX_NCHANNELS={channels} (= the CUs the persistent GEMM grids leave to "
                             f"RCCL; set NCCL_MAX_NCHANNELS or VPU_DIST_RESERVE_CUS to override)\n")
    return os.environ.get("NCCL_MAX_NCHANNELS")


def broadcast_parameters(flat_param, src=0, group=None):
    """Identical replicas at start (DDP ctor broadcast, trainer.py:118-120): one collective over the flat buffer."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_param, src=src, group=group)


def reduce_loss_dict(losses, group=None):
    """isegm/utils/distributed.py:25-47: mean of the logged scalar losses over ranks (one small all-reduce)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
        return losses
    keys = sorted(losses)
    t = torch.stack([losses[k].detach().float().reshape(()) for k in keys])
    dist.all_reduce(t, group=group)
    t /= dist.get_world_size(group)
    return {k: t[i] for i, k in enumerate(keys)}
