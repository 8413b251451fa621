"""hipGraph capture of the engine's backward pass in data-parallel runs.

A backward pass is ~330 kernel launches; the host needs 5-9 ms to enqueue them one by one, the GPU ~9 ms to run them, and
eight ranks share one host.  On one GPU the whole step is captured as one graph (bench.py, VPUTrainStep); with a gradient
reducer that is not possible as it stands: the reducer's collectives are launched from the backward's
``grad_ready_hook`` -- host code between kernels, RCCL launches with their own stream and event plumbing.  Here the
backward is captured as a CHAIN of graphs cut at exactly those points: segment k holds the kernels up to the k-th report,
and on replay the host calls the real hook between two segments -- same kernels, same order, same collectives at the
same places of the stream, ~20 graph launches + the collectives instead of ~330 kernel launches.

The cut is lazy: the capture-time hook only notes the range; the graph is closed right before the next launch of the HIP
library, so a segment is never empty and kernels that do not go through the library (none today) would land before the
cut -- a range can be reported later than necessary, never earlier.
"""
import torch

from . import _lib, ops

_capture_streams = {}


def capture(g, pool=None, device=None):
    """``torch.cuda.graph(g, ...)`` on this package's capture stream of ``device``.  The GEMMs' split-K scratch is kept per
    stream and zero-filled when it is created: created here, before the first capture, the fill is not captured (on torch's
    own capture stream the 128-MB fill became a node of every graph that was the first to need the scratch).
    ``thread_local`` error mode: RCCL's watchdog thread polls events while a rank captures."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    st = _capture_streams.get(dev.index)
    if st is None:
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            ops.split_k_workspace(dev)
        st.synchronize()
        _capture_streams[dev.index] = st
    return torch.cuda.graph(g, pool=pool, stream=st, capture_error_mode="thread_local")


class _CutHook:
    """Stands in for the reducer's bound ``ready`` during capture (the engine reads ``__self__.reserve_cus`` of its hook:
    Engine._reserved_cus)."""

    def __init__(self, reserve_cus, pending):
        self.reserve_cus = reserve_cus
        self._pending = pending

    def ready(self, lo, hi):
        self._pending.append((int(lo), int(hi)))


class SegmentedBackward:
    """``capture(eng, run, hook_owner=None, pool=None)`` runs ``run()`` (a callable that executes ``eng.backward(...)``)
    under capture -- ``hook_owner``: the reducer whose ``ready`` the replay will call (None: no reports, one segment) --;
    ``replay(hook)`` launches the segments and calls ``hook(lo, hi)`` for every range where the engine reported it.
    ``segments``: [(graph or None, [(lo, hi), ...])]."""

    def __init__(self):
        self.segments = []
        self.pool = None

    @classmethod
    def capture(cls, eng, run, hook_owner=None, pool=None):
        self = cls()
        self.pool = pool
        pending, state = [], {"g": None, "ctx": None, "launched": False}
        orig_call, orig_hook = _lib.call, eng.grad_ready_hook

        def open_():
            g = torch.cuda.CUDAGraph()
            ctx = capture(g, pool=self.pool, device=eng.dev)
            ctx.__enter__()
            state.update(g=g, ctx=ctx, launched=False)

        def close_():
            state["ctx"].__exit__(None, None, None)
            if self.pool is None:
                self.pool = state["g"].pool()
            ranges = list(pending)
            del pending[:]
            if state["launched"]:
                self.segments.append((state["g"], ranges))
            elif ranges:                      # reported before anything was launched
                self.segments.append((None, ranges))
            state.update(g=None, ctx=None)

        def call(name, *args):
            if "_option" in name or "_last_" in name:        # settings / queries of the library: nothing is launched
                return orig_call(name, *args)
            if pending:                       # a range was reported since the last launch: this launch opens the next segment
                close_()
                open_()
            state["launched"] = True
            return orig_call(name, *args)

        # (no reducer: the engine sees no hook -- it queues and launches as in any single-GPU backward -- and nothing is cut)
        eng.grad_ready_hook = None if hook_owner is None else _CutHook(int(getattr(hook_owner, "reserve_cus", 0) or 0), pending).ready
        _lib.call = call
        open_()
        try:
            run()
        except BaseException:
            _lib.call, eng.grad_ready_hook = orig_call, orig_hook
            try:
                state["ctx"].__exit__(None, None, None)
            except Exception:
                pass
            eng.abort_pass()          # what the aborted pass queued points at capture-pool buffers nothing has written
            raise
        _lib.call, eng.grad_ready_hook = orig_call, orig_hook
        close_()
        return self

    def replay(self, hook):
        for g, ranges in self.segments:
            if g is not None:
                g.replay()
            if hook is not None:
                for lo, hi in ranges:
                    hook(lo, hi)
