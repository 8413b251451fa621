#!/usr/bin/env python
"""Which lines of the engine still run torch (aten) device work inside the bench step?  Runs bench.py under a
TorchDispatchMode and counts every aten call that touches a GPU tensor by the innermost pvpuformer_amd / bench.py frame.
usage: python tools/prof_aten.py [steps]   (counts cover warm-up + steps + the roofline / enqueue passes of bench.py)"""
import collections
import os
import runpy
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sys.argv = ["bench.py", "--steps", str(steps), "--warmup", "0", "--no-cpu-baseline"]
VIEW_ONLY = ("view", "reshape", "slice", "select", "as_strided", "expand", "permute", "transpose", "t.", "unsqueeze", "squeeze",
             "detach", "alias", "empty", "_unsafe_view", "unbind", "split", "narrow", "numel", "size", "stride", "is_", "sym_",
             "_local_scalar_dense", "lift_fresh", "record_stream", "set_", "resize_")
counts = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        full = str(func).replace("aten.", "")
        if not any(full.startswith(v) for v in VIEW_ONLY):
            on_gpu = any(torch.is_tensor(a) and a.is_cuda for a in list(args) + list((kwargs or {}).values()))
            out = func(*args, **(kwargs or {}))
            on_gpu = on_gpu or (torch.is_tensor(out) and out.is_cuda)
            if on_gpu:
                fr = next((f for f in reversed(traceback.extract_stack()) if ("pvpuformer_amd" in f.filename or f.filename.endswith("bench.py"))
                           and "tools" not in f.filename), None)
                counts[(full, f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.line[:90]}" if fr else "?")] += 1
            return out
        return func(*args, **(kwargs or {}))


with Spy():
    try:
        runpy.run_path("bench.py", run_name="__main__")
    except SystemExit:
        pass
for (name, where), n in sorted(counts.items(), key=lambda kv: -kv[1]):
    print(f"{n:5d} {name:30s} {where}")
