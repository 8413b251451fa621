#!/usr/bin/env python
"""How far the tiles of one XCD drift apart inside a packed weight-gradient launch (K4P, one 256 x 256 x 9408 tile per CU):
vpu_debug_gemm_times stamps the constant 100-MHz real-time counter at the start, the quarters and the end of every workgroup's main loop.
usage: python tools/k4_drift.py"""
import os
os.environ.setdefault("VPU_LIB_DIAG", "1")       # the stamps exist in the -DVPU_DIAG build only (bash pvpuformer_amd/csrc/build.sh diag)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops, _lib  # noqa: E402
from tools.k3_bench import problems  # noqa: E402

shapes = [(3072, 768), (768, 3072), (2304, 768), (768, 768)] * 2 + [(768, 3072), (256, 1024)]
probs, fl = problems(shapes)
if os.environ.get("K4_DRIFT_DCS", "1") == "1":      # distributed bias column sums (the engine's default), else the classic form
    for (args, kw), (m, n) in zip(probs, shapes):
        tn = (n + 255) // 256
        if tn > 1:
            kw.update(colsum=torch.zeros(tn, m, device="cuda"), cs_tn=tn, cs_t0=0, cs_ld=m)
ops.gemm_set_option("k3", 24)
for _ in range(3):
    ops.gemm_grouped(probs)
buf = torch.zeros(256 * 8, dtype=torch.int64, device="cuda")
_lib.call("vpu_debug_gemm_times", buf.data_ptr())
ops.gemm_grouped(probs)
torch.cuda.synchronize()
_lib.call("vpu_debug_gemm_times", None)
t = buf.view(256, 8).cpu().double()
print(ops.gemm_last_kernel())
MHZ = 100.0      # s_memrealtime: constant 100 MHz
for x in range(8):
    w = t[x::8]
    base = w[:, 0].min()
    rel = (w[:, :5] - base) / MHZ
    names = ["start", "1/4", "1/2", "3/4", "end"]
    txt = "  ".join(f"{n} {rel[:, i].min():7.1f}..{rel[:, i].max():7.1f} (spread {rel[:, i].max() - rel[:, i].min():5.1f})" for i, n in enumerate(names))
    print(f"XCD {x}: {txt} us")
dur = (t[:, 4] - t[:, 0]) / MHZ
print(f"main loop per workgroup: min {dur.min():.1f} us  median {dur.median():.1f}  max {dur.max():.1f}")
# per problem: which tiles are the slow ones (workgroup b -> tile v = (b & 7) * (total / 8) + (b >> 3) for total % 8 == 0)
tiles = [((m + 255) // 256) * ((n + 255) // 256) for m, n in shapes]
total = sum(tiles)
if total % 8 == 0 and total <= 256:
    import itertools
    starts = [0] + list(itertools.accumulate(tiles))
    per = {}
    for b in range(total):
        v = (b & 7) * (total // 8) + (b >> 3)
        g = max(i for i in range(len(tiles)) if starts[i] <= v)
        per.setdefault(g, []).append(float(dur[b]))
    for g, ds in per.items():
        ds = sorted(ds)
        print(f"problem {g:2d} {str(shapes[g]):14s} tiles {len(ds):3d}: main loop min {ds[0]:6.1f}  median {ds[len(ds) // 2]:6.1f}  max {ds[-1]:6.1f} us")
# are the slow workgroups the same ones launch after launch (a property of the CU / the tile) or random?
slow_sets = []
for rep in range(3):
    buf.zero_()
    _lib.call("vpu_debug_gemm_times", buf.data_ptr())
    ops.gemm_grouped(probs)
    torch.cuda.synchronize()
    _lib.call("vpu_debug_gemm_times", None)
    tt = buf.view(256, 8).cpu().double()
    dd = (tt[:, 4] - tt[:, 0]) / MHZ
    slow = [int(i) for i in torch.nonzero(dd > dd.median() * 1.08).flatten()]
    slow_sets.append(set(slow))
    hist = torch.histc(dd.float(), bins=10, min=float(dd.min()), max=float(dd.max()))
    print(f"launch {rep}: {len(slow)} workgroups more than 8 % above the median; histogram {[int(v) for v in hist]} over [{dd.min():.0f}, {dd.max():.0f}] us")
    print("   slow workgroups:", slow[:48])
print("slow in all three launches:", sorted(slow_sets[0] & slow_sets[1] & slow_sets[2]))
