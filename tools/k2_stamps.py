#!/usr/bin/env python
"""Where a K2 forward / dgrad launch spends its time, per tile and workgroup: vpu_debug_gemm_times stamps s_memrealtime at
every tile's start, at the end of its main loop and at the end of its epilogue (3 stamps per tile, up to 5 tiles).
usage: python tools/k2_stamps.py"""
import os
os.environ.setdefault("VPU_LIB_DIAG", "1")       # the stamps exist in the -DVPU_DIAG build only (bash pvpuformer_amd/csrc/build.sh diag)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops, _lib  # noqa: E402

M = 9408
SHAPES = [
    ("qkv fwd", 0, M, 2304, 768, ops.EPI_BIAS),
    ("fc1 fwd gelu", 0, M, 3072, 768, ops.EPI_BIAS | ops.EPI_GELU | ops.EPI_SAVE_DGELU),
    ("fc2 fwd+res", 0, M, 768, 3072, ops.EPI_BIAS | ops.EPI_RESID),
    ("proj fwd+res", 0, M, 768, 768, ops.EPI_BIAS | ops.EPI_RESID),
    ("fc2 dgrad*aux", 1, M, 3072, 768, ops.EPI_MULAUX),
    ("fc1 dgrad", 1, M, 768, 3072, 0),
    ("qkv dgrad", 1, M, 768, 2304, 0),
    ("fc1 bias only", 0, M, 3072, 768, ops.EPI_BIAS),
]
if os.environ.get("K2_STAMPS_ONLY"):
    SHAPES = [s_ for s_ in SHAPES if any(k_ in s_[0] for k_ in os.environ["K2_STAMPS_ONLY"].split(","))]
dev = "cuda"
for name, tB, m, n, k, flags in SHAPES:
    A = (torch.rand(m, k, device=dev) - 0.5).to(torch.bfloat16)
    Bm = (torch.rand((k, n) if tB else (n, k), device=dev) - 0.5).to(torch.bfloat16)
    C = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
    kw = dict(transB=bool(tB), flags=flags, bias=torch.rand(n, device=dev), resid=torch.zeros(m, n, device=dev, dtype=torch.bfloat16), ldr=n,
              aux=torch.ones(m, n, device=dev, dtype=torch.bfloat16), ldaux=n, preact=torch.zeros(m, n, device=dev, dtype=torch.bfloat16))
    ldb = n if tB else k
    for _ in range(3):
        ops.gemm(A, Bm, C, m, n, k, k, ldb, n, 0, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm(A, Bm, C, m, n, k, k, ldb, n, 0, **kw)
    e1.record(); torch.cuda.synchronize()
    avg = e0.elapsed_time(e1) * 1e3 / 20
    buf = torch.zeros(256 * 16, dtype=torch.int64, device=dev)
    _lib.call("vpu_debug_gemm_times", buf.data_ptr())
    ops.gemm(A, Bm, C, m, n, k, k, ldb, n, 0, **kw)
    torch.cuda.synchronize()
    _lib.call("vpu_debug_gemm_times", None)
    t = buf.view(256, 16).cpu().double() / 100.0       # us
    live = t[:, 0] > 0
    t0 = t[live, 0].min()
    kern = ops.gemm_last_kernel()
    print(f"{name:14s} {kern[10:40]:30s} back-to-back {avg:6.1f} us; workgroups {int(live.sum())}")
    if (t[live, 11] > 0).all():     # fine stamps of tile 0 (256 x 128 form): barrier after the loop / exchange done / DMA primed / stores issued
        f = lambda i: float((t[live, i] - t[live, 1]).median())
        print(f"   tile 0 after its main loop: ring free +{f(10):4.2f} us, halves exchanged +{f(11):4.2f}, next stages requested +{f(12):4.2f}, "
              f"stores issued +{f(13):4.2f}, last barrier +{float((t[live, 2] - t[live, 1]).median()):4.2f}")
    elif (t[live, 13] > 0).all():   # 256-column direct form: operands requested / next stage requested / first half stored / second half stored
        f = lambda i: float((t[live, i] - t[live, 1]).median())
        print(f"   tile 0 after its main loop: operand loads requested +{f(10):4.2f} us, next stage requested +{f(11):4.2f}, first 64 rows stored +{f(12):4.2f}, "
              f"second +{f(13):4.2f}, last barrier +{float((t[live, 2] - t[live, 1]).median()):4.2f}")
    for ti in range(5):
        has = live & (t[:, 3 * ti + 2] > 0)
        if not has.any():
            break
        st, ml, ep = t[has, 3 * ti] - t0, t[has, 3 * ti + 1] - t0, t[has, 3 * ti + 2] - t0
        print(f"   tile {ti}: {int(has.sum()):3d} wgs  start {st.median():6.1f} (max {st.max():6.1f})  main loop {float((ml - st).median()):5.1f} us (max {float((ml - st).max()):5.1f})"
              f"  epilogue+handover {float((ep - ml).median()):5.1f} us (max {float((ep - ml).max()):5.1f})  done at {ep.median():6.1f} (max {ep.max():6.1f})")
