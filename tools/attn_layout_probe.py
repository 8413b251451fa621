"""Is the window attention bound by its operand layout?  The model's q / k / v are 128-byte row segments of a [tokens][3 x 768]
matrix (row stride 4608 bytes: every (window, head) problem gathers 196 x 3 scattered cache lines and scatters as many); the same
kernels on head-major panels ([problem][196][64] contiguous: nb = 576, H = 1, ld = 64) move the same bytes as 25-KB streams.
usage: python tools/attn_layout_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops
dev = "cuda"; D, Hh, n, nb = 768, 12, 196, 48
def timeit(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
nset = int(os.environ.get("PROBE_SETS", "4"))     # distinct operand sets cycled through (cold caches, as in the step)
def mk(rows, cols): return [torch.randn(rows, cols, device=dev).to(torch.bfloat16) for _ in range(nset)]
# model layout
qkv, o, do, dqkv = mk(nb * n, 3 * D), mk(nb * n, D), mk(nb * n, D), mk(nb * n, 3 * D)
lse = torch.randn(nb * Hh, n, device=dev); delta = torch.empty(nb * Hh, n, device=dev)
cnt = [0]
def fwd_a():
    i = cnt[0] % nset; cnt[0] += 1
    ops.attn_fwd((qkv[i], 0), (qkv[i], D), (qkv[i], 2 * D), o[i], lse, nb, Hh, n, 64, 3 * D, D, 0.125)
def bwd_a():
    i = cnt[0] % nset; cnt[0] += 1
    ops.attn_bwd((qkv[i], 0), (qkv[i], D), (qkv[i], 2 * D), o[i], do[i], lse, delta, (dqkv[i], 0), (dqkv[i], D), (dqkv[i], 2 * D), nb, Hh, n, 64, 3 * D, D, 3 * D, 0.125)
ta, ka = timeit(fwd_a), ops.attn_last_kernel()
tb, kb = timeit(bwd_a), ops.attn_last_kernel()
print(f"model layout  [tokens][3 x 768]      fwd {ta:6.1f} us ({ka})   bwd {tb:6.1f} us ({kb})")
# head-major panels
P = nb * Hh
q2, k2, v2, o2, do2 = mk(P * n, 64), mk(P * n, 64), mk(P * n, 64), mk(P * n, 64), mk(P * n, 64)
dq2, dk2, dv2 = mk(P * n, 64), mk(P * n, 64), mk(P * n, 64)
lse2 = torch.randn(P, n, device=dev); delta2 = torch.empty(P, n, device=dev)
def fwd_b():
    i = cnt[0] % nset; cnt[0] += 1
    ops.attn_fwd(q2[i], k2[i], v2[i], o2[i], lse2, P, 1, n, 64, 64, 64, 0.125)
def bwd_b():
    i = cnt[0] % nset; cnt[0] += 1
    ops.attn_bwd(q2[i], k2[i], v2[i], o2[i], do2[i], lse2, delta2, dq2[i], dk2[i], dv2[i], P, 1, n, 64, 64, 64, 64, 0.125)
ta, ka = timeit(fwd_b), ops.attn_last_kernel()
tb, kb = timeit(bwd_b), ops.attn_last_kernel()
print(f"head-major    [problem][196][64]     fwd {ta:6.1f} us ({ka})   bwd {tb:6.1f} us ({kb})")
